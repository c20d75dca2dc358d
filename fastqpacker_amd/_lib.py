"""ctypes loader for libfqzhip.so — the C ABI declared in include/fqz.h."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("FQZ_LIB_PATH") or os.path.join(_HERE, "lib", "libfqzhip.so")  # FQZ_LIB_PATH: A/B builds in tools/

ENCODING_PHRED33 = 0
ENCODING_PHRED64 = 1
DETECT_ENCODING = -1
BATCH_FINAL = 1
BATCH_V3 = 2
BATCH_HALVES = 8  # experimental: two halves of a large batch in flight (include/fqz.h)
BATCH_SEG = 4  # experimental FQZ-S1 segment framing (include/fqz.h)
DEFAULT_BLOCK_SIZE = 100000
STREAM_NAMES = ["seq", "qual", "headers", "plus", "npos", "lengths"]


class FqzError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().fqz_strerror(code).decode() if _lib is not None else "libfqzhip error %d" % code
        if code in (-30, -31) and _lib is not None:
            detail = (detail + " " + _lib.fqz_last_hip_error().decode()).strip()
        super().__init__(msg + (": " + detail if detail else ""))


class BatchResult(C.Structure):
    _fields_ = [("n_records", C.c_uint32), ("n_blocks", C.c_uint32), ("consumed", C.c_uint64), ("out_len", C.c_uint64),
                ("status", C.c_int32), ("error_record", C.c_uint32), ("qual_encoding", C.c_int32), ("n_chunks", C.c_uint32),
                ("stream_raw", C.c_uint64 * 6), ("stream_comp", C.c_uint64 * 6)]


class Options(C.Structure):
    """compress.Options (compress.go:74-77) + container_version (0 / 2: the reference's CurrentVersion; 3: FQZ-R1, rANS-coded
    qualities - SURVEY 8 f-4, not readable by the stock decoder)."""
    _fields_ = [("block_size", C.c_uint32), ("workers", C.c_int32), ("container_version", C.c_uint32), ("block_index", C.c_uint32)]


class DecompressOptions(C.Structure):
    """compress.DecompressOptions (compress.go:80-82)."""
    _fields_ = [("workers", C.c_int32)]


class FileHeader(C.Structure):
    _fields_ = [("version", C.c_uint8), ("block_size", C.c_uint32), ("flags", C.c_uint8)]


class BlockHeader(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_records", "seq_size", "qual_size", "header_size", "plus_size", "npos_size",
                                           "lengths_size", "original_seq_size", "original_qual_size")]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_record", C.c_uint64), ("min_len", C.c_uint32), ("max_len", C.c_uint32),
                ("n_permille", C.c_uint32), ("phred", C.c_uint32), ("quality_profile", C.c_uint32)]


def library_path():
    return _SO


def build(force=False):
    """Compile libfqzhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8", "-s"]
    if force:
        args.insert(1, "-B")
    subprocess.check_call(args)


_lib = None

# name -> (restype, argtypes); the single source of truth for tests that check the exports
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p
READ_FN = C.CFUNCTYPE(C.c_long, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)   # fqz_read_fn
WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)   # fqz_write_fn
SIGNATURES = {
    "fqz_strerror": (C.c_char_p, [C.c_int]),
    "fqz_last_hip_error": (C.c_char_p, []),
    "fqz_version": (C.c_char_p, []),
    "fqz_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "fqz_ctx_destroy": (None, [_vp]),
    "fqz_device_count": (C.c_int, []),
    "fqz_write_file_header": (None, [C.POINTER(FileHeader), _u8p]),
    "fqz_read_file_header": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(FileHeader)]),
    "fqz_write_block_header": (C.c_int, [C.POINTER(BlockHeader), C.c_uint8, _u8p]),
    "fqz_read_block_header": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint8, C.POINTER(BlockHeader)]),
    "fqz_encode_bound": (C.c_size_t, [C.c_size_t]),
    "fqz_encode_bound_blocks": (C.c_size_t, [C.c_size_t, C.c_uint32]),
    "fqz_encode_block": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "fqz_decode_block": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint8, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "fqz_decode_block_size": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint8, C.POINTER(C.c_size_t)]),
    "fqz_encode_batch_dev": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_int, C.c_uint32, _vp, C.c_size_t, C.POINTER(BatchResult),
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t, _vp]),
    "fqz_encode_batch_launch": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_int, C.c_uint32, _vp, C.c_size_t, _vp]),
    "fqz_encode_batch_finish": (C.c_int, [_vp, C.POINTER(BatchResult), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t]),
    "fqz_decode_batch_dev": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint8, C.c_int, _vp, C.c_size_t, C.POINTER(BatchResult), _vp]),
    "fqz_decode_batch_dev_hint": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint8, C.c_int, _vp, C.c_size_t, C.POINTER(BatchResult), C.POINTER(C.c_uint64), C.c_size_t, _vp]),
    "fqz_decode_batch_launch": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint8, C.c_int, _vp, C.c_size_t, _vp]),
    "fqz_decode_batch_finish": (C.c_int, [_vp, C.POINTER(BatchResult)]),
    "fqz_debug_get_streams": (C.c_int, [_vp, C.c_uint32, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "fqz_debug_get_stamps": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]),
    "fqz_pack_bases": (C.c_int, [_vp, C.c_char_p, C.c_size_t, _vp, _vp, C.POINTER(C.c_size_t)]),
    "fqz_unpack_bases": (C.c_int, [_vp, C.c_char_p, _vp, C.c_size_t, C.c_size_t, _vp]),
    "fqz_detect_encoding": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_int)]),
    "fqz_normalize_quality": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int]),
    "fqz_denormalize_quality": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int]),
    "fqz_delta_encode": (C.c_int, [_vp, _vp, C.c_size_t]),
    "fqz_delta_decode": (C.c_int, [_vp, _vp, C.c_size_t]),
    "fqz_entropy_bound": (C.c_size_t, [C.c_size_t]),
    "fqz_entropy_encode": (C.c_int, [_vp, C.c_char_p, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "fqz_entropy_decode": (C.c_int, [_vp, C.c_char_p, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "fqz_compress": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Options)]),
    "fqz_decompress": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(DecompressOptions)]),
    "fqz_compress_stream": (C.c_int, [_vp, READ_FN, _vp, WRITE_FN, _vp, C.POINTER(Options)]),
    "fqz_decompress_stream": (C.c_int, [_vp, READ_FN, _vp, WRITE_FN, _vp, C.POINTER(DecompressOptions)]),
    "fqz_decompress_alloc": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(_vp), C.POINTER(C.c_size_t), C.POINTER(DecompressOptions)]),
    "fqz_buffer_free": (None, [_vp]),
    "fqz_compress_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Options)]),
    "fqz_decompress_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, _vp, C.c_size_t, C.POINTER(_vp), C.POINTER(C.c_size_t), C.POINTER(DecompressOptions)]),
    "fqz_read_block_table": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t)]),
    "fqz_compress_file": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.POINTER(Options)]),
    "fqz_decompress_file": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.POINTER(DecompressOptions)]),
    "fqz_profile_enable": (C.c_int, [_vp, C.c_int]),
    "fqz_profile_reset": (C.c_int, [_vp]),
    "fqz_profile_read": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t)]),
    "fqz_synth_fastq": (C.c_int, [C.POINTER(SynthParams), C.c_uint64, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
}


def lib():
    """The loaded library; raises loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError("libfqzhip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C fastqpacker_amd/csrc`; there is no CPU fallback" % _SO)
        L = C.CDLL(_SO)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc, detail=""):
    if rc != 0:
        raise FqzError(rc, detail)


class Ctx:
    """fqz_ctx: one per worker, like the per-worker zstd encoder it replaces (compress.go:281)."""

    def __init__(self, device=0):
        self._h = _vp()
        check(lib().fqz_ctx_create(device, C.byref(self._h)))
        self.device = device

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            lib().fqz_ctx_destroy(self._h)
            self._h = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ctx_profile(self, on=True):
    """False/0 off, True/1 every kernel, 2 only the dominant encode kernel (k_entropy)"""
    check(lib().fqz_profile_enable(self._h, int(on)))
    check(lib().fqz_profile_reset(self._h))


def _ctx_profile_read(self):
    """{kernel: (total_ms, calls)} accumulated since the last profile(True)/reset."""
    names = C.create_string_buffer(4096)
    ms = (C.c_double * 64)()
    calls = (C.c_uint32 * 64)()
    n = C.c_size_t(0)
    check(lib().fqz_profile_read(self._h, names, 4096, ms, calls, 64, C.byref(n)))
    keys = names.value.decode().split("\n") if n.value else []
    return {k: (ms[i], calls[i]) for i, k in enumerate(keys)}


Ctx.profile = _ctx_profile
Ctx.profile_read = _ctx_profile_read

_default = {}


def default_ctx(device=0):
    if device not in _default:
        _default[device] = Ctx(device)
    return _default[device]
