"""Mirror of internal/compress (Compress / Decompress / Options) plus the per-block and
device-resident batch entry points of libfqzhip.  All codec work runs on the GPU."""
import ctypes as C

import numpy as np

from ._lib import (lib, check, default_ctx, BatchResult, Options, DecompressOptions, FqzError, DEFAULT_BLOCK_SIZE,
                   DETECT_ENCODING, BATCH_FINAL, BATCH_V3, BATCH_SEG, BATCH_HALVES, SynthParams)

DefaultBlockSize = DEFAULT_BLOCK_SIZE  # compress.go:71


def _as_u8(buf):
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf, dtype=np.uint8)
    return np.frombuffer(buf, dtype=np.uint8)


def Compress(fastq, opts: Options = None, ctx=None) -> bytes:
    """compress.Compress (compress.go:125): FASTQ bytes -> .fqz bytes."""
    ctx = ctx or default_ctx()
    a = _as_u8(fastq)
    cap = lib().fqz_encode_bound(a.size) + 10
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    check(lib().fqz_compress(ctx.handle, a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap, C.byref(n),
                             C.byref(opts) if opts is not None else None))
    return out[: n.value].tobytes()


def CompressMulti(fastq, devices, opts: Options = None) -> bytes:
    """compress.Compress with the reference's worker pool spread over several devices (compress.go:240-278): one host
    thread + context per entry of `devices`, contiguous ranges of whole blocks, output identical to Compress."""
    a = _as_u8(fastq)
    cap = lib().fqz_encode_bound(a.size) + 10
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    devs = (C.c_int * len(devices))(*devices)
    check(lib().fqz_compress_multi(devs, len(devices), a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap, C.byref(n),
                                   C.byref(opts) if opts is not None else None))
    return out[: n.value].tobytes()


def DecompressMulti(fqz, devices, opts: DecompressOptions = None) -> bytes:
    """compress.Decompress with the reference's worker pool spread over several devices (compress.go:630-668): contiguous
    ranges of whole blocks per device, texts concatenated in order; identical to Decompress."""
    a = _as_u8(fqz)
    n = C.c_size_t(0)
    p = C.c_void_p()
    devs = (C.c_int * len(devices))(*devices)
    check(lib().fqz_decompress_multi(devs, len(devices), a.ctypes.data if a.size else None, a.size, C.byref(p), C.byref(n),
                                     C.byref(opts) if opts is not None else None))
    try:
        return C.string_at(p, n.value)
    finally:
        lib().fqz_buffer_free(p)


def Decompress(fqz, opts: DecompressOptions = None, ctx=None) -> bytes:
    """compress.Decompress (compress.go:558): .fqz bytes -> FASTQ bytes (one decode; the library allocates the text)."""
    ctx = ctx or default_ctx()
    a = _as_u8(fqz)
    n = C.c_size_t(0)
    p = C.c_void_p()
    o = C.byref(opts) if opts is not None else None
    check(lib().fqz_decompress_alloc(ctx.handle, a.ctypes.data if a.size else None, a.size, C.byref(p), C.byref(n), o))
    try:
        return C.string_at(p, n.value)
    finally:
        lib().fqz_buffer_free(p)


def CompressStream(reader, writer, opts: Options = None, ctx=None):
    """compress.Compress(io.Reader, io.Writer, *Options) (compress.go:125) over file-like objects: streams, memory bounded."""
    ctx = ctx or default_ctx()
    rd, wr, err = _callbacks(reader, writer)
    rc = lib().fqz_compress_stream(ctx.handle, rd, None, wr, None, C.byref(opts) if opts is not None else None)
    if err:
        raise err[0]
    check(rc)


def DecompressStream(reader, writer, opts: DecompressOptions = None, ctx=None):
    """compress.Decompress(io.Reader, io.Writer, *DecompressOptions) (compress.go:558) over file-like objects."""
    ctx = ctx or default_ctx()
    rd, wr, err = _callbacks(reader, writer)
    rc = lib().fqz_decompress_stream(ctx.handle, rd, None, wr, None, C.byref(opts) if opts is not None else None)
    if err:
        raise err[0]
    check(rc)


def _callbacks(reader, writer):
    from ._lib import READ_FN, WRITE_FN
    err = []

    def _rd(_user, dst, cap):
        try:
            b = reader.read(cap)
            C.memmove(dst, b, len(b))
            return len(b)
        except Exception as e:  # noqa: BLE001 - reported to the caller after the C call returns
            err.append(e)
            return -1

    def _wr(_user, src, n):
        try:
            writer.write(C.string_at(src, n))
            return 0
        except Exception as e:  # noqa: BLE001
            err.append(e)
            return 1
    return READ_FN(_rd), WRITE_FN(_wr), err


def encode_block(fastq, qual_encoding=0, ctx=None):
    """One call per block = compressBlockWithBuffers (compress.go:471). Returns (block bytes, n_records)."""
    ctx = ctx or default_ctx()
    a = _as_u8(fastq)
    cap = lib().fqz_encode_bound(a.size)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    nrec = C.c_uint32(0)
    check(lib().fqz_encode_block(ctx.handle, a.ctypes.data if a.size else None, a.size, qual_encoding, out.ctypes.data, cap,
                                 C.byref(n), C.byref(nrec)))
    return out[: n.value].tobytes(), nrec.value


def read_block_table(fqz):
    """[(offset of the block header in the file, records)] from the block table of a version-3 file written with
    Options.block_index = 1 (include/fqz.h); raises FqzError if the file carries none.  Host only."""
    a = _as_u8(fqz)
    n = C.c_size_t(0)
    check(lib().fqz_read_block_table(a.ctypes.data, a.size, None, None, 0, C.byref(n)))
    off, rec = (C.c_uint64 * max(1, n.value))(), (C.c_uint32 * max(1, n.value))()
    check(lib().fqz_read_block_table(a.ctypes.data, a.size, off, rec, n.value, C.byref(n)))
    return [(int(off[i]), int(rec[i])) for i in range(n.value)]


def decode_block(block, version=2, qual_encoding=0, ctx=None) -> bytes:
    """decompressJobToPooledBuffer (compress.go:780): block header + payloads -> FASTQ text."""
    ctx = ctx or default_ctx()
    a = _as_u8(block)
    n = C.c_size_t(0)
    check(lib().fqz_decode_block_size(ctx.handle, a.ctypes.data, a.size, version, C.byref(n)))
    out = np.empty(max(n.value, 1), dtype=np.uint8)
    check(lib().fqz_decode_block(ctx.handle, a.ctypes.data, a.size, version, qual_encoding, out.ctypes.data, n.value, C.byref(n)))
    return out[: n.value].tobytes()


def get_streams(block=0, ctx=None):
    """Test hook: the six pre-entropy streams of `block` of the last encode call."""
    ctx = ctx or default_ctx()
    lens = (C.c_size_t * 6)(*([0] * 6))
    check(lib().fqz_debug_get_streams(ctx.handle, block, None, lens))
    bufs = [C.create_string_buffer(max(1, lens[k])) for k in range(6)]
    ptrs = (C.c_void_p * 6)(*[C.cast(b, C.c_void_p) for b in bufs])
    caps = (C.c_size_t * 6)(*[max(1, lens[k]) for k in range(6)])
    check(lib().fqz_debug_get_streams(ctx.handle, block, ptrs, caps))
    return [bufs[k].raw[: caps[k]] for k in range(6)]


def entropy_encode(data: bytes, ctx=None) -> bytes:
    ctx = ctx or default_ctx()
    cap = lib().fqz_entropy_bound(len(data)) + 16
    out = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    check(lib().fqz_entropy_encode(ctx.handle, data, len(data), out, cap, C.byref(n)))
    return out.raw[: n.value]


def entropy_decode(frame: bytes, cap: int, ctx=None) -> bytes:
    ctx = ctx or default_ctx()
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t(0)
    check(lib().fqz_entropy_decode(ctx.handle, frame, len(frame), out, cap, C.byref(n)))
    return out.raw[: n.value]


def encode_batch_dev(d_fastq_ptr, n_bytes, d_out_ptr, out_cap, records_per_block=DEFAULT_BLOCK_SIZE,
                     qual_encoding=DETECT_ENCODING, final=True, stream=None, ctx=None, max_blocks=0, container_version=2, segments=False):
    """Device-resident batch encode; pointers are raw device addresses (e.g. torch tensor.data_ptr()).  segments: the experimental
    FQZ-S1 framing (FQZ_BATCH_SEG)."""
    ctx = ctx or default_ctx()
    res = BatchResult()
    offs = (C.c_uint64 * max_blocks)() if max_blocks else None
    lens = (C.c_uint64 * max_blocks)() if max_blocks else None
    rc = lib().fqz_encode_batch_dev(ctx.handle, d_fastq_ptr, n_bytes, records_per_block, qual_encoding,
                                    (BATCH_FINAL if final else 0) | (BATCH_V3 if container_version == 3 else 0) | (BATCH_SEG if segments else 0),
                                    d_out_ptr, out_cap, C.byref(res), offs, lens, max_blocks, stream)
    if rc:
        raise FqzError(rc, "record %d" % res.error_record if res.status and -8 <= res.status <= -5 else "")
    if max_blocks:
        return res, list(offs[: res.n_blocks]), list(lens[: res.n_blocks])
    return res


def decode_batch_dev(d_blocks_ptr, n_bytes, d_out_ptr, out_cap, version=2, qual_encoding=0, stream=None, ctx=None, block_off=None):
    """block_off: offsets of the block headers inside the batch (what encode_batch_dev(max_blocks=...) returns): a hint that saves
    the device's walk along the chain of block headers (fqz_decode_batch_dev_hint)."""
    ctx = ctx or default_ctx()
    res = BatchResult()
    if block_off is not None and len(block_off):
        arr = (C.c_uint64 * len(block_off))(*block_off)
        check(lib().fqz_decode_batch_dev_hint(ctx.handle, d_blocks_ptr, n_bytes, version, qual_encoding, d_out_ptr, out_cap, C.byref(res), arr, len(block_off), stream))
    else:
        check(lib().fqz_decode_batch_dev(ctx.handle, d_blocks_ptr, n_bytes, version, qual_encoding, d_out_ptr, out_cap, C.byref(res), stream))
    return res


def synth_fastq(n_records, seed=0xF0057A57, first_record=0, min_len=150, max_len=150, n_permille=0, phred=33, quality_profile=0,
                cap=None):
    """Deterministic synthetic FASTQ as a numpy uint8 array. Host generator."""
    p = SynthParams(seed, first_record, min_len, max_len, n_permille, phred, quality_profile)
    if cap is None:
        cap = int(n_records) * (2 * max_len + 80) + 4096
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    wrote = C.c_uint64(0)
    check(lib().fqz_synth_fastq(C.byref(p), n_records, out.ctypes.data, cap, C.byref(n), C.byref(wrote)))
    return out[: n.value], wrote.value
