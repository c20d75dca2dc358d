"""ctypes binding of the CPU oracle (oracle/libfqz_oracle.so) — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libfqz_oracle.so")

NSTREAMS = 6
S_SEQ, S_QUAL, S_HEADERS, S_PLUS, S_NPOS, S_LENGTHS = range(6)
STREAM_NAMES = ["seq", "qual", "headers", "plus", "npos", "lengths"]
CHUNK = 16384


class Record(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("hdr_off", "hdr_len", "seq_off", "seq_len", "plus_off", "plus_len", "qual_off", "qual_len")]


class FileHeader(C.Structure):
    _fields_ = [("version", C.c_uint8), ("block_size", C.c_uint32), ("flags", C.c_uint8)]


class BlockHeader(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("num_records", "seq_size", "qual_size", "header_size", "plus_size", "npos_size",
                 "lengths_size", "original_seq_size", "original_qual_size")]


class Streams(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8) * NSTREAMS), ("len", C.c_size_t * NSTREAMS),
                ("original_seq_size", C.c_uint32), ("original_qual_size", C.c_uint32)]


class Options(C.Structure):
    _fields_ = [("block_size", C.c_uint32), ("workers", C.c_int), ("batch_records", C.c_uint32), ("entropy", C.c_int),
                ("force_encoding", C.c_int), ("block_index", C.c_int), ("framing", C.c_int)]


def build():
    src = [os.path.join(_ROOT, "oracle", f) for f in ("fqz_oracle.c", "fqz_entropy.c", "fqz_oracle.h")]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p = C.POINTER(C.c_uint8)
        L.fqzo_pack_bases.restype = C.c_size_t
        L.fqzo_pack_bases.argtypes = [C.c_char_p, C.c_size_t, u8p, C.POINTER(C.c_uint16)]
        L.fqzo_unpack_bases.restype = C.c_int
        L.fqzo_unpack_bases.argtypes = [u8p, C.POINTER(C.c_uint16), C.c_size_t, C.c_size_t, u8p]
        L.fqzo_detect_encoding.restype = C.c_int
        L.fqzo_detect_encoding.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t]
        for f in ("fqzo_normalize_quality", "fqzo_denormalize_quality"):
            getattr(L, f).restype = None
            getattr(L, f).argtypes = [u8p, C.c_size_t, C.c_int]
        for f in ("fqzo_delta_encode", "fqzo_delta_decode"):
            getattr(L, f).restype = None
            getattr(L, f).argtypes = [u8p, C.c_size_t]
        L.fqzo_write_file_header.restype = None
        L.fqzo_write_file_header.argtypes = [C.POINTER(FileHeader), u8p]
        L.fqzo_read_file_header.restype = C.c_int
        L.fqzo_read_file_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FileHeader)]
        L.fqzo_write_block_header.restype = C.c_int
        L.fqzo_write_block_header.argtypes = [C.POINTER(BlockHeader), C.c_uint8, u8p]
        L.fqzo_read_block_header.restype = C.c_int
        L.fqzo_read_block_header.argtypes = [C.c_char_p, C.c_size_t, C.c_uint8, C.POINTER(BlockHeader)]
        L.fqzo_strerror.restype = C.c_char_p
        L.fqzo_strerror.argtypes = [C.c_int]
        L.fqzo_parse_batch.restype = C.c_long
        L.fqzo_parse_batch.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Record), C.c_size_t,
                                       C.POINTER(C.c_int)]
        L.fqzo_split_block.restype = C.c_int
        L.fqzo_split_block.argtypes = [C.c_char_p, C.POINTER(Record), C.c_size_t, C.c_int, C.POINTER(Streams)]
        L.fqzo_streams_free.restype = None
        L.fqzo_streams_free.argtypes = [C.POINTER(Streams)]
        L.fqzo_join_block.restype = C.c_long
        L.fqzo_join_block.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_int, u8p, C.c_size_t]
        L.fqzo_entropy_bound.restype = C.c_size_t
        L.fqzo_entropy_bound.argtypes = [C.c_size_t]
        L.fqzo_entropy_encode.restype = C.c_size_t
        L.fqzo_entropy_encode.argtypes = [C.c_char_p, C.c_size_t, u8p]
        L.fqzo_entropy_encode_stream.restype = C.c_size_t
        L.fqzo_entropy_encode_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_int, u8p]
        L.fqzo_entropy_encode_stream_v.restype = C.c_size_t
        L.fqzo_entropy_encode_stream_v.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, u8p]
        L.fqzo_xxh64.restype = C.c_uint64
        L.fqzo_xxh64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.fqzo_entropy_decode.restype = C.c_long
        L.fqzo_entropy_decode.argtypes = [C.c_char_p, C.c_size_t, u8p, C.c_size_t]
        L.fqzo_entropy_content_size.restype = C.c_long
        L.fqzo_entropy_content_size.argtypes = [C.c_char_p, C.c_size_t]
        L.fqzo_huf_code_lengths.restype = C.c_int
        L.fqzo_huf_code_lengths.argtypes = [C.POINTER(C.c_uint32), u8p]
        L.fqzo_huf_codes.restype = None
        L.fqzo_huf_codes.argtypes = [u8p, C.c_int, C.POINTER(C.c_uint16)]
        L.fqzo_huf_write_tree.restype = C.c_size_t
        L.fqzo_huf_write_tree.argtypes = [u8p, C.c_int, u8p]
        L.fqzo_encode_chunk.restype = C.c_size_t
        L.fqzo_encode_chunk.argtypes = [C.c_char_p, C.c_size_t, C.c_int, u8p]
        L.fqzo_compress_bound.restype = C.c_size_t
        L.fqzo_compress_bound.argtypes = [C.c_size_t]
        L.fqzo_compress.restype = C.c_long
        L.fqzo_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(Options)]
        L.fqzo_decompress.restype = C.c_long
        L.fqzo_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.fqzo_libzstd_version.restype = C.c_uint
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        self.code = code
        super().__init__(lib().fqzo_strerror(code).decode())


def _u8(buf):
    return (C.c_uint8 * len(buf)).from_buffer(buf)


def pack_bases(seq: bytes):
    n = len(seq)
    packed = bytearray((n + 3) // 4)
    npos = (C.c_uint16 * max(1, min(n, 65536)))()
    k = lib().fqzo_pack_bases(seq, n, _u8(packed) if packed else None, npos)
    return bytes(packed), list(npos[:k])


def unpack_bases(packed: bytes, npos, seq_len: int):
    out = bytearray(seq_len)
    pk = bytearray(packed) or bytearray(1)
    arr = (C.c_uint16 * max(1, len(npos)))(*npos)
    r = lib().fqzo_unpack_bases(_u8(pk), arr, len(npos), seq_len, _u8(out) if out else None)
    if r:
        raise OracleError(-20)
    return bytes(out)


def detect_encoding(quals):
    n = len(quals)
    arr = (C.c_char_p * max(1, n))(*quals)
    lens = (C.c_size_t * max(1, n))(*[len(q) for q in quals])
    return lib().fqzo_detect_encoding(arr, lens, n)


def _inplace(fn, data: bytes, *args):
    buf = bytearray(data)
    if buf:
        fn(_u8(buf), len(buf), *args)
    return bytes(buf)


def normalize_quality(q, enc): return _inplace(lib().fqzo_normalize_quality, q, enc)
def denormalize_quality(q, enc): return _inplace(lib().fqzo_denormalize_quality, q, enc)
def delta_encode(q): return _inplace(lib().fqzo_delta_encode, q)
def delta_decode(q): return _inplace(lib().fqzo_delta_decode, q)


def parse_all(text: bytes, batch=100000):
    """All records of text as a ctypes array (reference ReadBatch semantics)."""
    cap = len(text) // 4 + 1
    recs = (Record * cap)()
    pos = C.c_size_t(0)
    eof = C.c_int(0)
    n = 0
    while True:
        want = min(batch, cap - n)
        if want == 0:
            break
        got = lib().fqzo_parse_batch(text, len(text), C.byref(pos), C.cast(C.byref(recs, n * C.sizeof(Record)), C.POINTER(Record)),
                                     want, C.byref(eof))
        if got < 0:
            raise OracleError(got)
        n += got
        if eof.value or got == 0:
            break
    return recs, n


def split_block(text: bytes, recs, n_rec, enc, first=0):
    s = Streams()
    p = C.cast(C.byref(recs, first * C.sizeof(Record)), C.POINTER(Record))
    r = lib().fqzo_split_block(text, p, n_rec, enc, C.byref(s))
    if r:
        raise OracleError(r)
    out = [C.string_at(s.data[k], s.len[k]) for k in range(NSTREAMS)]
    orig = (s.original_seq_size, s.original_qual_size)
    lib().fqzo_streams_free(C.byref(s))
    return out, orig


def join_block(streams, num_records, enc, cap=None):
    datas = (C.c_char_p * NSTREAMS)(*[s if s is not None else None for s in streams])
    lens = (C.c_size_t * NSTREAMS)(*[len(s) if s is not None else 0 for s in streams])
    if cap is None:
        cap = sum(lens) * 5 + 16 * num_records + 64
    out = bytearray(cap)
    r = lib().fqzo_join_block(datas, lens, num_records, enc, _u8(out), cap)
    if r < 0:
        raise OracleError(r)
    return bytes(out[:r])


def entropy_encode(src: bytes, stream: int = 1, version: int = 2) -> bytes:
    """FQZ-H2 payload of one pre-entropy stream (stream 0 = 2-bit packed bases: Raw blocks by definition); version 3: FQZ-R1
    (the quality stream, stream 1, in interleaved rANS blocks)."""
    cap = lib().fqzo_entropy_bound(len(src)) + 16
    out = bytearray(cap)
    n = lib().fqzo_entropy_encode_stream_v(src, len(src), stream, version, _u8(out))
    return bytes(out[:n])


def xxh64(data: bytes, seed: int = 0) -> int:
    return lib().fqzo_xxh64(data, len(data), seed)


def payload_frames(payload: bytes):
    """Splits an FQZ-H2 payload into (index frame or None, [zstd frames]) by walking the frame / block headers."""
    idx, frames, ip, n = None, [], 0, len(payload)
    while ip < n:
        if payload[ip + 1:ip + 4] == b"\x2a\x4d\x18" and (payload[ip] & 0xF0) == 0x50:
            sz = int.from_bytes(payload[ip + 4:ip + 8], "little")
            idx = payload[ip:ip + 8 + sz]
            ip += 8 + sz
            continue
        assert payload[ip:ip + 4] == bytes.fromhex("28b52ffd")
        fhd = payload[ip + 4]
        single, ck, fcs_flag = (fhd >> 5) & 1, (fhd >> 2) & 1, fhd >> 6
        q = ip + 5 + (0 if single else 1) + [single, 2, 4, 8][fcs_flag]
        while True:
            bh = int.from_bytes(payload[q:q + 3], "little")
            q += 3 + (1 if (bh >> 1) & 3 == 1 else bh >> 3)
            if bh & 1:
                break
        q += 4 * ck
        frames.append(payload[ip:q])
        ip = q
    return idx, frames


def entropy_decode(frame: bytes, cap=None) -> bytes:
    if cap is None:
        fcs = lib().fqzo_entropy_content_size(frame, len(frame))
        cap = fcs if fcs >= 0 else len(frame) * 64 + 65536
    out = bytearray(max(cap, 1))
    r = lib().fqzo_entropy_decode(frame, len(frame), _u8(out), cap)
    if r < 0:
        raise OracleError(r)
    return bytes(out[:r])


def encode_chunk(src: bytes, last=1) -> bytes:
    out = bytearray(len(src) + 64)
    n = lib().fqzo_encode_chunk(src, len(src), last, _u8(out))
    return bytes(out[:n])


def huf_code_lengths(counts):
    c = (C.c_uint32 * 256)(*counts)
    nb = bytearray(256)
    mx = lib().fqzo_huf_code_lengths(c, _u8(nb))
    return mx, bytes(nb)


def compress(fastq, block_size=0, workers=1, batch_records=0, entropy=0, force_encoding=0, block_index=0, framing=0) -> bytes:
    """compress.Compress on a memory buffer (bytes or numpy uint8 array).  framing: 0 = FQZ-H2 group framing (what the HIP encoder
    writes by default), 1 = FQZ-S1 segment framing for blocks that qualify (experimental; FQZ_BATCH_SEG / FQZ_ENC_SEG=1 there)."""
    a = np.frombuffer(fastq, dtype=np.uint8) if not isinstance(fastq, np.ndarray) else fastq
    cap = lib().fqzo_compress_bound(a.size)
    if batch_records:  # tiny blocks: 36-byte block headers and six frame headers per block dwarf the library's bound
        cap += (int(np.count_nonzero(a == 10)) // 4 // batch_records + 2) * 480
    out = np.empty(cap, dtype=np.uint8)
    opt = Options(block_size, workers, batch_records, entropy, force_encoding, block_index, framing)
    r = lib().fqzo_compress(a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap, C.byref(opt))
    if r < 0:
        raise OracleError(r)
    return out[:r].tobytes()


def decompress(fqz, workers=1) -> bytes:
    a = np.frombuffer(fqz, dtype=np.uint8) if not isinstance(fqz, np.ndarray) else fqz
    n = lib().fqzo_decompress(a.ctypes.data if a.size else None, a.size, None, 0, workers)
    if n < 0:
        raise OracleError(n)
    out = np.empty(max(n, 1), dtype=np.uint8)
    r = lib().fqzo_decompress(a.ctypes.data if a.size else None, a.size, out.ctypes.data, n, workers)
    if r < 0:
        raise OracleError(r)
    return out[:r].tobytes()


# ---- independent zstd decoder (system libzstd) --------------------------------
_zstd = None


def libzstd():
    global _zstd
    if _zstd is None:
        try:
            z = C.CDLL("libzstd.so.1")
        except OSError:
            _zstd = False
            return None
        z.ZSTD_decompress.restype = C.c_size_t
        z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        z.ZSTD_compress.restype = C.c_size_t
        z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
        z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]
        z.ZSTD_isError.restype = C.c_uint
        z.ZSTD_isError.argtypes = [C.c_size_t]
        z.ZSTD_getErrorName.restype = C.c_char_p
        z.ZSTD_getErrorName.argtypes = [C.c_size_t]
        _zstd = z
    return _zstd or None


def zstd_decompress(frame: bytes, cap: int) -> bytes:
    z = libzstd()
    out = C.create_string_buffer(max(cap, 1))
    r = z.ZSTD_decompress(out, cap, frame, len(frame))
    if z.ZSTD_isError(r):
        raise ValueError("libzstd: " + z.ZSTD_getErrorName(r).decode())
    return out.raw[:r]


def zstd_compress(data: bytes, level=1) -> bytes:
    z = libzstd()
    cap = z.ZSTD_compressBound(len(data))
    out = C.create_string_buffer(max(cap, 1))
    r = z.ZSTD_compress(out, cap, data, len(data), level)
    if z.ZSTD_isError(r):
        raise ValueError("libzstd: " + z.ZSTD_getErrorName(r).decode())
    return out.raw[:r]


def group_cases():
    """streams whose 16 KiB chunks differ inside a 64 KiB group (one Huffman table per group)"""
    rng = np.random.default_rng(23)
    skew = lambda n: bytes(np.minimum(rng.geometric(0.4, n) - 1, 40).astype(np.uint8))
    flat = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    C = 16384
    yield "rle-chunk-first", b"\x07" * C + skew(3 * C)                       # the tree travels with the second block
    yield "rle-chunk-in-the-middle", skew(C) + b"\x00" * C + skew(2 * C)
    yield "raw-chunk-inside-a-huffman-group", skew(C) + flat(C) + skew(2 * C)  # flat chunk: group table cannot shrink it
    yield "raw-chunk-first", flat(C) + skew(3 * C)                              # first Compressed block is the second chunk
    yield "two-groups-and-a-tail", skew(4 * C) + skew(4 * C) + skew(C + 77)
    yield "tail-group-of-tiny-chunk", skew(4 * C) + skew(5)
    yield "alternating-alphabets", b"".join((skew(C) if k % 2 else bytes(rng.choice([65, 67, 71, 84], C).astype(np.uint8))) for k in range(8))
