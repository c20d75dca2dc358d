"""Mirror of the reference's internal/encoder package over the HIP library.

Same names and argument meaning as internal/encoder/sequence.go and quality.go;
every function executes on the GPU through libfqzhip (no CPU implementation here).
"""
import ctypes as C

from ._lib import lib, check, default_ctx, ENCODING_PHRED33, ENCODING_PHRED64

MaxSequenceLength = 1 << 16   # sequence.go:11
Phred33Offset = 33            # quality.go:5
Phred64Offset = 64            # quality.go:6
EncodingPhred33 = ENCODING_PHRED33
EncodingPhred64 = ENCODING_PHRED64


def PackBases(seq: bytes, ctx=None):
    """sequence.go:58 — returns (packed, nPos); empty input returns (None, None)."""
    if len(seq) == 0:
        return None, None
    ctx = ctx or default_ctx()
    n = len(seq)
    packed = C.create_string_buffer((n + 3) // 4)
    npos = (C.c_uint16 * min(n, MaxSequenceLength))()
    cnt = C.c_size_t(0)
    check(lib().fqz_pack_bases(ctx.handle, seq, n, packed, npos, C.byref(cnt)))
    return packed.raw, list(npos[: cnt.value])


def AppendPackedBases(dst: bytearray, seq: bytes, nPos: list, ctx=None):
    """sequence.go:139 — appends to dst and to nPos, returns dst."""
    packed, pos = PackBases(seq, ctx)
    if packed is None:
        return dst
    dst += packed
    nPos += pos
    return dst


def UnpackBases(packed: bytes, nPos, seqLen: int, ctx=None):
    """sequence.go:103."""
    if seqLen == 0:
        return None
    ctx = ctx or default_ctx()
    out = C.create_string_buffer(seqLen)
    arr = (C.c_uint16 * max(1, len(nPos or [])))(*(nPos or []))
    check(lib().fqz_unpack_bases(ctx.handle, packed, arr, len(nPos or []), seqLen, out))
    return out.raw


def AppendUnpackBases(dst: bytearray, packed: bytes, nPos, seqLen: int, ctx=None):
    """sequence.go:188."""
    s = UnpackBases(packed, nPos, seqLen, ctx)
    if s is not None:
        dst += s
    return dst


def DetectEncoding(qualities, ctx=None):
    """quality.go:22."""
    ctx = ctx or default_ctx()
    flat = b"".join(qualities)
    offs = [0]
    for q in qualities:
        offs.append(offs[-1] + len(q))
    arr = (C.c_uint64 * len(offs))(*offs)
    enc = C.c_int(0)
    check(lib().fqz_detect_encoding(ctx.handle, flat, arr, len(qualities), C.byref(enc)))
    return enc.value


def _inplace(fn, qual: bytearray, *args, ctx=None):
    ctx = ctx or default_ctx()
    if len(qual):
        buf = (C.c_uint8 * len(qual)).from_buffer(qual)
        check(fn(ctx.handle, buf, len(qual), *args))
    return qual


def NormalizeQuality(qual: bytearray, enc, ctx=None):
    """quality.go:53 (in place on a bytearray)."""
    return _inplace(lib().fqz_normalize_quality, qual, enc, ctx=ctx)


def DenormalizeQuality(qual: bytearray, enc, ctx=None):
    """quality.go:66."""
    return _inplace(lib().fqz_denormalize_quality, qual, enc, ctx=ctx)


def DeltaEncode(qual: bytearray, ctx=None):
    """quality.go:81."""
    return _inplace(lib().fqz_delta_encode, qual, ctx=ctx)


def DeltaDecode(qual: bytearray, ctx=None):
    """quality.go:107."""
    return _inplace(lib().fqz_delta_decode, qual, ctx=ctx)
