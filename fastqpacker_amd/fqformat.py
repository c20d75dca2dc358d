"""Mirror of internal/fqformat/container.go (byte-exact framing) over libfqzhip."""
import ctypes as C

from ._lib import lib, check, FileHeader, BlockHeader, FqzError

Magic = b"FQZ\x00"       # container.go:11
FlagPairedEnd = 1 << 0   # container.go:15
FlagPhred64 = 1 << 1     # container.go:16
Version1 = 1             # container.go:21
Version2 = 2             # container.go:22
CurrentVersion = Version2


def WriteFileHeader(version, block_size, flags) -> bytes:
    h = FileHeader(version, block_size, flags)
    out = (C.c_uint8 * 10)()
    lib().fqz_write_file_header(C.byref(h), out)
    return bytes(out)


def ReadFileHeader(data: bytes):
    h = FileHeader()
    check(lib().fqz_read_file_header(data, len(data), C.byref(h)))
    return h.version, h.block_size, h.flags


def WriteBlockHeader(fields, version) -> bytes:
    b = BlockHeader(*fields)
    out = (C.c_uint8 * 36)()
    n = lib().fqz_write_block_header(C.byref(b), version, out)
    if n < 0:
        raise FqzError(n)
    return bytes(out[:n])


def ReadBlockHeader(data: bytes, version):
    b = BlockHeader()
    n = lib().fqz_read_block_header(data, len(data), version, C.byref(b))
    if n < 0:
        raise FqzError(n)
    return [getattr(b, f) for f, _ in BlockHeader._fields_], n
