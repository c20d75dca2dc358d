"""The streaming host pipeline (compress.Compress / compress.Decompress over readers and writers, compress.go:125-192,
558-604): many batches through small slices, blocks that do not fit a slice (long reads), file forms, bounded memory."""
import io
import os
import resource

import numpy as np
import pytest

import oracle_lib as O
from fastq_gen import make_fastq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fq():
    import fastqpacker_amd as fq
    fq.lib()
    return fq


@pytest.fixture()
def small_slices(monkeypatch):
    monkeypatch.setenv("FQZ_SLICE_KB", "256")   # 256 KiB slices: a 100 000-record block spans many of them
    yield


def test_many_slices_give_the_same_file(fq, small_slices):
    text = make_fastq(120_000, seed=31, min_len=60, max_len=120, n_frac=0.01)   # > 1 block of 100 000 records, ~100 slices
    want = O.compress(text, workers=8)
    assert fq.compress.Compress(text) == want                                   # memory -> memory through the three-slot pipeline
    src, dst = io.BytesIO(text), io.BytesIO()
    fq.compress.CompressStream(src, dst)                                        # reader -> writer
    assert dst.getvalue() == want
    back = io.BytesIO()
    fq.compress.DecompressStream(io.BytesIO(want), back)
    assert back.getvalue() == text
    assert fq.compress.Decompress(want) == text


def test_blocks_larger_than_a_slice_long_reads(fq, small_slices):
    """records of several KB - the first 100 000 records do not fit one slice; the batch grows instead of
    failing with FQZ_E_TOO_LARGE."""
    rng = np.random.default_rng(3)
    recs = []
    for i in range(300):
        L = int(rng.integers(3000, 9000))
        seq = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
        qual = bytes(rng.integers(35, 70, L).astype(np.uint8))
        recs.append(b"@ont_read_%d ch=%d\n%s\n+\n%s\n" % (i, i % 512, seq, qual))
    text = b"".join(recs)
    assert len(text) > 6 * (256 << 10)
    z = fq.compress.Compress(text)
    assert z == O.compress(text)
    assert fq.compress.Decompress(z) == text


def test_file_forms_stream_and_errors(fq, tmp_path, small_slices):
    import ctypes as C
    from fastqpacker_amd._lib import lib, check, default_ctx
    text = make_fastq(30_000, seed=9, min_len=80, max_len=151, n_frac=0.02, phred=64)
    a, b, c = tmp_path / "in.fq", tmp_path / "out.fqz", tmp_path / "back.fq"
    a.write_bytes(text)
    ctx = default_ctx()
    check(lib().fqz_compress_file(ctx.handle, str(a).encode(), str(b).encode(), None))
    assert b.read_bytes() == O.compress(text)
    check(lib().fqz_decompress_file(ctx.handle, str(b).encode(), str(c).encode(), None))
    assert c.read_bytes() == text
    # size-only decompress (out == NULL) and a too-small destination
    z = b.read_bytes()
    zin = np.frombuffer(z, dtype=np.uint8)
    n = C.c_size_t(0)
    check(lib().fqz_decompress(ctx.handle, zin.ctypes.data, zin.size, None, 0, C.byref(n), None))
    assert n.value == len(text)
    small = np.empty(len(text) - 1, dtype=np.uint8)
    assert lib().fqz_decompress(ctx.handle, zin.ctypes.data, zin.size, small.ctypes.data, small.size, C.byref(n), None) == -18  # FQZ_E_DST_SMALL
    # a parser error in a late slice still surfaces; partial output is the caller's business (compress.go:165, App. B-9)
    bad = text[:2_000_000] + b"oops\n" + text[2_000_000:]
    with pytest.raises(fq.FqzError):
        fq.compress.Compress(bad)
    # truncated containers
    with pytest.raises(fq.FqzError, match="unexpected EOF"):
        fq.compress.Decompress(z[:-5])
    # a reader that fails
    class Broken(io.RawIOBase):
        def read(self, n=-1):
            raise OSError("disk on fire")
    with pytest.raises(OSError):
        fq.compress.CompressStream(Broken(), io.BytesIO())
    # the context still works
    assert fq.compress.Decompress(fq.compress.Compress(text)) == text


def test_decode_block_size_does_not_assemble(fq):
    text = make_fastq(5000, seed=4, min_len=100, max_len=151)
    block, nrec = fq.compress.encode_block(text)
    import ctypes as C
    from fastqpacker_amd._lib import lib, check, default_ctx
    n = C.c_size_t(0)
    a = np.frombuffer(block, dtype=np.uint8)
    check(lib().fqz_decode_block_size(default_ctx().handle, a.ctypes.data, a.size, 2, C.byref(n)))
    assert n.value == len(text) and nrec == 5000


def test_memory_stays_bounded_for_a_large_stream(fq, tmp_path):
    """A 1.2 GB file through fqz_compress_file / fqz_decompress_file: the process' peak RSS grows by far less than the file."""
    import ctypes as C
    from fastqpacker_amd import compress
    from fastqpacker_amd._lib import lib, check, default_ctx
    ctx = default_ctx()
    text, n = compress.synth_fastq(3_400_000)
    src = tmp_path / "big.fq"
    text.tofile(str(src))
    size = text.size
    del text
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    check(lib().fqz_compress_file(ctx.handle, str(src).encode(), str(tmp_path / "big.fqz").encode(), None))
    check(lib().fqz_decompress_file(ctx.handle, str(tmp_path / "big.fqz").encode(), str(tmp_path / "big.out").encode(), None))
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert os.path.getsize(tmp_path / "big.out") == size
    # byte compare in pieces
    with open(src, "rb") as f, open(tmp_path / "big.out", "rb") as g:
        while True:
            x, y = f.read(1 << 26), g.read(1 << 26)
            assert x == y
            if not x:
                break
    grown_mb = max(0, after - before) / 1024
    assert grown_mb < 0.5 * size / 1e6, "peak RSS grew by %.0f MB for a %.0f MB file" % (grown_mb, size / 1e6)


def test_compress_multi_is_identical_to_one_device(fq):
    """fqz_compress_multi with the one device of the box listed two and three times: every 'device' encodes a contiguous
    range of whole 100 000-record blocks (cut by line count), the file is the same as the single-device one."""
    text, n = fq.compress.synth_fastq(450_017, min_len=35, max_len=90, n_permille=20)   # 5 blocks, the last one partial
    text = text.tobytes()
    want = fq.compress.Compress(text)
    for devs in ([0], [0, 0], [0, 0, 0], [0] * 7):                                    # (7: more shards than blocks -> empty shards)
        assert fq.compress.CompressMulti(text, devs) == want, devs
    small = text[: 2_000_000]                                                           # less than one block: the first shard takes it all
    small = small[: small.rfind(b"\n@") + 1]
    assert fq.compress.CompressMulti(small, [0, 0]) == fq.compress.Compress(small)
    assert fq.compress.CompressMulti(b"", [0, 0]) == fq.compress.Compress(b"")
    for devs in ([0], [0, 0], [0, 0, 0], [0] * 7):                                    # the way back: ranges of whole blocks per device
        assert fq.compress.DecompressMulti(want, devs) == text, devs
    assert fq.compress.DecompressMulti(fq.compress.Compress(b""), [0, 0]) == b""
    with pytest.raises(Exception):
        fq.compress.DecompressMulti(want[:-5], [0, 0])                                  # "reading block data: unexpected EOF"
    # Phred+64 decided by the first shard, applied by the others
    t64, _ = fq.compress.synth_fastq(250_000, min_len=40, max_len=60, phred=64)
    t64 = t64.tobytes()
    w64 = fq.compress.Compress(t64)
    assert w64[9] & 2 and fq.compress.CompressMulti(t64, [0, 0]) == w64
    assert fq.compress.DecompressMulti(w64, [0, 0]) == t64
    # a parse error in a later shard is reported
    bad = bytearray(text)
    at = text.rfind(b"\n+\n")
    bad[at + 1] = ord("-")
    with pytest.raises(Exception):
        fq.compress.CompressMulti(bytes(bad), [0, 0])


def test_compress_multi_range_boundary_on_a_block_boundary(fq):
    """the byte range of device 1 starts exactly where block 1 starts (the lines in front of it are a non-zero multiple
    of 400 000): the newline that ends block 0 lies in the EARLIER range.  Records of 2048 bytes: 100 000 of them fill
    3125 counting tiles of 64 KiB exactly, so two devices cut the 200 000-record text at that very byte."""
    rec = b"@hh\n" + b"ACGT" * 255 + b"\n+\n" + b"I" * 1020 + b"\n"
    assert len(rec) == 2048
    text = rec * 200_000
    want = fq.compress.Compress(text)
    assert fq.compress.CompressMulti(text, [0, 0]) == want
    assert fq.compress.DecompressMulti(want, [0, 0]) == text


def test_compress_multi_small_phred64_with_an_empty_first_shard(fq):
    """a text smaller than two counting tiles leaves the first device's range empty; the shard that holds block 0 (the
    second) must detect the encoding and write the header with FlagPhred64, as fqz_compress does."""
    t64, _ = fq.compress.synth_fastq(400, min_len=40, max_len=60, phred=64)
    t64 = t64.tobytes()
    assert len(t64) < 2 * 65536
    w64 = fq.compress.Compress(t64)
    assert w64[9] & 2
    assert fq.compress.CompressMulti(t64, [0, 0]) == w64
    assert fq.compress.DecompressMulti(w64, [0, 0]) == t64


def test_empty_v3_file_with_block_table_through_the_stream_reader(fq):
    """empty input, container version 3 with a block table = file header + a 24-byte table and no block; the callback
    reader must accept it like the memory path does."""
    z = fq.compress.Compress(b"", fq.Options(0, 0, 3, 1))
    assert len(z) == 10 + 24
    out = io.BytesIO()
    fq.compress.DecompressStream(io.BytesIO(z), out)
    assert out.getvalue() == b"" and fq.compress.Decompress(z) == b""
