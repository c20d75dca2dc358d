"""Container version 3 ("FQZ-R1"): a version-2 file whose quality payloads carry interleaved-rANS blocks
(oracle/fqz_entropy.c encode_group_rans / rans_decode_block, fastqpacker_amd/csrc/fqz_rans.h).  The reference has no version 3
(its decoder rejects the version byte, compress.go:571-573), so there is nothing of the reference's to pin the bytes to:
the oracle is the specification, the GPU must write the same bytes and both must read each other's files."""
import struct

import os

import numpy as np
import pytest

import oracle_lib as O
from fastq_gen import make_fastq


@pytest.fixture(scope="module")
def fq():
    import fastqpacker_amd as fq
    fq.lib()
    return fq


def _fastq(quals, seed=1, hdr=b"r%d"):
    """records with the given quality strings (bytes each); bases random"""
    rng = np.random.default_rng(seed)
    out = []
    for i, q in enumerate(quals):
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), len(q)).tobytes()
        out.append(b"@" + (hdr % i) + b"\n" + seq + b"\n+\n" + q + b"\n")
    return b"".join(out)


def _cases():
    rng = np.random.default_rng(11)
    yield "illumina 4 levels", make_fastq(4000, seed=41)
    yield "ragged + N + Phred64", make_fastq(3000, seed=42, min_len=35, max_len=301, n_frac=0.05, phred=64)
    yield "41 levels (little to gain)", make_fastq(2500, seed=43, qual_levels=list(range(2, 43)))
    yield "one record, one base", _fastq([b"I"])
    yield "tiny records", _fastq([bytes([int(v)]) * int(n) for v, n in zip(rng.integers(33, 74, 3000), rng.integers(1, 4, 3000))])
    yield "constant qualities (RLE chunks only)", _fastq([b"F" * 150] * 3000)
    # a long constant stretch (whole chunks of zero deltas) between noisy parts: RLE chunks inside coded groups
    yield "constant stretch between noisy parts", _fastq(
        [bytes(rng.choice([70, 58, 44], 150, p=[.9, .07, .03]).astype(np.uint8)) for _ in range(300)] + [b"F" * 20000] * 4 +
        [bytes(rng.choice([70, 58, 44], 150, p=[.9, .07, .03]).astype(np.uint8)) for _ in range(300)])
    yield "uniform random qualities (Raw groups)", _fastq([bytes(rng.integers(33, 127, 150, dtype=np.uint8)) for _ in range(1500)])
    yield "random then skewed (a chunk that coding expands inside a coded group)", _fastq(
        [b"F" * 150] * 150 + [bytes(rng.integers(33, 127, 150, dtype=np.uint8)) for _ in range(40)] + [bytes(rng.choice([70, 70, 70, 58], 150).astype(np.uint8)) for _ in range(400)])
    yield "long reads", _fastq([bytes(rng.choice([70, 58, 44, 35], int(n), p=[.85, .08, .05, .02]).astype(np.uint8)) for n in rng.integers(3000, 40000, 30)])
    yield "255 distinct deltas", _fastq([bytes((33 + (np.cumsum(rng.integers(0, 94, 150)) % 94)).astype(np.uint8)) for _ in range(800)] +
                                         [bytes(rng.choice([70, 58], 150, p=[.97, .03]).astype(np.uint8)) for _ in range(4000)])
    yield "empty", b""


def _v3(fq):
    return fq.Options(0, 0, 3)


def _qual_payload(z, block=0):
    """(offset, size) of the quality payload of the first block of a container"""
    assert block == 0
    sizes = struct.unpack_from("<9I", z, 10)
    return 10 + 36 + sizes[1], sizes[2]


def test_oracle_v3_round_trips_and_is_smaller_on_skewed_qualities():
    for name, text in _cases():
        z3 = O.compress(text, entropy=2)
        z2 = O.compress(text)
        assert z3[4] == 3 and z2[4] == 2, name
        assert O.decompress(z3) == text, name
        # everything but the quality payload is the version-2 payload
        if text:
            h2, h3 = struct.unpack_from("<9I", z2, 10), struct.unpack_from("<9I", z3, 10)
            assert h2[:2] + h2[3:] == h3[:2] + h3[3:], name
    text = make_fastq(20000, seed=44)
    z2, z3 = O.compress(text), O.compress(text, entropy=2)
    q2, q3 = struct.unpack_from("<9I", z2, 10)[2], struct.unpack_from("<9I", z3, 10)[2]
    assert q3 < 0.7 * q2, (q2, q3)  # order-0 entropy instead of >= 1 bit a symbol


def test_oracle_v3_stream_level_edge_sizes():
    rng = np.random.default_rng(3)
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 4095, 16383, 16384, 16385, 32768 + 100, 65535, 65536, 65537, 150000):
        for kind in range(5):
            if kind == 0:
                x = rng.choice(np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 12, 244, 14], dtype=np.uint8), n)
            elif kind == 1:
                x = rng.integers(0, 256, n, dtype=np.uint8)
            elif kind == 2:
                x = np.zeros(n, np.uint8)
                x[n // 2:] = rng.integers(0, 3, n - n // 2)
            elif kind == 3:
                x = (rng.normal(0, 3, n).astype(np.int64) & 255).astype(np.uint8)
            else:
                x = np.full(n, 7, np.uint8)
                x[::max(1, n // 3)] = 9  # almost one symbol: a frequency of 4095
            src = x.tobytes()
            z = O.entropy_encode(src, 1, 3)
            assert O.entropy_decode(z, n) == src, (n, kind)
            assert len(z) <= len(O.entropy_encode(src, 1, 2)) + 80 * ((n + 16383) // 16384), (n, kind)
            # other streams are not touched by the version
            assert O.entropy_encode(src, 5, 3) == O.entropy_encode(src, 5, 2)


def test_oracle_v3_rejects_damaged_blocks():
    text = make_fastq(3000, seed=45)
    z = bytearray(O.compress(text, entropy=2))
    off, size = _qual_payload(z)
    rng = np.random.default_rng(9)
    errors = 0
    for _ in range(150):
        y = bytearray(z)
        p = off + int(rng.integers(0, size))
        y[p] ^= 1 << int(rng.integers(0, 8))
        try:
            back = O.decompress(bytes(y))
        except O.OracleError:
            errors += 1
            continue
        assert back == text  # (a flip inside the index frame's hints may be harmless; wrong text never is)
    assert errors > 120


@pytest.mark.gpu
def test_gpu_v3_matches_oracle_and_round_trips(fq):
    for name, text in _cases():
        want = O.compress(text, entropy=2)
        got = fq.compress.Compress(text, _v3(fq))
        assert got == want, name
        assert fq.compress.Decompress(got) == text, name
        assert O.decompress(got) == text, name
        # the general path (what a payload falls back to when its index does not add up): the frame walk finds the
        # rANS groups itself
        os.environ["FQZ_DEC_GENERAL"] = "1"
        try:
            assert fq.compress.Decompress(got) == text, name
        finally:
            del os.environ["FQZ_DEC_GENERAL"]
        # and the version-2 path is what it was
        assert fq.compress.Compress(text) == O.compress(text), name


@pytest.mark.gpu
def test_gpu_v3_many_blocks_and_small_blocks(fq):
    import torch
    text = make_fastq(12000, seed=46, min_len=80, max_len=160)
    t = np.frombuffer(text, dtype=np.uint8)
    d_text = torch.from_numpy(t.copy()).cuda()
    cap = len(text) * 2 + (1 << 20)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for rpb in (1000, 64, 7):
        res = fq.compress.encode_batch_dev(d_text.data_ptr(), t.size, d_out.data_ptr(), cap, records_per_block=rpb, container_version=3)
        blocks = d_out[: res.out_len].cpu().numpy().tobytes()
        want = O.compress(text, batch_records=rpb, entropy=2)
        assert blocks == want[10:], rpb
        d_back = torch.empty(len(text) + 64, dtype=torch.uint8, device="cuda")
        r2 = fq.compress.decode_batch_dev(d_out.data_ptr(), int(res.out_len), d_back.data_ptr(), d_back.numel(), version=3, qual_encoding=res.qual_encoding)
        assert d_back[: r2.out_len].cpu().numpy().tobytes() == text, rpb


@pytest.mark.gpu
def test_gpu_v3_damage_is_an_error_never_wrong_text(fq):
    text = make_fastq(3000, seed=47)
    z = fq.compress.Compress(text, _v3(fq))
    off, size = _qual_payload(z)
    rng = np.random.default_rng(10)
    errors = 0
    for _ in range(120):
        y = bytearray(z)
        p = off + int(rng.integers(0, size))
        y[p] ^= 1 << int(rng.integers(0, 8))
        try:
            back = fq.compress.Decompress(bytes(y))
        except fq.FqzError:
            errors += 1
            continue
        assert back == text
    assert errors > 100
    # truncated: the states of the last block are gone
    with pytest.raises(fq.FqzError):
        fq.compress.Decompress(z[:-70])
    # rANS blocks in a file that claims version 2 are zstd's reserved block type: refused
    y = bytearray(z)
    y[4] = 2
    with pytest.raises(fq.FqzError):
        fq.compress.Decompress(bytes(y))
    # an unknown version stays an error, on both sides
    with pytest.raises(fq.FqzError, match="unsupported file version"):
        fq.compress.Compress(text, fq.Options(0, 0, 4))


def _table(z):
    """the block table parsed by hand: [(offset, records)], offset of the table"""
    assert z[-4:] == b"FQZX"
    at = int.from_bytes(z[-12:-4], "little")
    assert z[at:at + 8] == b"\xff\xff\xff\xffFQZX"
    nb = int.from_bytes(z[at + 8:at + 12], "little")
    assert at + 12 + 12 * nb + 12 == len(z)
    return [(int.from_bytes(z[at + 12 + 12 * b:at + 20 + 12 * b], "little"), int.from_bytes(z[at + 20 + 12 * b:at + 24 + 12 * b], "little")) for b in range(nb)], at


def test_oracle_v3_block_table():
    """FQZ-R1's optional on-disk block index: behind the last block, found from the end of the file, and the
    readers of the block chain stop at it"""
    text = make_fastq(2500, seed=50, min_len=40, max_len=120)
    plain = O.compress(text, batch_records=700, entropy=2)
    z = O.compress(text, batch_records=700, entropy=2, block_index=1)
    table, at = _table(z)
    assert z[:at] == plain and [r for _, r in table] == [700, 700, 700, 400]
    for off, rec in table:
        assert int.from_bytes(z[off:off + 4], "little") == rec  # a block header stands there
    assert O.decompress(z) == text
    assert _table(O.compress(b"", entropy=2, block_index=1))[0] == []
    with pytest.raises(O.OracleError):
        O.compress(text, block_index=1)  # version 2 is a plain chain of blocks: the stock reader would trip over a table


@pytest.mark.gpu
def test_gpu_v3_block_table(fq):
    text = make_fastq(230000, seed=51, min_len=30, max_len=50)  # three blocks of 100 000 records
    opts = fq.Options(0, 0, 3, 1)
    z = fq.compress.Compress(text, opts)
    assert z == O.compress(text, entropy=2, block_index=1)
    table = fq.compress.read_block_table(z)
    assert table == _table(z)[0] and [r for _, r in table] == [100000, 100000, 30000]
    assert fq.compress.Decompress(z) == text                          # the memory reader stops at the table
    assert fq.compress.DecompressMulti(z, [0, 0]) == text             # so does the walk that shares the blocks out
    assert fq.compress.CompressMulti(text, [0, 0, 0], opts) == z      # several devices: one table for the file
    # random access: a block decoded on its own from its offset
    ends = [o for o, _ in table[1:]] + [_table(z)[1]]
    parts = [fq.compress.decode_block(z[o:e], version=3) for (o, _), e in zip(table, ends)]
    assert b"".join(parts) == text
    # the streaming reader (callbacks) stops at the table as well
    import io
    out = io.BytesIO()
    fq.compress.DecompressStream(io.BytesIO(z), out)
    assert out.getvalue() == text
    # no table: asking for one is an error, and so is a table in a version-2 file
    with pytest.raises(fq.FqzError):
        fq.compress.read_block_table(fq.compress.Compress(text, _v3(fq)))
    with pytest.raises(fq.FqzError):
        fq.compress.Compress(text, fq.Options(0, 0, 2, 1))
    # a damaged table is refused by the lookup, the blocks are still read
    bad = bytearray(z)
    bad[-8] ^= 0x40
    with pytest.raises(fq.FqzError):
        fq.compress.read_block_table(bytes(bad))


@pytest.mark.gpu
def test_gpu_v3_block_level_entry_points(fq):
    """fqz_decode_block / fqz_decode_block_size (decompressJobToPooledBuffer's replacement) take version 3 blocks as well"""
    text = make_fastq(1500, seed=49, min_len=50, max_len=250)
    z = O.compress(text, entropy=2)
    assert fq.compress.decode_block(z[10:], version=3) == text
    with pytest.raises(fq.FqzError):
        fq.compress.decode_block(z[10:], version=2)  # reserved block type in a version-2 block


@pytest.mark.gpu
def test_gpu_v3_streaming_and_multi_device_paths(fq):
    text = make_fastq(30000, seed=48)
    want = O.compress(text, entropy=2)
    assert fq.compress.CompressMulti(text, [0, 0, 0], _v3(fq)) == want
    assert fq.compress.DecompressMulti(want, [0, 0]) == text
