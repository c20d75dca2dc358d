"""Headers-stream modelling of the FQZ-H2 profile (zstd sequences with the predefined tables; oracle/fqz_entropy.c
hdr_chunk_model, fastqpacker_amd/csrc/fqz_hdrlz.h): an independent decoder (libzstd) accepts it, the GPU writes the same
bytes as the oracle and reads them back on its fast path AND on the general one."""
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


def _records(headers, seed=3, L=40):
    rng = np.random.default_rng(seed)
    out = []
    for h in headers:
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L).tobytes()
        out.append(b"@" + h + b"\n" + seq + b"\n+\n" + b"I" * L + b"\n")
    return b"".join(out)


def _hdr_cases():
    rng = np.random.default_rng(5)
    yield "illumina", make_fastq(3000, seed=31, min_len=30, max_len=60)
    yield "identical", _records([b"read"] * 5000)
    yield "identical-long", _records([b"x" * 300] * 400)
    yield "sra-short (more records per chunk than the model takes)", _records([b"SRR1.%d" % i for i in range(9000)])
    yield "sra-long", _records([b"SRR1234567.%d %d length=150" % (i, i) for i in range(4000)])
    yield "random (nothing to match)", _records([bytes(rng.integers(33, 127, int(rng.integers(1, 80)), dtype=np.uint8)).replace(b"\n", b"x") for _ in range(3000)])
    yield "empty headers", _records([b""] * 3000)
    yield "one symbol", _records([b"A" * int(n) for n in rng.integers(0, 50, 3000)])
    yield "mixed lengths", _records([b"inst:%d:%s/1" % (i // 7, b"T" * int(rng.integers(0, 200))) for i in range(2500)])
    yield "long headers (few records per chunk)", _records([b"%d|" % i + b"ACME-SEQ-9000 run=77 lane=3 " * 40 for i in range(300)])
    yield "tail only", _records([bytes(rng.integers(65, 91, 12, dtype=np.uint8)) + b" common tail of some length" for _ in range(3000)])
    yield "very long headers (offsets beyond the decoder's ring)", _records([b"%07d|" % i + bytes(rng.integers(65, 91, 600, dtype=np.uint8)) * (0 if i % 5 else 1) + b"GATTACA" * 700 for i in range(60)])
    yield "long runs of one byte (overlapping matches)", _records([b"N" * int(n) for n in rng.integers(1500, 3000, 60)])
    yield "mostly headers, 3 MB (more headers chunks than the encoder's side buffers expect: it relaunches)", _records(
        [b"run7:lane%d:tile%d:" % (i % 8, i // 97) + bytes(rng.integers(97, 123, 900, dtype=np.uint8)) + b":%d" % i for i in range(3300)], L=8)
    yield "crlf", make_fastq(800, seed=32, crlf=True)


def _hdr_stream(text):
    recs, n = O.parse_all(text)
    return O.split_block(text, recs, n, 0)[0][2]


@pytest.mark.skipif(O.libzstd() is None, reason="system libzstd.so.1 not present")
def test_oracle_headers_payload_is_zstd_and_smaller():
    sizes = {}
    for name, text in _hdr_cases():
        h = _hdr_stream(text)
        f = O.entropy_encode(h, stream=2)
        assert O.zstd_decompress(f, len(h) + 1) == h, name
        assert O.entropy_decode(f, len(h)) == h, name
        plain = O.entropy_encode(h, stream=1)
        assert len(f) <= len(plain) * 1.05 + 64, name   # (no per-block choice between the two codings: a few percent can be lost on odd data)
        sizes[name] = (len(h), len(plain), len(f))
    h, plain, f = sizes["illumina"]
    assert f * 1.3 < plain                     # Illumina headers: prefix + suffix of the predecessor
    h, plain, f = sizes["identical-long"]
    assert f * 20 < h


def test_oracle_blocks_are_independent():
    """No match reaches in front of its 16 KiB block and the first sequence of a block never uses a repeat offset: any
    group (= zstd frame) of the payload decodes alone."""
    text = make_fastq(3000, seed=33, min_len=30, max_len=60)
    h = _hdr_stream(text)
    f = O.entropy_encode(h, stream=2)
    idx, frames = O.payload_frames(f)
    assert len(frames) >= 2
    pos = 0
    for fr in frames:
        part = O.entropy_decode(fr)
        assert part == h[pos:pos + len(part)]
        pos += len(part)
    assert pos == len(h)


@pytest.mark.gpu
def test_gpu_headers_match_oracle_and_round_trip(fq):
    for name, text in _hdr_cases():
        want = O.compress(text)
        got = fq.compress.Compress(text)
        assert got == want, name
        text = text.replace(b"\r\n", b"\n")     # (line ends are not kept: parser.go readLine strips '\r')
        assert fq.compress.Decompress(got) == text, name
        os.environ["FQZ_DEC_GENERAL"] = "1"   # the general walk hands such payloads to the full zstd decoder (k_dec_lz)
        try:
            assert fq.compress.Decompress(got) == text, name
        finally:
            del os.environ["FQZ_DEC_GENERAL"]


@pytest.mark.gpu
def test_gpu_rejects_damaged_sequences(fq):
    """Flipping bits inside a Sequences_Section must never pass: the content checksum of the frame catches what the
    structure checks let through."""
    text = make_fastq(2000, seed=34, min_len=30, max_len=60)
    good = bytearray(fq.compress.Compress(text))
    hdr = [int.from_bytes(good[10 + 4 * i: 14 + 4 * i], "little") for i in range(9)]
    h_off = 10 + 36 + hdr[1] + hdr[2]          # headers payload
    rng = np.random.default_rng(9)
    n_fail = 0
    for _ in range(24):
        bad = bytearray(good)
        at = h_off + int(rng.integers(40, hdr[3]))
        bad[at] ^= 1 << int(rng.integers(0, 8))
        try:
            out = fq.compress.Decompress(bytes(bad))
        except Exception:
            n_fail += 1
            continue
        assert out == text                     # (a flip in padding bits of a bitstream can be harmless)
    assert n_fail >= 20


@pytest.mark.gpu
def test_record_samples_are_a_hint(fq):
    """The index frame of the headers payload carries the stream offset of every 64th record.  A wrong sample must not
    change the result: the decoder checks every sample against the walk between its neighbours and falls back to walking
    the chain from the start."""
    text = make_fastq(1000, seed=35, min_len=30, max_len=60)
    good = fq.compress.Compress(text)
    assert good == O.compress(text)
    hdr = [int.from_bytes(good[10 + 4 * i: 14 + 4 * i], "little") for i in range(9)]
    h_off = 10 + 36 + hdr[1] + hdr[2]
    pay = good[h_off: h_off + hdr[3]]
    assert pay[8:12] == b"FQZI" and pay[14] == 3                      # flags: samples and entry points present
    nch = int.from_bytes(pay[20:24], "little")
    at = 24 + 3 * nch
    assert int.from_bytes(pay[at:at + 4], "little") == 1000             # the record count they were made for
    n_s = (1000 - 1) // 64
    samples = [int.from_bytes(pay[at + 4 + 4 * k: at + 8 + 4 * k], "little") for k in range(n_s)]
    recs, n = O.parse_all(text)
    h = O.split_block(text, recs, n, 0)[0][2]
    pos, starts = 0, []
    while pos < len(h):
        starts.append(pos)
        pos += 2 + int.from_bytes(h[pos:pos + 2], "little")
    assert samples == [starts[64 * (k + 1)] for k in range(n_s)]
    assert fq.compress.Decompress(good) == text
    for k in (0, 7, n_s - 1):
        bad = bytearray(good)
        bad[h_off + at + 4 + 4 * k] ^= 0x10
        assert fq.compress.Decompress(bytes(bad)) == text


@pytest.mark.gpu
def test_entry_points_are_a_hint(fq):
    """The index frames carry, per zstd block, three entry points into each of its four Huffman streams (FQZI flags bit 1): the
    decoder's quarter-lanes start there.  They are what the oracle says (the bits of the symbols behind the entry), every stock
    decoder ignores them, and a wrong one must not change the result: a quarter that does not end where the next one began sends
    the batch to the general path, which reads the streams from their end marks."""
    text = make_fastq(3000, seed=36, min_len=100, max_len=150)
    good = fq.compress.Compress(text)
    assert good == O.compress(text)                                   # (two implementations computed the same entry points)
    hdr = [int.from_bytes(good[10 + 4 * i: 14 + 4 * i], "little") for i in range(9)]
    q_off = 10 + 36 + hdr[1]
    pay = good[q_off: q_off + hdr[2]]
    assert pay[8:12] == b"FQZI" and pay[13] == 1 and pay[14] == 2      # the quality payload: entry points, no record samples
    nch = int.from_bytes(pay[20:24], "little")
    assert nch >= 10
    at = 24 + 3 * nch
    ent = np.frombuffer(pay[at: at + 24 * nch], dtype="<u2").reshape(nch, 4, 3).astype(np.int64)
    sizes = [int.from_bytes(pay[24 + 3 * c: 27 + 3 * c], "little") for c in range(nch)]
    assert (ent[:, :, 0] >= ent[:, :, 1]).all() and (ent[:, :, 1] >= ent[:, :, 2]).all()
    assert (ent[0] > 0).all() and (ent[0, :, 0] < 8 * sizes[0]).all()   # a full block: every quarter has bits
    # (the seq payload - Raw blocks by definition - carries none)
    assert good[10 + 36 + 8: 10 + 36 + 12] == b"FQZI" and good[10 + 36 + 14] == 0
    assert fq.compress.Decompress(good) == text
    for c, j, flip in ((0, 0, 0x01), (0, 5, 0x80), (nch - 1, 11, 0x02), (3, 7, 0xFF)):
        bad = bytearray(good)
        bad[q_off + at + 24 * c + 2 * j] ^= flip
        assert fq.compress.Decompress(bytes(bad)) == text
    bad = bytearray(good)
    bad[q_off + at: q_off + at + 24 * nch] = b"\xff" * (24 * nch)
    assert fq.compress.Decompress(bytes(bad)) == text
