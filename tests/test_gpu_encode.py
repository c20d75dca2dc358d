"""GPU parity, encode side: the HIP path (through the C ABI) against the CPU oracle on the same
inputs — pre-entropy streams byte-for-byte, compressed blocks byte-for-byte, and the payloads
decodable by the independent system libzstd."""
import os

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


def _oracle_block(text, enc=0):
    recs, n = O.parse_all(text)
    streams, orig = O.split_block(text, recs, n, enc)
    return streams, n


def _dump_diff(name, a, b):
    os.makedirs("gpurun_out", exist_ok=True)
    i = next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), min(len(a), len(b)))
    return "%s differs: len gpu=%d oracle=%d first diff at %d gpu=%s oracle=%s" % (
        name, len(a), len(b), i, a[max(0, i - 8): i + 24].hex(), b[max(0, i - 8): i + 24].hex())


CASES = {
    "sample": None,
    "fixed150": dict(n_records=2000, seed=1),
    "ragged_N_phred64": dict(n_records=1500, seed=2, min_len=35, max_len=301, n_frac=0.05, phred=64, plus_payload=True),
    "short": dict(n_records=300, seed=3, min_len=0, max_len=9, n_frac=0.2),
    "crlf": dict(n_records=100, seed=4, crlf=True),
    "one": dict(n_records=1, seed=5),
}


@pytest.mark.parametrize("name", list(CASES))
def test_streams_and_block_match_oracle(fq, sample_fq, name):
    text = sample_fq if CASES[name] is None else make_fastq(**CASES[name])
    enc = 1 if name == "ragged_N_phred64" else 0
    want_streams, n = _oracle_block(text, enc)
    # default pipeline: the six pre-entropy streams exist in HBM and can be compared one by one
    block, nrec = fq.compress.encode_block(text, enc)
    assert nrec == n
    got_streams = fq.compress.get_streams(0)
    for k, nm in enumerate(O.STREAM_NAMES):
        assert got_streams[k] == want_streams[k], _dump_diff(nm, got_streams[k], want_streams[k])
    # block = 36-byte header + payloads; compare with the oracle's compress() minus the 10-byte file header
    want = O.compress(text, batch_records=10 ** 9)[10:]
    if enc == 1:
        assert O.compress(text)[9] == 2
    assert block == want, _dump_diff("block", block, want)
    # independent conformance: every payload is a zstd frame libzstd accepts
    if O.libzstd() is not None:
        hdr = [int.from_bytes(block[4 * i: 4 * i + 4], "little") for i in range(9)]
        pos = 36
        for k, size in enumerate(hdr[1:7]):
            payload = block[pos: pos + size]
            pos += size
            assert O.zstd_decompress(payload, len(want_streams[k]) + 1) == want_streams[k], O.STREAM_NAMES[k]
        assert pos == len(block)


def test_entropy_stage_matches_oracle(fq):
    rng = np.random.default_rng(11)
    p = np.array([0.7] + [0.3 / 255] * 255)
    cases = [
        b"\x00" * 50000,
        bytes(rng.integers(0, 256, 40000, dtype=np.uint8)),
        bytes(rng.choice(256, 100000, p=p).astype(np.uint8)),
        bytes(np.minimum(rng.geometric(0.5, 120000) - 1, 60).astype(np.uint8)),
        bytes(np.minimum(rng.geometric(0.3, 16384) - 1, 200).astype(np.uint8)),
        (150).to_bytes(4, "little") * 30000,
    ]
    for n in [1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 16383, 16384, 16385, 16384 * 3 + 5]:
        cases.append(bytes(rng.choice([0, 1, 2, 255, 254, 7], n, p=[.6, .15, .1, .1, .03, .02]).astype(np.uint8)))
    cases += [d for _, d in O.group_cases()]
    for i, data in enumerate(cases):
        got = fq.compress.entropy_encode(data)
        want = O.entropy_encode(data)
        assert got == want, _dump_diff("case %d (n=%d)" % (i, len(data)), got, want)
        assert fq.compress.entropy_decode(got, len(data)) == data


def test_multi_block_batch_matches_oracle(fq):
    text = make_fastq(1000, seed=21, min_len=80, max_len=160, n_frac=0.01)
    import ctypes as C
    import torch
    dev = torch.device("cuda:0")
    t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    out = torch.empty(len(text) * 2 + 65536, dtype=torch.uint8, device=dev)
    res, offs, lens = fq.compress.encode_batch_dev(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), records_per_block=128,
                                                   final=True, max_blocks=64)
    assert res.n_records == 1000 and res.n_blocks == 8
    body = out[: res.out_len].cpu().numpy().tobytes()
    want = O.compress(text, batch_records=128)
    assert body == want[10:]
    assert offs[0] == 0 and offs[1] == lens[0]
    # not final: only whole blocks are consumed
    res2 = fq.compress.encode_batch_dev(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), records_per_block=300, final=False)
    assert res2.n_records == 900 and res2.n_blocks == 3
    recs, n = O.parse_all(text)
    assert res2.consumed == recs[900].hdr_off - 1


def test_tiny_records_fall_back_to_two_pass_index(fq):
    """Lines of 2 bytes on average: a 4 KiB tile holds ~1800 of them, more than a tile-local slot of the single-pass line
    index (512) and more lines than the optimistic line table; the encoder redoes the batch with the two-pass index and a
    larger table.  The same context must keep working for ordinary input afterwards (on the two-pass path for the next 16
    launches, then single-pass again)."""
    import torch
    dev = torch.device("cuda:0")
    ctx = fq.Ctx(0)
    tiny = b"".join(b"@%c\n%c\n+\n%c\n" % (97 + i % 26, b"ACGTN"[i % 5], 33 + i % 40) for i in range(30000))
    normal = make_fastq(3000, seed=77, min_len=100, max_len=151, n_frac=0.01)
    for text, rpb in [(tiny, 7000), (normal, 1000), (tiny, 7000)] + [(normal, 1000)] * 18 + [(tiny, 7000)]:
        t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
        out = torch.empty(len(text) * 4 + (1 << 20), dtype=torch.uint8, device=dev)
        res = fq.compress.encode_batch_dev(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), records_per_block=rpb, final=True, ctx=ctx)
        body = out[: res.out_len].cpu().numpy().tobytes()
        want = O.compress(text, batch_records=rpb)
        assert body == want[10:], _dump_diff("container", body, want[10:])
        assert O.decompress(want[:10] + body) == text


def test_parser_errors_match_reference_messages(fq):
    with pytest.raises(fq.FqzError, match="header line must start with @"):
        fq.compress.encode_block(b"r1\nACGT\n+\nIIII\n")
    with pytest.raises(fq.FqzError, match="separator line must start with \\+"):
        fq.compress.encode_block(b"@r1\nACGT\n-\nIIII\n")
    with pytest.raises(fq.FqzError, match="lengths must match"):
        fq.compress.encode_block(b"@r1\nACGT\n+\nIII\n")
    with pytest.raises(fq.FqzError, match="header line must start with @"):
        fq.compress.encode_block(b"@r1\nACGT\n+\nIIII\n\n")
    seq = bytearray(b"ACGT" * 17500)
    seq[66000] = ord("N")
    with pytest.raises(fq.FqzError, match="ambiguous bases beyond position"):
        fq.compress.encode_block(b"@SEQ_LONG\n" + bytes(seq) + b"\n+\n" + b"I" * 70000 + b"\n")
    # dropped unterminated tail (App. B-3)
    block, n = fq.compress.encode_block(b"@r1\nACGT\n+\nIIII\n@r2\nAC\n+\nII")
    assert n == 1


def test_ragged_phred64_default_blocks_match_oracle_and_round_trip(fq):
    """BASELINE config 5 shape (35-301 bp, 5 % N in runs, Phred+64, plus payloads) at the real block size:
    120 000 records = 1.2 blocks of 100 000.  Whole-file bytes == oracle pipeline, and the text round-trips
    through both decoders."""
    text = make_fastq(n_records=120000, seed=99, min_len=35, max_len=301, n_frac=0.05, phred=64, plus_payload=True)
    z = fq.compress.Compress(text)
    want = O.compress(text)
    assert len(z) == len(want) and z == want
    assert z[9] & 0x02                                # FlagPhred64 detected from block 0 (quality.go:43-45, container.go:16)
    assert fq.compress.Decompress(z) == text
    assert O.decompress(z) == text
