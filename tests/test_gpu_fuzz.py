"""Randomised GPU-vs-oracle parity: many small inputs with odd shapes through the whole C ABI pipeline."""
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


def _cases():
    # FQZ_FUZZ_N / FQZ_FUZZ_SEED widen the sweep for a one-off soak (e.g. 400 cases with another seed after a kernel change)
    import os
    n_cases, seed = int(os.environ.get("FQZ_FUZZ_N", "60")), int(os.environ.get("FQZ_FUZZ_SEED", "20240607"))
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_cases):
        n = int(rng.choice([1, 2, 3, 7, 63, 64, 65, 127, 300, 1000, 2500]))
        lo = int(rng.choice([0, 1, 4, 15, 16, 17, 35, 63, 64, 100, 151]))
        hi = lo + int(rng.choice([0, 1, 3, 16, 50, 300, 1100]))
        out.append(dict(n_records=n, seed=1000 + i + (seed - 20240607) % 1000003, min_len=lo, max_len=hi, n_frac=float(rng.choice([0, 0, 0.01, 0.2, 1.0])),
                        phred=int(rng.choice([33, 33, 64])), plus_payload=bool(rng.integers(0, 2)), crlf=bool(rng.random() < 0.15),
                        block=int(rng.choice([1, 3, 64, 100, 1000, 100000]))))
    return out


def _gpu_encode_blocks(fq, text, block, enc, version=2):
    """device-resident batch encode with `block` records per block (the pipeline itself always batches 100 000)"""
    import torch
    dev = torch.device("cuda:0")
    t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev) if len(text) else torch.empty(0, dtype=torch.uint8, device=dev)
    out = torch.empty(fq.lib().fqz_encode_bound_blocks(len(text), block), dtype=torch.uint8, device=dev)
    res = fq.compress.encode_batch_dev(t.data_ptr() if len(text) else 0, len(text), out.data_ptr(), out.numel(), records_per_block=block,
                                       qual_encoding=enc, final=True, container_version=version)
    return out[: res.out_len].cpu().numpy().tobytes(), res


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "n%d_L%d-%d_b%d" % (c["n_records"], c["min_len"], c["max_len"], c["block"]))
def test_random_shapes_match_oracle(fq, case):
    case = dict(case)
    block = case.pop("block")
    text = make_fastq(**case)
    want = O.compress(text, batch_records=block)            # 10-byte file header + blocks of `block` records
    enc = 1 if want[9] & 0x02 else 0                        # what DetectEncoding decided on block 0
    body, res = _gpu_encode_blocks(fq, text, block, enc)
    assert res.n_records == case["n_records"]
    assert body == want[10:]
    back = fq.compress.Decompress(want[:10] + body)
    assert back == O.decompress(want)
    if not case["crlf"]:
        assert back == text  # {A,C,G,T,N} data round-trips exactly
    # the same shape as a version-3 container (FQZ-R1: the qualities in rANS blocks)
    want3 = O.compress(text, batch_records=block, entropy=2)
    body3, res3 = _gpu_encode_blocks(fq, text, block, enc, version=3)
    assert body3 == want3[10:]
    assert fq.compress.Decompress(want3[:10] + body3) == back


def test_long_reads_and_long_headers(fq):
    """reads far longer than a wave's piece round (64 x 16 bytes) and headers near the u16 limit"""
    rng = np.random.default_rng(5)
    recs = []
    for i, L in enumerate([70000, 1, 0, 4097, 65535, 20000]):
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L))
        qual = bytes(rng.integers(35, 74, L, dtype=np.uint8))
        hdr = b"@r%d " % i + b"x" * int(rng.choice([0, 10, 300, 5000, 60000]))
        recs.append(hdr + b"\n" + seq + b"\n+" + (b"p" * (i * 37)) + b"\n" + qual + b"\n")
    text = b"".join(recs)
    want = O.compress(text, batch_records=4)
    body, res = _gpu_encode_blocks(fq, text, 4, 0)
    assert body == want[10:]
    assert fq.compress.Decompress(want) == text


def test_corrupted_containers_are_refused_or_decoded_never_crash(fq):
    """bit flips and truncations anywhere in a container (block header, frame headers, tree descriptions, bit streams,
    foreign LZ payloads): the decoder must answer with an error or with some text — never hang or fault"""
    rng = np.random.default_rng(99)
    text = make_fastq(n_records=400, seed=5, min_len=60, max_len=140, n_frac=0.05, plus_payload=True)
    ours = O.compress(text)
    import struct
    body = ours[10:]
    hdr = list(struct.unpack("<9I", body[:36]))
    recs, nrec = O.parse_all(text)
    streams, _ = O.split_block(text, recs, nrec, 0)
    pays = [O.zstd_compress(streams[k], 3) if len(streams[k]) else b"" for k in range(6)]
    hdr[1:7] = [len(p) for p in pays]
    foreign = ours[:10] + struct.pack("<9I", *hdr) + b"".join(pays)
    assert fq.compress.Decompress(foreign) == text
    n_err = n_ok = 0
    for base in (ours, foreign):
        for trial in range(40):
            z = bytearray(base)
            kind = trial % 4
            if kind == 0:
                z[int(rng.integers(10, len(z)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                for _ in range(8):
                    z[int(rng.integers(10, len(z)))] = int(rng.integers(0, 256))
            elif kind == 2:
                z = z[: int(rng.integers(11, len(z)))]
            else:
                p = int(rng.integers(10, len(z) - 64))
                z[p:p + 32] = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
            try:
                out = fq.compress.Decompress(bytes(z))
                n_ok += 1
                assert isinstance(out, bytes)
                # our own frames carry content checksums (as the reference's do): damage is either refused or harmless
                # (a field no decoder reads, the index, which is only a hint) - never silently wrong text
                if base is ours and kind != 2:
                    assert out == text, "trial %d: corrupted container decoded to different text without an error" % trial
            except fq.FqzError:
                n_err += 1
    assert n_err > 40 and n_err + n_ok == 80
    # the context still works afterwards
    assert fq.compress.Decompress(ours) == text
