"""FQZ-S1, the experimental segment framing (FQZ_BATCH_SEG): the HIP segment path writes the bytes of the
oracle's specification (oracle framing=1), block by block the group framing where a block does not qualify, and every decoder
reads the result."""
import numpy as np
import pytest

import oracle_lib as O
from fastq_gen import make_fastq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fq():
    import fastqpacker_amd as fq
    return fq


def _encode_seg(fq, text, rpb=100000):
    import torch
    dev = torch.device("cuda:0")
    a = np.frombuffer(text, dtype=np.uint8)
    d_in = torch.from_numpy(a.copy()).to(dev) if a.size else torch.empty(16, dtype=torch.uint8, device=dev)
    d_out = torch.empty(int(fq.lib().fqz_encode_bound_blocks(a.size, rpb)) + 64, dtype=torch.uint8, device=dev)
    res = fq.compress.encode_batch_dev(d_in.data_ptr(), a.size, d_out.data_ptr(), d_out.numel(), records_per_block=rpb, segments=True)
    return bytes(d_out[: int(res.out_len)].cpu().numpy()), res


CASES = [
    ("tiny", dict(n=3, seed=1), 100000),
    ("fixed 150", dict(n=3000, seed=1), 100000),
    ("ragged + N, Phred+64", dict(n=2000, seed=2, min_len=35, max_len=301, n_frac=0.05, phred=64), 100000),
    ("long reads", dict(n=50, seed=3, min_len=5000, max_len=30000), 100000),
    ("reads too long for a segment: group framing", dict(n=8, seed=3, min_len=60000, max_len=70000), 100000),
    ("short reads: too many records a segment", dict(n=5000, seed=5, min_len=1, max_len=40), 100000),
    ("several blocks", dict(n=2500, seed=7, min_len=100, max_len=151), 700),
]


@pytest.mark.parametrize("name,kw,rpb", CASES, ids=[c[0] for c in CASES])
def test_segment_path_equals_oracle(fq, name, kw, rpb):
    kw = dict(kw)
    text = make_fastq(kw.pop("n"), **kw)
    got, res = _encode_seg(fq, text, rpb)
    want = O.compress(text, batch_records=rpb, framing=1)
    enc = want[9] >> 1 & 1
    assert got == want[10:], name                       # (the batch API returns the blocks; the file header is the caller's)
    z = want[:10] + got
    assert O.decompress(z) == text
    assert fq.compress.Decompress(z) == text
    assert res.qual_encoding == enc


def test_mixed_blocks_choose_their_framing_one_by_one(fq):
    """Blocks of 300 records; the middle of the file holds reads no segment workgroup can take: those blocks - and only those - are
    written with the group framing, the file is the oracle's byte for byte."""
    a = make_fastq(700, seed=11, min_len=100, max_len=151)
    b = make_fastq(6, seed=12, min_len=60000, max_len=70000)
    c = make_fastq(650, seed=13, min_len=100, max_len=151)
    text = a + b + c
    got, res = _encode_seg(fq, text, 300)
    want = O.compress(text, batch_records=300, framing=1)
    assert got == want[10:]
    assert want != O.compress(text, batch_records=300) and fq.compress.Decompress(want) == text
