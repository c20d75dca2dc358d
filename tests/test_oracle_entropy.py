"""Entropy-stage oracle: every payload is a valid zstd frame (checked with the system
libzstd as an INDEPENDENT decoder) that regenerates the pre-entropy stream exactly."""
import numpy as np
import pytest

import oracle_lib as O
from fastq_gen import make_fastq

needs_zstd = pytest.mark.skipif(O.libzstd() is None, reason="system libzstd.so.1 not present")


def _cases():
    rng = np.random.default_rng(7)
    yield "empty", b""
    yield "rle", b"\x00" * 70000
    yield "two-syms", bytes(rng.integers(0, 2, 5000, dtype=np.uint8) * 255)
    yield "uniform256", bytes(rng.integers(0, 256, 40000, dtype=np.uint8))
    p = np.array([0.7] + [0.3 / 255] * 255)
    yield "skew-high-symbols(FSE weights)", bytes(rng.choice(256, 100000, p=p).astype(np.uint8))
    yield "geometric(deep tree, length limit)", bytes(np.minimum(rng.geometric(0.5, 120000) - 1, 60).astype(np.uint8))
    yield "geometric2", bytes(np.minimum(rng.geometric(0.3, 16384) - 1, 200).astype(np.uint8))
    for n in [1, 2, 3, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 16383, 16384, 16385, 16384 * 3 + 5]:
        yield "n=%d" % n, bytes(rng.choice([0, 1, 2, 255, 254, 7], n, p=[.6, .15, .1, .1, .03, .02]).astype(np.uint8))
    yield "lengths-stream", (150).to_bytes(4, "little") * 30000


@needs_zstd
def test_frames_decode_with_libzstd():
    import itertools
    for name, data in itertools.chain(_cases(), O.group_cases()):
        f = O.entropy_encode(data)
        if not data:
            assert f == b""
            continue
        assert f[:4] == bytes.fromhex("28b52ffd")
        assert O.zstd_decompress(f, len(data)) == data, name
        assert O.entropy_decode(f) == data, name


def test_own_decoder_roundtrip_and_rejects_garbage():
    import itertools
    for name, data in itertools.chain(_cases(), O.group_cases()):
        assert O.entropy_decode(O.entropy_encode(data), len(data)) == data, name
    f = bytearray(O.entropy_encode(bytes(np.random.default_rng(1).integers(0, 4, 20000, dtype=np.uint8))))
    with pytest.raises(O.OracleError):
        O.entropy_decode(bytes(f[:-3]), 20000)
    f[0] ^= 1
    with pytest.raises(O.OracleError):
        O.entropy_decode(bytes(f), 20000)


def test_code_lengths_are_complete_and_limited():
    rng = np.random.default_rng(3)
    for trial in range(200):
        k = int(rng.integers(2, 257))
        counts = np.zeros(256, dtype=np.int64)
        syms = rng.choice(256, k, replace=False)
        kind = trial % 4
        if kind == 0:
            counts[syms] = rng.integers(1, 50, k)
        elif kind == 1:
            counts[syms] = np.maximum(1, (16384 * 0.5 ** np.arange(1, k + 1)).astype(np.int64))
        elif kind == 2:
            fib = [1, 1]
            while len(fib) < k:
                fib.append(min(fib[-1] + fib[-2], 100000))
            counts[syms] = fib[:k]
        else:
            counts[syms] = 1
        mx, nb = O.huf_code_lengths([int(c) for c in counts])
        nb = np.frombuffer(nb, dtype=np.uint8)
        assert mx == nb.max() <= 11
        assert ((nb > 0) == (counts > 0)).all()
        assert sum(2.0 ** -int(x) for x in nb if x) == 1.0  # Kraft equality: complete prefix code


@needs_zstd
def test_all_six_streams_of_generated_fastq():
    for kw in (dict(), dict(min_len=35, max_len=301, n_frac=0.05, phred=64, plus_payload=True)):
        text = make_fastq(700, seed=5, **kw)
        recs, n = O.parse_all(text)
        enc = 1 if kw else 0
        streams, _ = O.split_block(text, recs, n, enc)
        for s in streams:
            f = O.entropy_encode(s)
            assert O.zstd_decompress(f, len(s)) == s
        # whole-file: the stock decoder only needs valid frames + the container; emulate it
        z = O.compress(text, workers=2, batch_records=256)
        assert O.decompress(z) == text


@needs_zstd
def test_oracle_decodes_foreign_zstd_frames_via_libzstd():
    # a file whose payloads come from a real zstd level-1 encoder (what the stock encoder emits)
    text = make_fastq(300, seed=9)
    z = O.compress(text, entropy=1)
    assert O.decompress(z) == text


def test_group_shapes_use_treeless_blocks():
    """the blocks after the first Compressed block of a group are treeless (Literals_Block_Type 3)"""
    data = dict(O.group_cases())["two-groups-and-a-tail"]
    f = O.entropy_encode(data)
    pos, types = 10, []
    while True:
        bh = int.from_bytes(f[pos:pos + 3], "little")
        last, btype, bs = bh & 1, (bh >> 1) & 3, bh >> 3
        types.append((btype, f[pos + 3] & 3 if btype == 2 else None))
        pos += 3 + (1 if btype == 1 else bs)
        if last:
            break
    assert pos == len(f)
    assert types == [(2, 2), (2, 3), (2, 3), (2, 3)] * 2 + [(2, 2), (2, 3)]
