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
        idx, frames = O.payload_frames(f)
        assert idx is not None and f.startswith(idx) and idx[8:12] == b"FQZI"      # skippable index frame first
        assert len(frames) == (len(data) + 65535) // 65536                            # one zstd frame per 64 KiB group
        assert all(fr[:4] == bytes.fromhex("28b52ffd") for fr in frames)
        assert O.zstd_decompress(f, len(data)) == data, name                        # libzstd skips the index, verifies every checksum
        assert O.entropy_decode(f) == data, name


def test_own_decoder_roundtrip_and_rejects_garbage():
    import itertools
    for name, data in itertools.chain(_cases(), O.group_cases()):
        assert O.entropy_decode(O.entropy_encode(data), len(data)) == data, name
    f = bytearray(O.entropy_encode(bytes(np.random.default_rng(1).integers(0, 4, 20000, dtype=np.uint8))))
    with pytest.raises(O.OracleError):
        O.entropy_decode(bytes(f[:-3]), 20000)
    idx, _ = O.payload_frames(bytes(f))
    f[len(idx)] ^= 1          # the magic of the first zstd frame (byte 0 belongs to the skippable index frame, whose 16 magics differ in that nibble)
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
    idx, frames = O.payload_frames(f)
    types, sizes = [], []
    for fr in frames:                       # one frame per group
        pos = 7 if fr[4] >> 6 else 6        # magic, FHD, FCS (2 bytes from 256 bytes of content on)
        while True:
            bh = int.from_bytes(fr[pos:pos + 3], "little")
            last, btype, bs = bh & 1, (bh >> 1) & 3, bh >> 3
            types.append((btype, fr[pos + 3] & 3 if btype == 2 else None))
            step = 3 + (1 if btype == 1 else bs)
            sizes.append(step)
            pos += step
            if last:
                break
        assert pos + 4 == len(fr)            # the content checksum ends the frame
    # the index frame lists the size of every zstd block
    n_blocks = int.from_bytes(idx[20:24], "little")
    assert int.from_bytes(idx[16:20], "little") == len(data) and n_blocks == len(sizes)
    assert [int.from_bytes(idx[24 + 3 * k:27 + 3 * k], "little") for k in range(n_blocks)] == sizes
    assert types == [(2, 2), (2, 3), (2, 3), (2, 3)] * 2 + [(2, 2), (2, 3)]


def test_xxh64_known_answers():
    """XXH64 known answers (public test vectors of the algorithm): the zstd content checksum is its low 32 bits."""
    assert O.xxh64(b"") == 0xEF46DB3751D8E999
    assert O.xxh64(b"", 1) == 0xD5AFBA1336A3BE4B
    assert O.xxh64(b"a") == 0xD24EC4F1A98C6E5B
    assert O.xxh64(b"abc") == 0x44BC2CF5AD770999
    assert O.xxh64(b"Nobody inspects the spammish repetition") == 0xFBCEA83C8A378BF1


@needs_zstd
def test_content_checksum_is_emitted_and_verified():
    """The reference keeps zstd's frame checksum on purpose (PERFORMANCE.md E033, README.md:87): every frame of a payload
    carries one, libzstd accepts it, and one flipped payload bit is an error for libzstd and for the oracle decoder."""
    rng = np.random.default_rng(5)
    data = bytes(np.minimum(rng.geometric(0.4, 200000) - 1, 40).astype(np.uint8))
    f = O.entropy_encode(data)
    idx, frames = O.payload_frames(f)
    pos = len(idx)
    for k, fr in enumerate(frames):
        assert fr[4] & 0x04                                   # Content_Checksum_flag
        want = O.xxh64(data[65536 * k:65536 * (k + 1)]) & 0xFFFFFFFF
        assert int.from_bytes(fr[-4:], "little") == want
        pos += len(fr)
    assert pos == len(f)
    # raw stream (2-bit packed bases): Raw blocks only, still checksummed
    f0 = O.entropy_encode(data, stream=0)
    assert O.zstd_decompress(f0, len(data)) == data and O.entropy_decode(f0) == data
    assert len(f0) > len(data)
    bad = 0
    for off in (len(idx) + 40, len(idx) + len(frames[0]) // 2, len(f) - 1, len(f) - 9):
        g = bytearray(f)
        g[off] ^= 0x10
        with pytest.raises(ValueError):
            O.zstd_decompress(bytes(g), len(data))
        with pytest.raises(O.OracleError):
            O.entropy_decode(bytes(g), len(data))
        bad += 1
    assert bad == 4
