"""GPU parity, decode side and whole-file round trips through the C ABI."""
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


CASES = {
    "sample": None,
    "fixed150": dict(n_records=2000, seed=1),
    "ragged_N_phred64": dict(n_records=1500, seed=2, min_len=35, max_len=301, n_frac=0.05, phred=64, plus_payload=True),
    "short": dict(n_records=300, seed=3, min_len=0, max_len=9, n_frac=0.2),
    "one": dict(n_records=1, seed=5),
}


@pytest.mark.parametrize("name", list(CASES))
def test_decode_block_matches_oracle(fq, sample_fq, name):
    text = sample_fq if CASES[name] is None else make_fastq(**CASES[name])
    enc = 1 if name == "ragged_N_phred64" else 0
    fqz = O.compress(text, batch_records=10 ** 9)            # oracle-encoded block
    assert fq.compress.decode_block(fqz[10:], 2, enc) == text
    block, _ = fq.compress.encode_block(text, enc)            # GPU-encoded block
    assert fq.compress.decode_block(block, 2, enc) == text


def test_entropy_decode_matches_oracle(fq):
    rng = np.random.default_rng(11)
    p = np.array([0.7] + [0.3 / 255] * 255)
    cases = [b"\x00" * 50000, bytes(rng.integers(0, 256, 40000, dtype=np.uint8)),
             bytes(rng.choice(256, 100000, p=p).astype(np.uint8)),
             bytes(np.minimum(rng.geometric(0.5, 120000) - 1, 60).astype(np.uint8))]
    for n in [1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 16383, 16384, 16385]:
        cases.append(bytes(rng.choice([0, 1, 2, 255, 254, 7], n, p=[.6, .15, .1, .1, .03, .02]).astype(np.uint8)))
    for i, data in enumerate(cases):
        frame = O.entropy_encode(data)
        assert fq.compress.entropy_decode(frame, len(data)) == data, "case %d" % i
    # garbage is refused, not decoded
    frame = bytearray(O.entropy_encode(cases[2]))
    with pytest.raises(fq.FqzError):
        fq.compress.entropy_decode(bytes(frame[:-3]), len(cases[2]))


def test_roundtrip_texts_from_reference_tests(fq):
    import json, os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
    for text in kat["roundtrip_texts"]["cases"]:
        t = text.encode("latin-1")
        z = fq.compress.Compress(t)
        assert z == O.compress(t)                              # same bytes as the oracle pipeline
        assert fq.compress.Decompress(z) == t
        assert O.decompress(z) == t
    assert fq.compress.Compress(b"").hex() == "46515a0002a086010000"   # App. B-7
    assert fq.compress.Decompress(fq.compress.Compress(b"")) == b""


def test_multi_block_files_and_options(fq):
    seq, qual = b"ACGT" * 38, b"efgh" * 38
    t = b"".join(b"@SEQ_%d\n%s\n+\n%s\n" % (i, seq, qual) for i in range(500))
    z = fq.compress.Compress(t, fq.Options(100, 4))
    assert z[:10].hex() == "46515a000264000000" + "02"          # BlockSize lands in the header, Phred64 flag set
    assert fq.compress.Decompress(z, fq.DecompressOptions(4)) == t
    # a file of many blocks written by the oracle (multi-block path the reference tests never reach)
    t2 = make_fastq(900, seed=8, min_len=50, max_len=120, n_frac=0.02, plus_payload=True)
    z2 = O.compress(t2, batch_records=64, workers=3)
    assert fq.compress.Decompress(z2) == t2


def test_v1_container_and_errors(fq):
    text = b"@SEQ_1\nACGTACGT\n+\nIIIIIIII\n"
    recs, n = O.parse_all(text)
    streams, _ = O.split_block(text, recs, 1, 0)
    comp = [O.entropy_encode(streams[k]) for k in (0, 1, 2, 4, 5)]
    fh = bytes.fromhex("46515a00") + bytes([1]) + (1).to_bytes(4, "little") + b"\x00"
    bh = b"".join(x.to_bytes(4, "little") for x in [1] + [len(c) for c in comp] + [8, 8])
    assert fq.compress.Decompress(fh + bh + b"".join(comp)) == text
    with pytest.raises(fq.FqzError, match="unsupported file version"):
        fq.compress.Decompress(bytes.fromhex("46515a00") + bytes([4]) + bytes(5))
    with pytest.raises(fq.FqzError, match="invalid magic"):
        fq.compress.Decompress(b"XYZ\x00" + bytes(6))
    z = fq.compress.Compress(text)
    with pytest.raises(fq.FqzError):
        fq.compress.Decompress(z[:-1])
    with pytest.raises(fq.FqzError):
        fq.compress.Decompress(z[:20])


def test_long_read_and_lossy_bases(fq):
    seq = bytearray(b"ACGT" * 17500)
    seq[100] = ord("N")
    t = b"@SEQ_LONG\n" + bytes(seq) + b"\n+\n" + b"I" * 70000 + b"\n"
    assert fq.compress.Decompress(fq.compress.Compress(t)) == t
    t = b"@x\nacgtnRY.\n+\nIIIIIIII\n@e\n\n+\n\n"
    assert fq.compress.Decompress(fq.compress.Compress(t)) == b"@x\nACGTNNNN\n+\nIIIIIIII\n@e\n\n+\n\n"


def test_encoder_primitives_match_reference_tables(fq):
    import json, os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
    E = fq.encoder
    for c in kat["delta_encode"]["cases"]:
        assert list(E.DeltaEncode(bytearray(c["in"]))) == c["out"]
    for c in kat["delta_decode"]["cases"]:
        assert list(E.DeltaDecode(bytearray(c["in"]))) == c["out"]
    for c in kat["normalize"]["cases"]:
        assert list(E.NormalizeQuality(bytearray(c["in"].encode("latin-1")), c["enc"])) == c["out"]
        assert bytes(E.DenormalizeQuality(bytearray(c["out"]), c["enc"])) == c["in"].encode("latin-1")
    for c in kat["detect_encoding"]["cases"]:
        assert E.DetectEncoding([x.encode("latin-1") for x in c["in"]]) == c["enc"], c
    for c in kat["n_positions"]["cases"]:
        assert E.PackBases(c["seq"].encode())[1] == c["npos"]
    for c in kat["case_folding"]["cases"]:
        p, n = E.PackBases(c["seq"].encode())
        assert E.UnpackBases(p, n, len(c["seq"])) == c["decoded"].encode()
    for c in kat["packed_bytes"]["cases"]:
        assert E.PackBases(c["seq"].encode())[0].hex() == c["hex"]
    a = kat["append_semantics"]
    dst, npos = bytearray(), []
    E.AppendPackedBases(dst, a["first"].encode(), npos)
    assert npos == []
    npos2 = []
    E.AppendPackedBases(dst, a["second"].encode(), npos2)
    assert len(dst) == a["total_packed"] and npos2 == a["second_npos"]
    assert E.PackBases(b"") == (None, None)
    rng = np.random.default_rng(5)
    s = bytes(rng.choice(np.frombuffer(b"ACGTNacgtn", dtype=np.uint8), 70001))
    got = E.PackBases(s)
    want = O.pack_bases(s)
    assert got[0] == want[0] and got[1] == want[1]
    q = bytes(rng.integers(33, 74, 5000, dtype=np.uint8))
    assert bytes(E.DeltaDecode(E.DeltaEncode(bytearray(q)))) == q
    assert bytes(E.DeltaEncode(bytearray(q))) == O.delta_encode(q)


def test_walk_edge_cases_long_headers_and_records_spanning_tiles(fq):
    """The decoder cuts the length-prefixed streams into 16 KiB tiles: cover entries beyond the precomputed
    range (headers > 254 bytes), records longer than a tile (40 kbp of N -> an 80 KB nPos record) and many
    tiny records."""
    rng = np.random.default_rng(77)
    recs = []
    for i in range(400):
        hlen = int(rng.choice([5, 40, 250, 255, 256, 300, 1000, 5000]))
        hdr = bytes(rng.integers(97, 123, hlen, dtype=np.uint8))
        L = int(rng.choice([0, 1, 7, 150, 400]))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L))
        plus = hdr[: int(rng.integers(0, 3)) * 100]
        q = bytes(rng.integers(33, 74, L, dtype=np.uint8))
        recs.append(b"@" + hdr + b"\n" + seq + b"\n+" + plus + b"\n" + q + b"\n")
    big = b"N" * 40000 + b"ACGT" * 100
    recs.insert(200, b"@big\n" + big + b"\n+\n" + b"I" * len(big) + b"\n")
    recs.insert(201, b"@big2\n" + big[::-1] + b"\n+\n" + b"#" * len(big) + b"\n")
    text = b"".join(recs)
    z = O.compress(text, batch_records=150)
    assert fq.compress.Decompress(z) == text
    z2 = fq.compress.Compress(text)
    assert z2 == O.compress(text)
    assert fq.compress.Decompress(z2) == text
    # truncated streams are refused with the reference's messages
    blk, n = fq.compress.encode_block(text, 0)
    hdr = [int.from_bytes(blk[4 * i: 4 * i + 4], "little") for i in range(9)]
    bad = bytearray(blk)
    bad[0:4] = (hdr[0] + 5).to_bytes(4, "little")    # claims more records than the streams hold
    with pytest.raises(fq.FqzError, match="truncated"):
        fq.compress.decode_block(bytes(bad), 2, 0)


def test_foreign_frame_layout_takes_the_general_path(fq):
    """The decoder sizes its tables from the frame headers (one frame, 16 KiB blocks: what our encoder writes) and
    verifies that guess while it walks the blocks.  A payload cut differently — here the lengths stream as ONE raw
    block, and the quality stream as two frames — must still decode (the decoder falls back to the two-walk path)."""
    import struct
    text = make_fastq(n_records=10000, seed=31, min_len=100, max_len=100)
    fqz = O.compress(text, batch_records=10 ** 9)
    body = fqz[10:]
    hdr = list(struct.unpack("<9I", body[:36]))   # records, 6 sizes, orig size, reserved (container.go:97-109)
    sizes = hdr[1:7]
    pays, pos = [], 36
    for n in sizes:
        pays.append(body[pos:pos + n]); pos += n
    assert pos == len(body)

    def raw_frame(data):
        assert len(data) < (1 << 17)
        return b"\x28\xb5\x2f\xfd" + bytes([0x80, 0x38]) + struct.pack("<I", len(data)) + \
            struct.pack("<I", 1 | (0 << 1) | (len(data) << 3))[:3] + data

    lengths = struct.pack("<I", 100) * 10000                 # 40 000 bytes: 3 chunks expected, 1 block found
    pays[5] = raw_frame(lengths)
    qual = O.zstd_decompress(pays[1], 10000 * 100)
    pays[1] = O.entropy_encode(qual[:300000]) + O.entropy_encode(qual[300000:])   # two concatenated frames
    hdr[1:7] = [len(p) for p in pays]
    block = struct.pack("<9I", *hdr) + b"".join(pays)
    assert fq.compress.decode_block(block, 2, 0) == text


def _lz_cases():
    rng = np.random.default_rng(77)
    text = make_fastq(n_records=3000, seed=41, min_len=100, max_len=151, n_frac=0.01)
    words = [b"ACGT", b"GATTACA", b"TTTT", b"@SEQ_ID_", b" length=", b"\n+\n", b"IIIIIIII", b"1:N:0:ATCACG"]
    cases = {
        "fastq_text": text,                                                   # literals + matches, several blocks
        "repeats": b"".join(words[i] for i in rng.integers(0, len(words), 60000)),
        "zeros": b"\x00" * 300000,                                           # RLE-like: one long overlapping match
        "short_period": (b"ab" * 70000) + b"c" + (b"xyz" * 50000),         # offset < match length
        "random": bytes(rng.integers(0, 256, 200000, dtype=np.uint8)),       # raw blocks
        "skewed": bytes(rng.choice(256, 150000, p=np.array([0.6] + [0.4 / 255] * 255)).astype(np.uint8)),
        "tiny": b"hello hello hello hello",
        "one": b"x",
    }
    return cases


@pytest.mark.parametrize("level", [1, 3, 19])
def test_foreign_zstd_frames_with_lz_sequences(fq, level):
    """frames written by another zstd encoder (here the system libzstd, which plays the role of the
    klauspost encoder of the stock fqpack) carry LZ sequences; the GPU decodes them block by block."""
    for name, data in _lz_cases().items():
        frame = O.zstd_compress(data, level)
        assert fq.compress.entropy_decode(frame, len(data)) == data, (name, level)
    # a corrupted frame is refused
    frame = bytearray(O.zstd_compress(_lz_cases()["fastq_text"], level))
    frame[len(frame) // 2] ^= 0x55
    try:
        out = fq.compress.entropy_decode(bytes(frame), len(_lz_cases()["fastq_text"]))
        assert out != _lz_cases()["fastq_text"]      # (a flipped literal byte can still be a valid stream)
    except fq.FqzError:
        pass


def test_stock_style_container_decodes(fq):
    """A .fqz block whose six payloads were produced by a general zstd encoder (what the stock fqpack writes) decodes to
    the same text as the oracle pipeline."""
    import struct
    text = make_fastq(n_records=20000, seed=43, min_len=80, max_len=151, n_frac=0.02)
    fqz = O.compress(text, batch_records=10 ** 9)
    body = fqz[10:]
    hdr = list(struct.unpack("<9I", body[:36]))
    pays, pos = [], 36
    for n in hdr[1:7]:
        pays.append(body[pos:pos + n]); pos += n
    recs, nrec = O.parse_all(text)
    streams, _ = O.split_block(text, recs, nrec, 0)
    order = [0, 1, 2, 3, 4, 5]  # wire order = seq, qual, headers, plus, nPos, lengths = stream order of the oracle
    new = [O.zstd_compress(streams[k], 1) if len(streams[k]) else b"" for k in order]
    hdr[1:7] = [len(p) for p in new]
    block = struct.pack("<9I", *hdr) + b"".join(new)
    assert fq.compress.decode_block(block, 2, 0) == text
    # and whole-file: header + block through Decompress
    assert fq.compress.Decompress(fqz[:10] + block) == text


def _v2_block(nrec, streams, orig=0):
    """A v2 block whose six payloads (stream order seq, qual, headers, plus, nPos, lengths) are oracle frames."""
    import struct
    pays = [O.entropy_encode(s) if len(s) else b"" for s in streams]
    return struct.pack("<9I", nrec, *[len(p) for p in pays], orig, orig) + b"".join(pays)


def test_crafted_blocks_whose_sizes_wrap_32_bits_are_refused(fq):
    """per-record sizes and their sums must not wrap: 16 records of L = 2^30 make every 32-bit total 0."""
    import struct
    nrec = 16
    lengths = struct.pack("<I", 0x40000000) * nrec
    block = _v2_block(nrec, [b"", b"", b"\x00\x00" * nrec, b"", b"\x00\x00" * nrec, lengths])
    with pytest.raises(fq.FqzError, match="truncated"):
        fq.compress.decode_block(block, 2, 0)
    # a single record with the largest length the old bound let through
    block = _v2_block(1, [b"", b"", b"\x00\x00", b"", b"\x00\x00", struct.pack("<I", 0x7FFFFFFF)])
    with pytest.raises(fq.FqzError, match="truncated"):
        fq.compress.decode_block(block, 2, 0)
    # lengths that fit the streams individually but not in sum (64-bit block totals)
    seq, qual = b"\x00" * 3, b"\x05" * 10
    block = _v2_block(3, [seq, qual, b"\x00\x00" * 3, b"", b"\x00\x00" * 3, struct.pack("<3I", 10, 10, 10)])
    with pytest.raises(fq.FqzError, match="truncated (sequence|quality) data"):
        fq.compress.decode_block(block, 2, 0)


def test_streams_too_short_for_their_records_are_refused(fq):
    """ADVICE r1 (medium / low): NumRecords is tied to the stream sizes before anything is sized or walked from it."""
    import struct
    one = struct.pack("<I", 4)
    # records but an empty header stream (compress.go:977-980)
    with pytest.raises(fq.FqzError, match="truncated header data"):
        fq.compress.decode_block(_v2_block(1, [b"\x00", b"IIII", b"", b"", b"\x00\x00", one]), 2, 0)
    # ... an empty N-position stream (compress.go:1055-1060)
    with pytest.raises(fq.FqzError, match="truncated N position data"):
        fq.compress.decode_block(_v2_block(1, [b"\x00", b"IIII", b"\x01\x00A", b"", b"", one]), 2, 0)
    # a block header that claims half a billion records over a 4-byte lengths stream: refused before any large allocation
    with pytest.raises(fq.FqzError, match="truncated length data"):
        fq.compress.decode_block(_v2_block(500_000_000, [b"\x00", b"IIII", b"\x01\x00A", b"", b"\x00\x00", one]), 2, 0)
    # a plus stream that is present but shorter than its prefixes
    with pytest.raises(fq.FqzError, match="truncated plus-line payload data"):
        fq.compress.decode_block(_v2_block(2, [b"\x00\x00", b"IIIIIIII", b"\x01\x00A\x01\x00B", b"\x00\x00", b"\x00\x00" * 2, one * 2]), 2, 0)
    # the well-formed twin decodes
    ok = _v2_block(1, [b"\x00", b"\x28\x00\x00\x00", b"\x01\x00A", b"", b"\x00\x00", one])
    assert fq.compress.decode_block(ok, 2, 0) == b"@A\nAAAA\n+\nIIII\n"


@pytest.mark.gpu
def test_block_offset_hint_is_a_hint(fq):
    """fqz_decode_batch_dev_hint: with the offsets of the block headers the device does not walk their chain; the offsets are
    checked against the headers (every block must end where the next one starts) and anything that does not add up - a shifted
    offset, a missing block, one too many - falls back to the walk: the text is the same in every case.  Both container versions
    (a version-3 batch may end with a block table), and a stock-style file written by the oracle's reference pipeline."""
    import torch
    from fastq_gen import make_fastq
    text = make_fastq(5000, seed=91, min_len=60, max_len=120)
    t = np.frombuffer(text, dtype=np.uint8)
    d_text = torch.from_numpy(t.copy()).cuda()
    cap = len(text) * 2 + (1 << 20)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(len(text) + 64, dtype=torch.uint8, device="cuda")
    for version in (2, 3):
        res, offs, lens = fq.compress.encode_batch_dev(d_text.data_ptr(), t.size, d_out.data_ptr(), cap, records_per_block=700, max_blocks=64,
                                                       container_version=version)
        assert res.n_blocks == 8 and offs[0] == 0 and all(offs[i] + lens[i] == offs[i + 1] for i in range(7))
        def dec(block_off):
            d_back.zero_()
            torch.cuda.synchronize()  # (torch's stream and the library's are not ordered)
            r = fq.compress.decode_batch_dev(d_out.data_ptr(), int(res.out_len), d_back.data_ptr(), d_back.numel(), version=version,
                                             qual_encoding=res.qual_encoding, block_off=block_off)
            return d_back[: r.out_len].cpu().numpy().tobytes()
        assert dec(None) == text
        assert dec(offs) == text
        assert dec([o + (3 if i == 4 else 0) for i, o in enumerate(offs)]) == text      # a wrong offset
        assert dec(offs[:-1]) == text                                                    # a block missing at the end
        assert dec(offs + [int(res.out_len)]) == text                                    # one too many
        assert dec([5] + offs[1:]) == text                                               # the first block not at 0
    # a whole version-3 file with its block table behind the last block: the hint ends where the table begins
    fqz = O.compress(text, batch_records=700, entropy=2, block_index=1)
    body = np.frombuffer(fqz, dtype=np.uint8)[10:]
    d_body = torch.from_numpy(body.copy()).cuda()
    offs3, pos = [], 0
    for _ in range(8):
        offs3.append(pos)
        hdr = [int.from_bytes(body[pos + 4 * i: pos + 4 * i + 4].tobytes(), "little") for i in range(9)]
        pos += 36 + sum(hdr[1:7])
    d_back.zero_()
    torch.cuda.synchronize()
    r = fq.compress.decode_batch_dev(d_body.data_ptr(), d_body.numel(), d_back.data_ptr(), d_back.numel(), version=3, qual_encoding=0, block_off=offs3)
    assert d_back[: r.out_len].cpu().numpy().tobytes() == text
