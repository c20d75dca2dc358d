"""Pins the CPU oracle to the known-answer vectors the reference's own tests hold
(tests/golden/kat.json).  CPU only."""
import ctypes as C
import hashlib
import json
import os

import pytest

import oracle_lib as O

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def b(s):
    return s.encode("latin-1")


def test_delta_tables():
    for c in KAT["delta_encode"]["cases"]:
        assert list(O.delta_encode(bytes(c["in"]))) == c["out"]
    for c in KAT["delta_decode"]["cases"]:
        assert list(O.delta_decode(bytes(c["in"]))) == c["out"]


def test_normalize_tables():
    for c in KAT["normalize"]["cases"]:
        assert list(O.normalize_quality(b(c["in"]), c["enc"])) == c["out"]
        assert O.denormalize_quality(bytes(c["out"]), c["enc"]) == b(c["in"])


def test_detect_encoding_table():
    for c in KAT["detect_encoding"]["cases"]:
        assert O.detect_encoding([b(x) for x in c["in"]]) == c["enc"], c


def test_n_positions_and_case_folding():
    for c in KAT["n_positions"]["cases"]:
        packed, npos = O.pack_bases(b(c["seq"]))
        assert npos == c["npos"]
    for c in KAT["case_folding"]["cases"]:
        packed, npos = O.pack_bases(b(c["seq"]))
        assert O.unpack_bases(packed, npos, len(c["seq"])) == b(c["decoded"])


def test_packed_lengths_and_bytes():
    for n, want in KAT["packed_len"]["cases"]:
        packed, _ = O.pack_bases(b"ACGT" * (n // 4) + b"ACGT"[: n % 4])
        assert len(packed) == want
    for c in KAT["packed_bytes"]["cases"]:
        assert O.pack_bases(b(c["seq"]))[0].hex() == c["hex"]


def test_append_semantics():
    a = KAT["append_semantics"]
    p1, n1 = O.pack_bases(b(a["first"]))
    p2, n2 = O.pack_bases(b(a["second"]))
    assert n1 == [] and n2 == a["second_npos"] and len(p1 + p2) == a["total_packed"]
    assert O.pack_bases(b"") == (b"", [])


def test_roundtrip_base_sequences():
    # sequence_test.go:87-111
    for s in [b"N", b"A", b"ACGT" * 25, b"ACGTN" * 20, b"NNNNACGT"]:
        p, n = O.pack_bases(s)
        assert O.unpack_bases(p, n, len(s)) == s


def test_container_headers():
    # container_test.go:11-104
    fh = O.FileHeader(2, 100000, 0)
    buf = bytearray(10)
    O.lib().fqzo_write_file_header(C.byref(fh), (C.c_uint8 * 10).from_buffer(buf))
    assert bytes(buf).hex() == KAT["sample_fq"]["file_header_hex"]
    got = O.FileHeader()
    assert O.lib().fqzo_read_file_header(bytes(buf), 10, C.byref(got)) == 0
    assert (got.version, got.block_size, got.flags) == (2, 100000, 0)
    bad = b"XYZ\x00" + bytes(6)
    r = O.lib().fqzo_read_file_header(bad, 10, C.byref(got))
    assert r < 0 and b"invalid magic" in O.lib().fqzo_strerror(r)
    bh = O.BlockHeader(1000, 100, 200, 50, 7, 10, 20, 15000, 15001)
    out = bytearray(36)
    assert O.lib().fqzo_write_block_header(C.byref(bh), 2, (C.c_uint8 * 36).from_buffer(out)) == 36
    back = O.BlockHeader()
    assert O.lib().fqzo_read_block_header(bytes(out), 36, 2, C.byref(back)) == 36
    assert [getattr(back, f) for f, _ in O.BlockHeader._fields_] == [1000, 100, 200, 50, 7, 10, 20, 15000, 15001]
    out1 = bytearray(32)
    assert O.lib().fqzo_write_block_header(C.byref(bh), 1, (C.c_uint8 * 32).from_buffer(out1)) == 32
    assert O.lib().fqzo_read_block_header(bytes(out1), 32, 1, C.byref(back)) == 32
    assert back.plus_size == 0 and back.npos_size == 10 and back.original_qual_size == 15001
    assert O.lib().fqzo_write_block_header(C.byref(bh), 4, (C.c_uint8 * 36).from_buffer(out)) < 0  # (3 is FQZ-R1: the version-2 header)


def test_sample_fq_streams(sample_fq):
    g = KAT["sample_fq"]
    recs, n = O.parse_all(sample_fq)
    assert n == 3
    assert [recs[i].hdr_len for i in range(3)] == [25, 8, 21]
    quals = [sample_fq[recs[i].qual_off: recs[i].qual_off + recs[i].qual_len] for i in range(3)]
    enc = O.detect_encoding(quals)
    assert enc == 0
    streams, orig = O.split_block(sample_fq, recs, n, enc)
    assert orig == (180, 180)
    for k, name in enumerate(O.STREAM_NAMES):
        assert len(streams[k]) == g["lens"][name], name
        assert hashlib.sha256(streams[k]).hexdigest()[:16] == g["sha256_16"][name], name
    assert streams[0][:15].hex() == g["rec1_seq_hex"]
    assert streams[0][30:45].hex() == g["rec3_seq_hex"]
    assert streams[1].hex().startswith(g["qual_prefix_hex"])
    assert streams[4].hex() == g["npos_hex"]
    assert O.join_block(streams, 3, enc) == sample_fq


def test_parser_quirks():
    # parser_test.go:13-27,64-95,122-132 and SURVEY App. B
    recs, n = O.parse_all(b"@r1 x\nACGT\n+r1 x\nIIII\n")
    assert n == 1 and recs[0].hdr_len == 4 and recs[0].plus_len == 4
    with pytest.raises(O.OracleError, match="header line must start with @"):
        O.parse_all(b"r1\nACGT\n+\nIIII\n")
    with pytest.raises(O.OracleError, match="separator line must start with \\+"):
        O.parse_all(b"@r1\nACGT\n-\nIIII\n")
    with pytest.raises(O.OracleError, match="lengths must match"):
        O.parse_all(b"@r1\nACGT\n+\nIII\n")
    # CRLF is stripped (App. B-2)
    recs, n = O.parse_all(b"@r1\r\nACGT\r\n+\r\nIIII\r\n")
    assert n == 1 and recs[0].seq_len == 4 and recs[0].hdr_len == 2
    # unterminated last record is dropped silently (App. B-3)
    recs, n = O.parse_all(b"@r1\nACGT\n+\nIIII\n@r2\nAC\n+\nII")
    assert n == 1
    # blank trailing line is a hard error (App. B-3)
    with pytest.raises(O.OracleError, match="header line must start with @"):
        O.parse_all(b"@r1\nACGT\n+\nIIII\n\n")
    assert O.parse_all(b"")[1] == 0


def test_roundtrip_texts():
    for text in KAT["roundtrip_texts"]["cases"]:
        t = b(text)
        for w in (1, 4):
            z = O.compress(t, workers=w)
            assert O.decompress(z, workers=w) == t
    assert O.compress(b"") == bytes.fromhex(KAT["sample_fq"]["file_header_hex"])  # App. B-7
    assert O.decompress(O.compress(b"")) == b""


def test_roundtrip_generated_batches():
    # compress_test.go:125-158, 198-229, 381-447: 500-1000 records of 152 bp
    seq, qual = b"ACGT" * 38, b"I" * 152
    t = b"".join(b"@SEQ_" + bytes([65 + i % 26]) + b"\n" + seq + b"\n+\n" + qual + b"\n" for i in range(1000))
    assert O.decompress(O.compress(t)) == t
    t = b"".join(b"@SEQ_%d\n%s\n+\n%s\n" % (i, seq, b"efgh" * 38) for i in range(500))
    z = O.compress(t, block_size=100, workers=4)
    assert z[9] == 2  # FlagPhred64
    assert O.decompress(z, workers=4) == t
    # multi-block, out-of-order completion (the gap noted in SURVEY §4)
    z = O.compress(t, workers=4, batch_records=37)
    assert O.decompress(z, workers=3) == t
    assert z[:10] == O.compress(t, workers=1, batch_records=37)[:10]
    assert z == O.compress(t, workers=1, batch_records=37)


def test_long_read_guard():
    # compress_test.go:651-697
    seq = bytearray(b"ACGT" * 17500)
    seq[66000] = ord("N")
    t = b"@SEQ_LONG\n" + bytes(seq) + b"\n+\n" + b"I" * 70000 + b"\n"
    with pytest.raises(O.OracleError, match="ambiguous bases beyond position"):
        O.compress(t)
    seq = bytearray(b"ACGT" * 17500)
    seq[100] = ord("N")
    t = b"@SEQ_LONG\n" + bytes(seq) + b"\n+\n" + b"I" * 70000 + b"\n"
    assert O.decompress(O.compress(t)) == t


def test_v1_container_decodes():
    # compress_test.go:502-592: hand-built v1 file, bare '+' line on output
    text = b"@SEQ_1\nACGTACGT\n+\nIIIIIIII\n"
    recs, n = O.parse_all(text)
    streams, orig = O.split_block(text, recs, 1, 0)
    comp = [O.entropy_encode(streams[k]) for k in (0, 1, 2, 4, 5)]
    fh = bytes.fromhex("46515a00") + bytes([1]) + (1).to_bytes(4, "little") + b"\x00"
    bh = b"".join(x.to_bytes(4, "little") for x in [1] + [len(c) for c in comp] + [8, 8])
    assert O.decompress(fh + bh + b"".join(comp)) == text
    # unsupported version
    with pytest.raises(O.OracleError, match="unsupported file version"):
        O.decompress(bytes.fromhex("46515a00") + bytes([4]) + bytes(5))
    # truncated block header / payload
    z = O.compress(text)
    with pytest.raises(O.OracleError):
        O.decompress(z[:-1])
    with pytest.raises(O.OracleError):
        O.decompress(z[:20])


def test_zero_length_reads_and_lossy_bases():
    t = b"@e\n\n+\n\n@f\nAC\n+\nII\n"  # App. B-8
    assert O.decompress(O.compress(t)) == t
    t = b"@x\nacgtnRY.\n+\nIIIIIIII\n"  # App. B-1: lossy normalisation
    assert O.decompress(O.compress(t)) == b"@x\nACGTNNNN\n+\nIIIIIIII\n"
