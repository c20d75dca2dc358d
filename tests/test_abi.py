"""CPU-side checks of the drop-in boundary: libfqzhip.so loads, exports every symbol that
include/fqz.h declares, binds no torch types, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "fqz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fqz_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import fastqpacker_amd as fq
    L = fq.lib()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), "include/fqz.h declares %s but libfqzhip.so does not export it" % n
    # and the Python binding table covers the same set
    from fastqpacker_amd import _lib
    assert sorted(_lib.SIGNATURES) == names


def test_header_has_no_torch_or_cxx_types():
    src = open(os.path.join(ROOT, "include", "fqz.h")).read()
    assert "torch" not in src and "at::" not in src and "std::" not in src
    assert 'extern "C"' in src


def test_container_framing_is_byte_exact_on_host():
    from fastqpacker_amd import fqformat
    assert fqformat.WriteFileHeader(2, 100000, 0).hex() == "46515a0002a086010000"
    assert fqformat.ReadFileHeader(bytes.fromhex("46515a0002a086010002")) == (2, 100000, 2)
    import fastqpacker_amd as fq
    with pytest.raises(fq.FqzError, match="invalid magic"):
        fqformat.ReadFileHeader(b"XYZ\x00" + bytes(6))
    f = [1000, 100, 200, 50, 7, 10, 20, 15000, 15001]
    b2 = fqformat.WriteBlockHeader(f, 2)
    assert len(b2) == 36 and fqformat.ReadBlockHeader(b2, 2) == (f, 36)
    b1 = fqformat.WriteBlockHeader(f, 1)
    got, n = fqformat.ReadBlockHeader(b1, 1)
    assert n == 32 and got[4] == 0 and got[5] == 10
    with pytest.raises(fq.FqzError, match="unsupported block header version"):
        fqformat.WriteBlockHeader(f, 4)
    # same bytes as the oracle's framing
    import oracle_lib as O
    bh = O.BlockHeader(*f)
    out = bytearray(36)
    O.lib().fqzo_write_block_header(C.byref(bh), 2, (C.c_uint8 * 36).from_buffer(out))
    assert bytes(out) == b2


def test_no_cpu_fallback_without_gpu():
    import fastqpacker_amd as fq
    if fq.lib().fqz_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(fq.FqzError, match="no HIP device"):
        fq.Ctx(0)


def test_synth_generator_is_deterministic_and_parses():
    from fastqpacker_amd import compress
    import oracle_lib as O
    a, n = compress.synth_fastq(500)
    b, _ = compress.synth_fastq(250, first_record=250)
    assert n == 500 and a.tobytes().endswith(b.tobytes())
    recs, cnt = O.parse_all(a.tobytes())
    assert cnt == 500 and all(recs[i].seq_len == 150 for i in range(cnt))
    assert 345 < a.size / 500 < 356
    c, n = compress.synth_fastq(300, min_len=35, max_len=301, n_permille=50, phred=64, quality_profile=1)
    t = c.tobytes()
    recs, cnt = O.parse_all(t)
    assert cnt == 300
    quals = [t[recs[i].qual_off: recs[i].qual_off + recs[i].qual_len] for i in range(cnt)]
    assert O.detect_encoding(quals) == 1
    frac_n = t.count(b"N") / sum(recs[i].seq_len for i in range(cnt))
    assert 0.02 < frac_n < 0.09
    assert O.decompress(O.compress(t)) == t
