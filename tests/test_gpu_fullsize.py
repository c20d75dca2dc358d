"""Full-size parity: the default bench batch (29 blocks of 100 000 reads, ~1 GB) and BASELINE config 5's
shape at >= 10 blocks are compared byte for byte with the oracle's container - not just round-tripped - and the sharded
(multi-GPU) code path is driven with the HIP encoder at world size 1."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fq():
    import fastqpacker_amd as fq
    fq.lib()
    return fq


def _encode_dev(fq, text_np, enc, version=2, extra_flags=0):
    import torch
    from fastqpacker_amd._lib import BatchResult, lib
    dev = torch.device("cuda:0")
    d_text = torch.from_numpy(text_np).to(dev)
    d_out = torch.empty(int(lib().fqz_encode_bound(text_np.size)) // 2 + (1 << 20), dtype=torch.uint8, device=dev)
    ctx = fq.Ctx(0)
    res = BatchResult()
    nb = text_np.size // 200 // fq.DEFAULT_BLOCK_SIZE + 8
    offs, lens = (C.c_uint64 * nb)(), (C.c_uint64 * nb)()
    fq._lib.check(lib().fqz_encode_batch_dev(ctx.handle, d_text.data_ptr(), text_np.size, fq.DEFAULT_BLOCK_SIZE, enc,
                                             fq.BATCH_FINAL | (fq.BATCH_V3 if version == 3 else 0) | extra_flags,
                                             d_out.data_ptr(), d_out.numel(), C.byref(res), offs, lens, nb, None))
    body = d_out[: int(res.out_len)].cpu().numpy()
    # and back on the device
    d_back = torch.empty(text_np.size + 4096, dtype=torch.uint8, device=dev)
    dres = BatchResult()
    fq._lib.check(lib().fqz_decode_batch_dev(ctx.handle, d_out.data_ptr(), int(res.out_len), version, enc, d_back.data_ptr(), d_back.numel(), C.byref(dres), None))
    back_ok = bool(dres.out_len == text_np.size and torch.equal(d_back[: text_np.size], d_text))
    return body, res, list(lens[: res.n_blocks]), back_ok


def _first_diff(a, b):
    n = min(a.size, b.size)
    d = np.flatnonzero(a[:n] != b[:n])
    return "sizes %d / %d, first difference at %s" % (a.size, b.size, d[0] if d.size else "-")


def test_default_bench_batch_is_byte_identical_to_the_oracle(fq):
    """BASELINE config 2 ('1 GB' reading): the whole batch the bench times, GPU container bytes == oracle container bytes."""
    from fastqpacker_amd import compress
    n_bytes = int(1e9)
    text, _ = compress.synth_fastq(n_bytes // 351 + 1, cap=n_bytes + 4096)
    text = text[:n_bytes]
    k = bytes(text[-4096:]).rfind(b"\n@SIM:")
    text = text[: text.size - 4096 + k + 1]
    body, res, lens, back_ok = _encode_dev(fq, text, fq.ENCODING_PHRED33)
    assert res.n_blocks == 29 and res.n_records == 2849002 and back_ok
    want = np.frombuffer(O.compress(text, workers=16), dtype=np.uint8)
    assert want[9] == 0                                                   # Phred+33 detected by the oracle too
    assert body.size + 10 == want.size and np.array_equal(body, want[10:]), _first_diff(body, want[10:])
    assert sum(lens) == body.size
    # the same batch as two halves in flight (FQZ_BATCH_HALVES: the second half starts at the block boundary the first one reports)
    body_h, res_h, lens_h, back_h = _encode_dev(fq, text, fq.ENCODING_PHRED33, extra_flags=fq.BATCH_HALVES)
    assert back_h and res_h.n_blocks == 29 and res_h.n_records == res.n_records and lens_h == lens and np.array_equal(body_h, body)
    assert list(res_h.stream_raw) == list(res.stream_raw) and list(res_h.stream_comp) == list(res.stream_comp)
    # the same batch as a version-3 container (FQZ-R1: rANS-coded qualities)
    body3, res3, lens3, back3 = _encode_dev(fq, text, fq.ENCODING_PHRED33, version=3)
    want3 = np.frombuffer(O.compress(text, workers=16, entropy=2), dtype=np.uint8)
    assert back3 and want3[4] == 3 and np.array_equal(body3, want3[10:]), _first_diff(body3, want3[10:])
    assert body3.size < 0.86 * body.size


def test_config5_shape_ten_blocks_is_byte_identical_to_the_oracle(fq):
    """BASELINE config 5's shape on one GPU: 35-301 bp, 5 % N in runs, Phred+64, > 10 blocks of 100 000 reads."""
    from fastqpacker_amd import compress
    text, n = compress.synth_fastq(1_050_000, min_len=35, max_len=301, n_permille=50, phred=64)
    assert n == 1_050_000
    body, res, lens, back_ok = _encode_dev(fq, text, fq.ENCODING_PHRED64)
    assert res.n_blocks == 11 and res.n_records == n and back_ok
    want = np.frombuffer(O.compress(text, workers=16), dtype=np.uint8)
    assert want[9] == 2                                                   # FlagPhred64: the oracle detected Phred+64 on block 0
    assert np.array_equal(body, want[10:]), _first_diff(body, want[10:])
    assert bytes(O.decompress(want[:10].tobytes() + body.tobytes(), workers=16)) == text.tobytes()
    body3, res3, lens3, back3 = _encode_dev(fq, text, fq.ENCODING_PHRED64, version=3)
    want3 = np.frombuffer(O.compress(text, workers=16, entropy=2), dtype=np.uint8)
    assert back3 and np.array_equal(body3, want3[10:]), _first_diff(body3, want3[10:])


def test_sharded_path_with_the_hip_encoder_at_world_size_one(fq, tmp_path):
    """The code a rank runs in the N-GPU job (sharding.shard_records -> encode its shard -> block_offsets_allgather -> pwrite
    at the gathered offsets), here with the HIP encoder and world size 1: the file must equal the single-call container."""
    import os
    from fastqpacker_amd import compress, sharding
    text, n = compress.synth_fastq(2500, min_len=100, max_len=151, n_permille=10)
    rpb = 300
    r0, cnt = sharding.shard_records(n, rpb, 0, 1)
    assert (r0, cnt) == (0, n)
    import torch
    from fastqpacker_amd._lib import BatchResult, lib
    dev = torch.device("cuda:0")
    d_text = torch.from_numpy(text).to(dev)
    d_out = torch.empty(int(lib().fqz_encode_bound_blocks(text.size, rpb)), dtype=torch.uint8, device=dev)
    res, offs, lens = compress.encode_batch_dev(d_text.data_ptr(), text.size, d_out.data_ptr(), d_out.numel(), records_per_block=rpb,
                                                qual_encoding=fq.DETECT_ENCODING, final=True, max_blocks=16)
    enc = sharding.broadcast_encoding(res.qual_encoding)                  # rank 0 detects; a no-op at world size 1
    file_offs, total, allsz = sharding.block_offsets_allgather(lens, max_blocks=16)
    path = str(tmp_path / "sharded.fqz")
    body = d_out[: res.out_len].cpu().numpy().tobytes()
    with open(path, "wb") as f:
        f.truncate(total)
    fd = os.open(path, os.O_RDWR)
    for o, l, fo in zip(offs, lens, file_offs):                            # any order: positional writes
        os.pwrite(fd, body[o:o + l], fo)
    os.pwrite(fd, bytes.fromhex("46515a00") + bytes([2]) + (100000).to_bytes(4, "little") + bytes([2 if enc else 0]), 0)
    os.close(fd)
    got = open(path, "rb").read()
    assert got == O.compress(text.tobytes(), batch_records=rpb)
    assert compress.Decompress(got) == text.tobytes()
