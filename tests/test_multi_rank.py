"""N>1 path on CPU: world_size-2 gloo run of the block-offset exchange, checked against the oracle's
single-process file (each rank 'encodes' its shard with the oracle; the assembled file must be identical)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from fastq_gen import make_fastq

RPB = 50


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, text, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastqpacker_amd.sharding import block_offsets_allgather, shard_records, broadcast_encoding
    recs, n = O.parse_all(text)
    r0, cnt = shard_records(n, RPB, rank, world)
    start = recs[r0].hdr_off - 1 if cnt else len(text)
    end = (recs[r0 + cnt].hdr_off - 1) if r0 + cnt < n else len(text)
    shard = text[start:end]
    # the encoding is the file's: rank 0 (block 0) detects it, the others take the broadcast value instead of detecting on
    # their own shard (shard 1 of the test text holds only qualities >= '@' and would pick Phred+64 for itself)
    mine = 0
    if rank == 0:
        quals = [shard[r.qual_off:r.qual_off + r.qual_len] for r in O.parse_all(shard)[0][:RPB]]
        mine = O.detect_encoding(quals)
    enc = broadcast_encoding(mine, src=0)
    fqz = O.compress(shard, batch_records=RPB, force_encoding=1 + enc)[10:] if cnt else b""
    # split the shard's body into blocks by walking the headers
    lens, pos = [], 0
    while pos < len(fqz):
        sizes = [int.from_bytes(fqz[pos + 4 * i: pos + 4 * i + 4], "little") for i in range(9)]
        blk = 36 + sum(sizes[1:7])
        lens.append(blk)
        pos += blk
    offs, total, allsz = block_offsets_allgather(lens, max_blocks=16)
    # positional writes into a shared file (pwrite in any order)
    fd = os.open(out_path, os.O_RDWR)
    pos = 0
    for o, l in zip(offs, lens):
        os.pwrite(fd, fqz[pos:pos + l], o)
        pos += l
    if rank == 0:
        os.pwrite(fd, O.compress(text, batch_records=RPB)[:10], 0)
    os.close(fd)
    dist.barrier()
    if rank == 0:
        assert total == len(O.compress(text, batch_records=RPB))
        assert int(allsz.sum()) + 10 == total
    dist.destroy_process_group()


def test_two_rank_offset_exchange(tmp_path):
    a = make_fastq(250, seed=12, min_len=60, max_len=150, n_frac=0.01)              # shard 0: Phred+33 qualities from '!' up
    b = make_fastq(180, seed=13, min_len=60, max_len=150, n_frac=0.01, phred=64)    # shard 1: every quality byte >= '@'
    text = a + b
    assert O.detect_encoding([q for q in b.split(b"\n")[3::4]]) == 1 and O.compress(text)[9] == 0
    want = O.compress(text, batch_records=RPB)
    path = str(tmp_path / "out.fqz")
    with open(path, "wb") as f:
        f.truncate(len(want))
    mp.spawn(_worker, args=(2, _free_port(), text, path), nprocs=2, join=True)
    got = open(path, "rb").read()
    assert got == want
    assert O.decompress(got) == text


def test_shard_records_covers_everything():
    from fastqpacker_amd.sharding import shard_records
    for total in (0, 1, 99, 100, 101, 1234):
        for world in (1, 2, 3, 8):
            spans = [shard_records(total, 100, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == total
            pos = 0
            for r0, c in spans:
                if c:
                    assert r0 == pos
                    pos += c
                    assert r0 % 100 == 0


def test_bench_gpus2_starts_two_ranks():
    """`bench.py --gpus 2` must start two ranks (launcher -> torchrun -> ranks) and rank 0 must print n_gpus: 2.  --dry-ranks
    keeps the codec and the GPU out of it (gloo): what is proven is the launch path the driver's SCALE run depends on."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-ranks"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["dry_ranks"] is True and out["sharded_file_layout_ok"] is True


def test_offset_exchange_matches_allgather_single_rank():
    from fastqpacker_amd.sharding import OffsetExchange, block_offsets_allgather
    lens = [5, 7, 11]
    ex = OffsetExchange(8, 1)
    offs, total = ex.run(lens)
    o2, t2, _ = block_offsets_allgather(lens, 8)
    assert offs == o2 == [10, 15, 22] and total == t2 == 33
