"""Multi-GPU placement of independent blocks.

Blocks are independent and the container is their concatenation in block order
(compress.go:365-403 is the reference's ordered collector).  Each rank encodes a contiguous
range of blocks; the only exchange is an all-gather of per-block compressed sizes whose exclusive
prefix sum gives every block's absolute offset in the output file.  `torch.distributed` is used
for the collective: backend "nccl" is RCCL over xGMI on MI355X, "gloo" on CPU for tests.
"""
import torch
import torch.distributed as dist

FILE_HEADER_SIZE = 10


def block_offsets_allgather(local_block_lens, max_blocks, device=None, group=None):
    """local_block_lens: list[int] of this rank's compressed block sizes (in block order).
    Returns (offsets of the local blocks in the output file, total file size, all sizes [world, max_blocks])."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(local_block_lens)
    if n > max_blocks:
        raise ValueError("more local blocks than max_blocks")
    mine = torch.zeros(max_blocks, dtype=torch.int64, device=device)
    if n:
        mine[:n] = torch.as_tensor(local_block_lens, dtype=torch.int64).to(mine.device)
    if world > 1:
        allsz = torch.empty(world * max_blocks, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allsz, mine, group=group)
    else:
        allsz = mine.clone()
    allsz = allsz.view(world, max_blocks)
    flat = allsz.reshape(-1)
    excl = torch.cumsum(flat, 0) - flat + FILE_HEADER_SIZE
    start = rank * max_blocks
    return excl[start:start + n].tolist(), int(flat.sum().item()) + FILE_HEADER_SIZE, allsz


def broadcast_encoding(encoding, src=0, device=None, group=None):
    """The quality encoding is a property of the FILE: the reference detects it once, on the first batch, and writes one
    FlagPhred64 for all blocks (compress.go:146-164).  Rank `src` (the rank that encodes block 0) detects it
    (qual_encoding = FQZ_DETECT_ENCODING on its first batch, result in fqz_batch_result.qual_encoding); every other rank
    passes the broadcast value explicitly.  A shard that detected for itself could normalise with another offset than
    the file header states, and the file would decode to wrong qualities without any error."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(encoding)
    t = torch.tensor([int(encoding)], dtype=torch.int32, device=device)
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def shard_records(total_records, records_per_block, rank, world):
    """Contiguous block ranges per rank: returns (first_record, n_records) for `rank`."""
    n_blocks = (total_records + records_per_block - 1) // records_per_block
    per = (n_blocks + world - 1) // world
    b0 = min(n_blocks, rank * per)
    b1 = min(n_blocks, b0 + per)
    r0 = b0 * records_per_block
    r1 = min(total_records, b1 * records_per_block)
    return r0, max(0, r1 - r0)
