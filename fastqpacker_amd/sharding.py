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


class OffsetExchange:
    """The same exchange with every buffer allocated once (the per-step form `bench.py` times): fill() copies this rank's block
    sizes into a pinned staging row, run() moves it to the device, all-gathers and prefix-sums there, and returns device
    tensors - nothing in a step allocates, builds Python lists or waits for the host."""

    def __init__(self, max_blocks, world, device=None, group=None):
        self.max_blocks, self.world, self.group = int(max_blocks), int(world), group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        on_gpu = device is not None and torch.device(device).type == "cuda"
        self.host = torch.zeros(self.max_blocks, dtype=torch.int64)
        if on_gpu:
            self.host = self.host.pin_memory()
        self.mine = torch.zeros(self.max_blocks, dtype=torch.int64, device=device)
        self.all = torch.zeros(self.world * self.max_blocks, dtype=torch.int64, device=device)
        self.n = 0

    def fill(self, lens, at=0):
        """lens: a sequence / numpy array of this rank's compressed block sizes, written at position `at` of the row."""
        n = len(lens)
        if at + n > self.max_blocks:
            raise ValueError("more local blocks than max_blocks")
        if n:
            self.host[at:at + n] = torch.as_tensor(lens, dtype=torch.int64)
        self.n = at + n
        return self.n

    def exchange(self):
        """-> (offsets of the local blocks [n], file size [1]) as device tensors, all sizes [world, max_blocks]"""
        self.host[self.n:] = 0
        self.mine.copy_(self.host, non_blocking=True)
        if self.world > 1:
            dist.all_gather_into_tensor(self.all, self.mine, group=self.group)
        else:
            self.all.copy_(self.mine)
        excl = torch.cumsum(self.all, 0) - self.all + FILE_HEADER_SIZE
        start = self.rank * self.max_blocks
        return excl[start:start + self.n], self.all.sum() + FILE_HEADER_SIZE, self.all.view(self.world, self.max_blocks)

    def run(self, lens):
        """fill + exchange, results on the host: (offsets list, total)"""
        self.fill(lens)
        offs, total, _ = self.exchange()
        return offs.tolist(), int(total.item())


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
