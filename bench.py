#!/usr/bin/env python3
"""bench.py — the per-block FASTQ encode hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the whole encode pipeline (line index -> six streams -> Huffman-literal
zstd blocks -> framed .fqz blocks) over one batch of synthetic 150 bp Illumina FASTQ that is already
resident in HBM.  At N ranks every rank encodes its own shard of the same size (weak scaling);
the only exchange is the all-gather of per-block compressed sizes that turns into the
container's block offsets.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bytes B] [--profile 0|1] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bytes", type=float, default=1e9, help="FASTQ bytes per rank (config 2: synthetic 1 GB)")
    ap.add_argument("--reads", type=float, default=0, help="reads per rank instead of --bytes (config 2's other reading: 1e7 reads = 3.5 GB, run as "
                    "several device batches of whole 100k-read blocks back to back)")
    ap.add_argument("--profile", type=int, default=1, help="bracket kernels with HIP events (roofline.achieved)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--decode-steps", type=int, default=3)
    ap.add_argument("--quality-profile", type=int, default=0)
    ap.add_argument("--container", type=int, default=2, help="container version the timed encode writes: 2 = the reference's (default), "
                    "3 = FQZ-R1 (rANS-coded qualities, SURVEY 8 f-4; not readable by the stock decoder)")
    ap.add_argument("--no-v3", action="store_true", help="skip the supplementary container_v3 reading (profiling runs: one kind of launch per kernel)")
    ap.add_argument("--inflight", type=int, default=3, help="batches in flight for the supplementary pipelined figure (0 = skip)")
    ap.add_argument("--no-supp", action="store_true", help="skip the supplementary readings (config 2 as 10 M reads, config 5's shape)")
    ap.add_argument("--dry-ranks", action="store_true", help="CPU rehearsal of the N-rank path (gloo, no GPU, no codec): every rank runs the "
                    "launcher, the rendezvous, the block-offset all-gather and the max-over-ranks timing on made-up block sizes")
    return ap.parse_args()


def launch_ranks(a):
    """`bench.py --gpus N` (N > 1) outside a torchrun job: start N ranks, one per GPU, and relay rank 0's JSON line.  This process
    never imports torch.cuda or libfqzhip and never touches a device; the ranks are children of the launcher child
    (`python -m torch.distributed.run`), nothing re-execs a process that has initialised the GPU.  Stands where the reference
    starts its worker pool (internal/compress/compress.go:240-278); the ordered collector (:365-403) is the offset exchange."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    return p.returncode if p.returncode else (0 if line else 1)


def dry_ranks(a, rank, world):
    """--dry-ranks: the N-rank control path on CPU (gloo).  No codec runs, so the line carries no throughput."""
    import torch
    import torch.distributed as dist
    from fastqpacker_amd import sharding
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    enc = sharding.broadcast_encoding(1 if rank == 0 else 0, src=0)       # rank 0 "detected" Phred+64
    lens = [1000 + 13 * rank + k for k in range(3 + rank)]                # made-up compressed block sizes
    ex = sharding.OffsetExchange(8, world, device=None)
    for _ in range(a.warmup):
        ex.run(lens)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        offs, total = ex.run(lens)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    want_total = 10 + sum(1000 + 13 * r + k for r in range(world) for k in range(3 + r))
    want_first = 10 + sum(1000 + 13 * r + k for r in range(rank) for k in range(3 + r))
    ok = torch.tensor([1 if (total == want_total and offs[0] == want_first and enc == 1) else 0], dtype=torch.int32)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"metric": "encode MB/s (input FASTQ), 150 bp Illumina", "value": None, "unit": "MB/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": round(float(tt.item()) / max(1, a.steps) * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "none",
                          "config": {"workload": "dry ranks: launcher + gloo rendezvous + block-offset exchange on made-up sizes; no codec, no GPU"},
                          "dry_ranks": True, "sharded_file_layout_ok": bool(ok.item())}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(text_np, want_seconds=20.0):
    """Oracle (CPU restatement of the reference pipeline: one parser thread, W workers each owning an entropy context,
    ordered writer; entropy stage = libzstd level 1 when the system library is present) timed on the host cores over a
    bounded sample of the same workload, at W = all cores and at W = 1.  Protocol of
    scripts/benchmark_fqpack_9gb.sh:71-96: one verified run, then timed runs, mean."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O  # checker / baseline only
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    entropy = 1 if O.lib().fqzo_libzstd_version() else 0
    rec_bytes = 351

    def sample(blocks):
        n = min(text_np.size, blocks * 100000 * rec_bytes)
        cut = text_np[:n]
        tail = bytes(cut[-4096:])  # cut on a record boundary: every record of this workload starts with "@SIM:"
        k = tail.rfind(b"\n@SIM:")
        return cut[: n - len(tail) + k + 1]

    def timed(cut, workers, budget):
        z = O.compress(cut, workers=workers, entropy=entropy)            # the verified run
        ok = O.decompress(z, workers=workers) == bytes(cut)
        t0 = time.perf_counter()
        z = O.compress(cut, workers=workers, entropy=entropy)
        dt = time.perf_counter() - t0
        reps = 1
        if dt < budget / 4:
            reps = int(min(8, max(3, budget / 2 / max(dt, 1e-3))))
            t0 = time.perf_counter()
            for _ in range(reps):
                z = O.compress(cut, workers=workers, entropy=entropy)
            dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        O.decompress(z, workers=workers)
        ddt = time.perf_counter() - t0
        return cut.size / dt / 1e6, cut.size / ddt / 1e6, cut.size / len(z), ok, reps

    total_blocks = max(1, int(text_np.size // (rec_bytes * 100000)))
    cut_all = sample(total_blocks)                                      # every block of the batch: jobs >= cores needs blocks >= cores
    v_all, d_all, ratio, ok_all, reps_all = timed(cut_all, cores, want_seconds * 0.6)
    cut_one = sample(min(total_blocks, 3))                              # W = 1: ~100 MB keeps the leg within seconds
    v_one, d_one, _, ok_one, reps_one = timed(cut_one, 1, want_seconds * 0.4)
    # per-stream ratios of the CPU pipeline's file: compressed sizes from its block headers, pre-entropy sizes by
    # the stream layouts from the decoded record count and read lengths of the same sample
    stream_ratio = None
    try:
        z = O.compress(cut_all, workers=cores, entropy=entropy)
        comp = [0] * 6
        pos, nrec = 10, 0
        while pos + 36 <= len(z):
            h = [int.from_bytes(z[pos + 4 * i: pos + 4 * i + 4], "little") for i in range(9)]
            nrec += h[0]
            for i in range(6):
                comp[i] += h[1 + i]
            pos += 36 + sum(h[1:7])
        lines = cut_all.size  # raw sizes: seq ceil(L/4), qual L, headers 2 + H, plus 2 + P, nPos 2 + 2n (no N in this workload), lengths 4
        L = 150
        hdr_bytes = int(cut_all.size - nrec * (2 * L + 6))  # text = (H + 2) + (L + 1) + (P + 2) + (L + 1) per record, P = 0
        raw = [nrec * ((L + 3) // 4), nrec * L, hdr_bytes + 2 * nrec, 2 * nrec, 2 * nrec, 4 * nrec]
        names = ["seq", "qual", "headers", "plus", "npos", "lengths"]
        stream_ratio = {names[i]: (round(raw[i] / comp[i], 3) if comp[i] else None) for i in range(6)}
        del lines
    except Exception:
        pass
    ent = "libzstd-%d level 1 entropy stage" % O.lib().fqzo_libzstd_version() if entropy else "its own Huffman entropy stage"
    jobs = int(cut_all.size // (rec_bytes * 100000)) + 1
    return {
        "value": round(v_all, 1), "unit": "MB/s", "cores": cores, "kind": "port",
        "sample": "%d MB (%d blocks of 100k reads = %d block jobs for %d worker threads: %.0f %% of the cores can be busy) of the same "
                  "synthetic FASTQ, 1 verified + %d timed pass(es), oracle C pipeline (CPU restatement, not the reference Go binary) with %s"
                  % (cut_all.size // 1000000, jobs, jobs, cores, 100.0 * min(1.0, jobs / cores), reps_all, ent),
        "decode_MBps": round(d_all, 1), "ratio": round(ratio, 3), "stream_ratio": stream_ratio, "roundtrip_ok": bool(ok_all and ok_one),
        "w1": {"value": round(v_one, 1), "unit": "MB/s", "cores": 1, "decode_MBps": round(d_one, 1),
               "sample": "%d MB, 1 verified + %d timed pass(es), one worker thread" % (cut_one.size // 1000000, reps_one)},
    }


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))  # (before anything in this process imports torch.cuda or loads libfqzhip)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.dry_ranks:
        return dry_ranks(a, rank, world)
    import torch  # device memory, streams and torch.distributed only; imported before libfqzhip so both share one HIP runtime
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.zeros(1, device=dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import numpy as np
    import ctypes as C
    import fastqpacker_amd as fq
    from fastqpacker_amd import compress
    from fastqpacker_amd import sharding
    from fastqpacker_amd._lib import BatchResult, lib

    # ---- workload: config 2 of BASELINE.json, one shard per rank, as one or more device batches (< 2 GiB each) ----------
    RPB = fq.DEFAULT_BLOCK_SIZE
    batches = []  # (text as numpy, device tensor)
    if a.reads:
        total_reads = int(a.reads)
        per_batch = (int(1.9e9) // 351 // RPB) * RPB                      # whole 100k-read blocks per batch
        r0 = 0
        while r0 < total_reads:
            nr = min(per_batch, total_reads - r0)
            t_np, wrote = compress.synth_fastq(nr, first_record=rank * total_reads + r0, quality_profile=a.quality_profile)
            assert wrote == nr
            batches.append(t_np)
            r0 += nr
        workload = "synthetic 150 bp Phred+33 FASTQ, %d reads = %.2f GB per GPU (BASELINE.json configs[1] in its '10 M reads' reading), " \
                   "%d device batches of whole 100k-read blocks back to back" % (total_reads, sum(b.size for b in batches) / 1e9, len(batches))
    else:
        n_bytes = int(a.bytes)
        n_rec = n_bytes // 351 + 1
        t_np, wrote = compress.synth_fastq(n_rec, first_record=rank * n_rec, quality_profile=a.quality_profile, cap=n_bytes + 4096)
        t_np = t_np[: min(t_np.size, n_bytes)]
        k = bytes(t_np[-4096:]).rfind(b"\n@SIM:")                          # end on a record boundary
        batches.append(t_np[: t_np.size - 4096 + k + 1])
        workload = "synthetic 150 bp Phred+33 FASTQ, %.2f GB per GPU (BASELINE.json configs[1] in its '1 GB' reading), one device batch" \
                   % (batches[0].size / 1e9)
    workload += ", device-resident, 100k-record blocks, quality profile %d" % a.quality_profile
    if a.container == 3:
        workload += ", container version 3 (FQZ-R1: rANS-coded qualities)"
    enc_flags = fq.BATCH_FINAL | (fq.BATCH_V3 if a.container == 3 else 0)
    d_texts = [torch.from_numpy(b).to(dev) for b in batches]
    d_outs = [torch.empty(int(lib().fqz_encode_bound(b.size)) // 2 + (1 << 20), dtype=torch.uint8, device=dev) for b in batches]
    in_bytes = int(sum(b.size for b in batches))
    ctx = fq.Ctx(local_rank)
    stream = torch.cuda.current_stream(dev)
    sptr = C.c_void_p(stream.cuda_stream)
    max_blocks = max(b.size for b in batches) // (351 * RPB) + 8
    offs = [(C.c_uint64 * max_blocks)() for _ in batches]
    lens = [(C.c_uint64 * max_blocks)() for _ in batches]
    ress = [BatchResult() for _ in batches]
    world_blocks = max_blocks * len(batches)
    exch = sharding.OffsetExchange(world_blocks, world, device=dev) if world > 1 else None
    lens_np = [np.ctypeslib.as_array(l) for l in lens]

    def encode_step():
        for i, b in enumerate(batches):
            fq._lib.check(lib().fqz_encode_batch_dev(ctx.handle, d_texts[i].data_ptr(), b.size, RPB, fq.ENCODING_PHRED33,
                                                     enc_flags, d_outs[i].data_ptr(), d_outs[i].numel(), C.byref(ress[i]), offs[i], lens[i],
                                                     max_blocks, sptr))
        if world > 1:
            # container index: all-gather of per-block compressed sizes -> exclusive prefix = file offsets (RCCL over xGMI);
            # the function the world-size-2 gloo test covers
            at = 0
            for i in range(len(batches)):
                at = exch.fill(lens_np[i][: ress[i].n_blocks], at)
            return exch.exchange()  # device tensors: nothing here allocates or waits for the host
        return None

    for _ in range(a.warmup):
        encode_step()
    if a.profile:
        ctx.profile(3 if a.container == 3 else 2)  # HIP events around the dominant kernel only: two records per step, nothing else perturbs the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        placed = encode_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern = ctx.profile_read() if a.profile else {}
    ctx.profile(False)
    launches_per_step = len(batches)
    if a.profile:  # per-kernel breakdown of the rest of the pipeline: a separate, untimed pass with every kernel bracketed
        ctx.profile(1)
        for _ in range(3):
            encode_step()
        torch.cuda.synchronize()
        allk = ctx.profile_read()
        ctx.profile(False)
        for k, v in allk.items():
            if k not in kern:
                kern[k] = (v[0] / max(1, v[1]) * a.steps * launches_per_step, a.steps * launches_per_step)  # scaled to the timed region
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out_bytes = int(sum(int(r.out_len) for r in ress))
    n_records = int(sum(int(r.n_records) for r in ress))
    n_blocks = int(sum(int(r.n_blocks) for r in ress))
    total_in = torch.tensor([in_bytes], dtype=torch.float64, device=dev)
    total_out = torch.tensor([out_bytes], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(total_in)
        dist.all_reduce(total_out)
    total_in, total_out = float(total_in.item()), float(total_out.item())

    # ---- N > 1: the file the ranks would write (blocks at their all-gathered offsets) is checked once, outside the timed
    #      region: block sizes of every rank, offsets = their exclusive prefix sum + 10, and the total
    sharded_ok = None
    if world > 1:
        my_offs, file_total, allsz = placed
        my_offs, file_total = my_offs.tolist(), int(file_total.item())
        mine = [int(x) for i in range(len(batches)) for x in lens[i][: ress[i].n_blocks]]
        ok = int(file_total) == int(total_out) + 10 and len(my_offs) == len(mine)
        flat = allsz.reshape(-1).tolist()
        run = 10
        for r in range(world):
            for k in range(world_blocks):
                if r == rank and k < len(mine):
                    ok = ok and my_offs[k] == run and mine[k] == flat[r * world_blocks + k]
                run += flat[r * world_blocks + k]
        tok = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(tok, op=dist.ReduceOp.MIN)
        sharded_ok = bool(tok.item())

    # ---- bit-exact gate + decode rate (rank-local) ------------------------------------------------
    fqz_devs = [d_outs[i][: int(ress[i].out_len)].clone() for i in range(len(batches))]
    d_backs = [torch.empty(b.size + 4096, dtype=torch.uint8, device=dev) for b in batches]
    dress = [BatchResult() for _ in batches]
    dkern = {}
    torch.cuda.synchronize()  # (the copies above ran on torch's stream; the library decodes on its own non-blocking stream)

    def decode_step():
        for i in range(len(batches)):
            # (with the block offsets the encode reported: whoever holds a batch of blocks knows where they start - the reference's
            #  reader takes them header by header, compress.go:721-758 - and the device need not walk the chain of headers again)
            fq._lib.check(lib().fqz_decode_batch_dev_hint(ctx.handle, fqz_devs[i].data_ptr(), fqz_devs[i].numel(), a.container, fq.ENCODING_PHRED33,
                                                          d_backs[i].data_ptr(), d_backs[i].numel(), C.byref(dress[i]), offs[i], int(ress[i].n_blocks), sptr))
    try:
        decode_step()
        roundtrip_ok = all(bool(dress[i].out_len == batches[i].size and torch.equal(d_backs[i][: batches[i].size], d_texts[i]))
                           for i in range(len(batches)))
        if a.profile:
            ctx.profile(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.decode_steps):
            decode_step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ddt = (time.perf_counter() - t1) / max(1, a.decode_steps)
        dkern = ctx.profile_read() if a.profile else {}
        ctx.profile(False)
    except fq.FqzError as e:  # only reachable in the FQZ_DBG_STOP timing experiments (garbage blocks)
        roundtrip_ok, ddt, dkern = False, float("inf"), {}
        print("decode failed: %s" % e, file=sys.stderr)
    if world > 1:  # whole-job decode figure: every rank's shard, the slowest rank's time; the round trip must hold on every rank
        dd = torch.tensor([ddt if ddt != float("inf") else 1e30], dtype=torch.float64, device=dev)
        dist.all_reduce(dd, op=dist.ReduceOp.MAX)
        ddt = float(dd.item())
        rt = torch.tensor([1 if roundtrip_ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(rt, op=dist.ReduceOp.MIN)
        roundtrip_ok = bool(rt.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel ------------------------------------------------------------
    ms_per_step = dt / a.steps * 1e3
    roof = None
    kernels = {}
    if kern:
        kernels = {k: round(v[0] / max(1, v[1]), 4) for k, v in kern.items()}  # avg ms per launch
        # the dominant kernel = the one bracketed live inside the timed region (profile mode 2: k_entropy, the largest by chip work).
        # Kernels that run BESIDE it on side streams (the headers' serial chains) can show a longer elapsed time in the
        # untimed breakdown pass: that is latency under a saturated chip, not work, and not what the roofline is about.
        dom = "k_rans" if a.container == 3 and "k_rans" in kern else ("k_entropy" if "k_entropy" in kern else max(kern, key=lambda k: kern[k][0]))
        avg_s = kern[dom][0] / kern[dom][1] / 1e3
        algorithmic = (in_bytes + out_bytes) / launches_per_step  # B_in + B_out per launch
        ach = algorithmic / avg_s / 1e9
        # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
        # separately; summary committed under profiles/): static evidence, not re-measured in this process
        traffic = None
        for rnd in ("r03", "r02", "r01"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_hbm_traffic.json")))
                if abs(in_bytes / launches_per_step - pmc.get("batch_bytes", 0)) <= 0.02 * in_bytes and dom in pmc["kernels"]:
                    traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
                    break
            except Exception:
                pass
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(algorithmic), "avg_launch_ms": round(avg_s * 1e3, 4),
                "pipeline_GBps": round((in_bytes + out_bytes) / (ms_per_step / 1e3) / 1e9, 1)}
    res = ress[0]
    sraw = [sum(int(r.stream_raw[i]) for r in ress) for i in range(6)]
    scomp = [sum(int(r.stream_comp[i]) for r in ress) for i in range(6)]
    out = {
        "metric": "encode MB/s (input FASTQ), 150 bp Illumina", "value": round(total_in / dt * a.steps / 1e6, 1), "unit": "MB/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": workload, "bytes_per_gpu": in_bytes, "records_per_gpu": n_records, "blocks_per_gpu": n_blocks,
                   "device_batches_per_step": launches_per_step},
        "ratio": round(total_in / total_out, 3),
        "decode_MBps": round(total_in / ddt / 1e6, 1), "roundtrip_bit_exact": roundtrip_ok,
        "input_frac_of_hbm_peak": round(total_in / dt * a.steps / 1e9 / (HBM_PEAK_GBS * world), 4),
        "roofline": roof, "kernel_ms": kernels,
        "decode_kernel_ms": {k: round(v[0] / max(1, v[1]), 4) for k, v in dkern.items()},
        "stream_ratio": {n: (round(sraw[i] / scomp[i], 3) if scomp[i] else None) for i, n in enumerate(fq.STREAM_NAMES)},
    }
    if sharded_ok is not None:
        out["sharded_file_layout_ok"] = sharded_ok
    # ---- supplementary: several batches in flight on separate contexts / streams (what a streaming compressor does).
    # The headline `value` above stays the single-stream figure BASELINE.json's config asks for; kernel times there are
    # undisturbed.  Here the kernels of different batches overlap (tails of one fill with work of the next).
    if a.inflight > 1 and world == 1 and len(batches) == 1:
        try:
            nc = a.inflight
            d_text, d_out, text_np = d_texts[0], d_outs[0], batches[0]
            pctx = [fq.Ctx(local_rank) for _ in range(nc)]
            pouts = [torch.empty_like(d_out) for _ in range(nc)]
            pstreams = [torch.cuda.Stream(dev) for _ in range(nc)]
            pres = [BatchResult() for _ in range(nc)]
            busy = [False] * nc

            def p_finish(i):
                if busy[i]:
                    fq._lib.check(lib().fqz_encode_batch_finish(pctx[i].handle, C.byref(pres[i]), None, None, 0))
                    busy[i] = False

            def p_launch(i):
                fq._lib.check(lib().fqz_encode_batch_launch(pctx[i].handle, d_text.data_ptr(), text_np.size, RPB, fq.ENCODING_PHRED33,
                                                            enc_flags, pouts[i].data_ptr(), pouts[i].numel(), C.c_void_p(pstreams[i].cuda_stream)))
                busy[i] = True

            psteps = max(2 * nc, a.steps)
            for rep in range(2):  # first round warms the contexts up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(psteps):
                    p_finish(k % nc)
                    p_launch(k % nc)
                for i in range(nc):
                    p_finish(i)
                torch.cuda.synchronize()
                pdt = time.perf_counter() - t0
            same = all(int(r.out_len) == out_bytes for r in pres) and all(torch.equal(o[:out_bytes], d_out[:out_bytes]) for o in pouts)
            out["pipelined"] = {"batches_in_flight": nc, "steps": psteps, "value": round(in_bytes * psteps / pdt / 1e6, 1), "unit": "MB/s",
                                "ms_per_step": round(pdt / psteps * 1e3, 3), "same_bytes_as_single_stream": bool(same)}
            del pctx, pouts
        except Exception as e:
            out["pipelined"] = {"error": repr(e)}
    # ---- supplementary: the same batch as a version-3 container (FQZ-R1, SURVEY 8 f-4: the qualities in interleaved rANS
    # blocks).  Not the headline: the stock decoder does not read it.
    if a.container == 2 and world == 1 and len(batches) == 1 and not a.no_v3:
        try:
            r3, d3 = BatchResult(), BatchResult()
            d_out3 = torch.empty_like(d_outs[0])

            def enc3():
                fq._lib.check(lib().fqz_encode_batch_dev(ctx.handle, d_texts[0].data_ptr(), batches[0].size, RPB, fq.ENCODING_PHRED33,
                                                         fq.BATCH_FINAL | fq.BATCH_V3, d_out3.data_ptr(), d_out3.numel(), C.byref(r3), None, None, 0, sptr))

            def dec3():
                fq._lib.check(lib().fqz_decode_batch_dev(ctx.handle, d_out3.data_ptr(), int(r3.out_len), 3, fq.ENCODING_PHRED33,
                                                         d_backs[0].data_ptr(), d_backs[0].numel(), C.byref(d3), sptr))
            enc3()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                enc3()
            torch.cuda.synchronize()
            e3 = (time.perf_counter() - t0) / 5
            d_backs[0].zero_()
            dec3()
            ok3 = bool(d3.out_len == batches[0].size and torch.equal(d_backs[0][: batches[0].size], d_texts[0]))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                dec3()
            torch.cuda.synchronize()
            dd3 = (time.perf_counter() - t0) / 3
            out["container_v3"] = {"encode_MBps": round(in_bytes / e3 / 1e6, 1), "decode_MBps": round(in_bytes / dd3 / 1e6, 1),
                                   "ratio": round(in_bytes / int(r3.out_len), 3), "roundtrip_bit_exact": ok3,
                                   "quality_stream_ratio": round(int(r3.stream_raw[1]) / max(1, int(r3.stream_comp[1])), 3)}
            del d_out3
        except Exception as e:
            out["container_v3"] = {"error": repr(e)}
    # ---- supplementary: BASELINE config 2 in its '10 M reads' reading (3.5 GB: two device batches back to back) and
    # config 5's shape on one GPU (35-301 bp, 5 % N, Phred+64, 1 GB).  Same library calls as the headline; not the headline.
    if world == 1 and len(batches) == 1 and not a.no_supp and not a.reads and a.container == 2:
        supp = {}
        try:
            del d_backs, fqz_devs
            torch.cuda.empty_cache()

            def run_batches(texts, enc_id, steps=3):
                dts = [torch.from_numpy(t).to(dev) for t in texts]
                dos = [torch.empty(int(lib().fqz_encode_bound(t.size)) // 2 + (1 << 20), dtype=torch.uint8, device=dev) for t in texts]
                rs = [BatchResult() for _ in texts]

                def enc():
                    for i, t in enumerate(texts):
                        fq._lib.check(lib().fqz_encode_batch_dev(ctx.handle, dts[i].data_ptr(), t.size, RPB, enc_id, fq.BATCH_FINAL, dos[i].data_ptr(), dos[i].numel(),
                                                                 C.byref(rs[i]), None, None, 0, sptr))
                enc()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    enc()
                torch.cuda.synchronize()
                et = (time.perf_counter() - t0) / steps
                zs = [dos[i][: int(rs[i].out_len)].clone() for i in range(len(texts))]
                bk = [torch.empty(t.size + 4096, dtype=torch.uint8, device=dev) for t in texts]
                ds = [BatchResult() for _ in texts]
                torch.cuda.synchronize()  # (torch's copies before the library's streams read them)

                def dec():
                    for i in range(len(texts)):
                        fq._lib.check(lib().fqz_decode_batch_dev(ctx.handle, zs[i].data_ptr(), zs[i].numel(), 2, enc_id, bk[i].data_ptr(), bk[i].numel(), C.byref(ds[i]), sptr))
                dec()
                ok = all(bool(ds[i].out_len == texts[i].size and torch.equal(bk[i][: texts[i].size], dts[i])) for i in range(len(texts)))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    dec()
                torch.cuda.synchronize()
                dt2 = (time.perf_counter() - t0) / 2
                nb = sum(t.size for t in texts)
                return {"bytes": int(nb), "device_batches": len(texts), "encode_MBps": round(nb / et / 1e6, 1), "decode_MBps": round(nb / dt2 / 1e6, 1),
                        "ratio": round(nb / sum(int(r.out_len) for r in rs), 3), "roundtrip_bit_exact": bool(ok)}

            # config 2, '10 M reads': whole 100k-read blocks per device batch (< 2 GiB each)
            total_reads, per_batch, r0, texts = 10_000_000, (int(1.9e9) // 351 // RPB) * RPB, 0, []
            while r0 < total_reads:
                nr = min(per_batch, total_reads - r0)
                t_np, wrote = compress.synth_fastq(nr, first_record=r0, quality_profile=a.quality_profile)
                texts.append(t_np)
                r0 += nr
            supp["config2_reads_1e7"] = dict(run_batches(texts, fq.ENCODING_PHRED33), workload="synthetic 150 bp Phred+33, 10 M reads (BASELINE configs[1], '10 M reads' reading)")
            del texts
            # config 5's shape on one GPU
            t5, _ = compress.synth_fastq(2_400_000, min_len=35, max_len=301, n_permille=50, phred=64)
            t5 = t5[:1_000_000_000]
            k5 = bytes(t5[-8192:]).rfind(b"\n@SIM:")
            t5 = t5[: t5.size - 8192 + k5 + 1]
            supp["config5_shape_1gpu"] = dict(run_batches([t5], fq.ENCODING_PHRED64), workload="synthetic 35-301 bp, 5 % N, Phred+64, 1 GB on ONE GPU (BASELINE configs[4] is this shape on 8)")
        except Exception as e:
            supp["error"] = repr(e)
        out["supplementary"] = supp
    if not a.no_cpu and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(batches[0])
        except Exception as e:  # the baseline leg must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
