"""Experiment: throughput of back-to-back batches with 1 context (synchronous) vs 2 contexts in flight."""
import ctypes as C, os, sys, time
sys.path.insert(0, ".")
import torch
import fastqpacker_amd as fq
from fastqpacker_amd import compress
from fastqpacker_amd._lib import lib, BatchResult, check
n_rec = 2849003
text, _ = compress.synth_fastq(n_rec)
text = text[:1000000000]
k = bytes(text[-4096:]).rfind(b"\n@SIM:")
text = text[: text.size - 4096 + k + 1]
dev = torch.device("cuda:0")
d_text = torch.from_numpy(text).to(dev)
NC = int(os.environ.get("NCTX", "2"))
ctxs = [fq.Ctx(0) for _ in range(NC)]
outs = [torch.empty(text.size // 2 + (1 << 20), dtype=torch.uint8, device=dev) for _ in range(NC)]
streams = [torch.cuda.Stream(dev) for _ in range(NC)]
res = [BatchResult() for _ in range(NC)]
inflight = [False] * NC

def launch(i):
    check(lib().fqz_encode_batch_launch(ctxs[i].handle, d_text.data_ptr(), text.size, fq.DEFAULT_BLOCK_SIZE, fq.ENCODING_PHRED33, fq.BATCH_FINAL,
                                        outs[i].data_ptr(), outs[i].numel(), C.c_void_p(streams[i].cuda_stream)))
    inflight[i] = True

def finish(i):
    if inflight[i]:
        check(lib().fqz_encode_batch_finish(ctxs[i].handle, C.byref(res[i]), None, None, 0))
        inflight[i] = False

PROF = int(os.environ.get("PROF", "0"))
for steps in (4, 20):
    for c in ctxs:
        c.profile(bool(PROF))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        i = s % NC
        finish(i)
        launch(i)
    for i in range(NC):
        finish(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("ctxs %d steps %d: %.3f ms/step, %.1f GB/s" % (NC, steps, dt / steps * 1e3, text.size * steps / dt / 1e9), flush=True)
    if PROF:
        k = ctxs[0].profile_read()
        print({n: round(v[0] / v[1], 3) for n, v in k.items() if n in ("k_entropy", "k_split", "k_line_starts", "k_count_nl", "k_compact")}, flush=True)
