"""BASELINE config 5 shape on one GPU: 35-301 bp, 5 % N, Phred+64 — encode / decode rates and kernel times."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import torch
import fastqpacker_amd as fq
from fastqpacker_amd import compress
from fastqpacker_amd._lib import lib, BatchResult, check
text, _ = compress.synth_fastq(2400000, min_len=35, max_len=301, n_permille=50, phred=64)
text = text[:1000000000]
k = bytes(text[-8192:]).rfind(b"\n@SIM:")
text = text[: text.size - 8192 + k + 1]
dev = torch.device("cuda:0")
d_text = torch.from_numpy(text).to(dev)
d_out = torch.empty(text.size, dtype=torch.uint8, device=dev)
ctx = fq.Ctx(0)
res = BatchResult()
def enc():
    check(lib().fqz_encode_batch_dev(ctx.handle, d_text.data_ptr(), text.size, fq.DEFAULT_BLOCK_SIZE, fq.ENCODING_PHRED64, fq.BATCH_FINAL,
                                     d_out.data_ptr(), d_out.numel(), C.byref(res), None, None, 0, None))
enc()
ctx.profile(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): enc()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
kern = ctx.profile_read(); ctx.profile(False)
print("encode %.3f ms  %.1f GB/s  ratio %.3f  records %d" % (dt * 1e3, text.size / dt / 1e9, text.size / res.out_len, res.n_records))
print({n: round(v[0] / v[1], 3) for n, v in kern.items()})
z = d_out[: res.out_len].clone()
torch.cuda.synchronize()  # (torch's copy before the library's stream reads it)
back = torch.empty(text.size + 4096, dtype=torch.uint8, device=dev)
dres = BatchResult()
def dec():
    check(lib().fqz_decode_batch_dev(ctx.handle, z.data_ptr(), z.numel(), 2, fq.ENCODING_PHRED64, back.data_ptr(), back.numel(), C.byref(dres), None))
dec()
ctx.profile(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): dec()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
kern = ctx.profile_read(); ctx.profile(False)
print("decode %.3f ms  %.1f GB/s  bit-exact %s" % (dt * 1e3, text.size / dt / 1e9, bool(dres.out_len == text.size and torch.equal(back[: text.size], d_text))))
print({n: round(v[0] / v[1], 3) for n, v in kern.items()})
