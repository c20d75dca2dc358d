"""Timing experiment: k_dec_huf variants (FQZ_DBG_DEC); outputs are garbage for non-zero modes."""
import os, sys, ctypes as C
sys.path.insert(0, ".")
import torch
import fastqpacker_amd as fq
from fastqpacker_amd import compress
from fastqpacker_amd._lib import lib, BatchResult
text, n = compress.synth_fastq(2850000)
dev = torch.device("cuda:0")
t = torch.from_numpy(text).to(dev)
out = torch.empty(text.size, dtype=torch.uint8, device=dev)
ctx = fq.Ctx(0)
res = compress.encode_batch_dev(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), ctx=ctx)
z = out[: res.out_len].clone()
torch.cuda.synchronize()  # (torch's copy before the library's stream reads it)
back = torch.empty(text.size + 4096, dtype=torch.uint8, device=dev)
for mode in [0, 2, 1, 0]:
    os.environ["FQZ_DBG_DEC"] = str(mode)
    dres = BatchResult()
    ctx.profile(True)
    for _ in range(3):
        rc = lib().fqz_decode_batch_dev(ctx.handle, z.data_ptr(), z.numel(), 2, 0, back.data_ptr(), back.numel(), C.byref(dres), None)
    k = ctx.profile_read()
    ctx.profile(False)
    print("mode", mode, "rc", rc, {n: round(v[0] / v[1], 3) for n, v in k.items() if n in ("k_dec_huf", "k_dec_entropy")}, flush=True)
