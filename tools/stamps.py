"""Diagnostic: per-phase cycle counts of k_entropy from in-kernel s_memtime stamps (FQZ_DBG_STAMPS=1)."""
import ctypes as C, os, sys
import numpy as np
os.environ["FQZ_DBG_STAMPS"] = "1"
sys.path.insert(0, ".")
import torch
import fastqpacker_amd as fq
from fastqpacker_amd import compress
from fastqpacker_amd._lib import lib, check
text, n = compress.synth_fastq(int(os.environ.get("FQZ_STAMP_RECORDS", "2849002")))
dev = torch.device("cuda:0")
t = torch.from_numpy(text).to(dev)
out = torch.empty(text.size, dtype=torch.uint8, device=dev)
ctx = fq.Ctx(0)
for _ in range(2):
    res = compress.encode_batch_dev(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), ctx=ctx)
nch = res.n_chunks
buf = np.zeros((nch, 16), dtype=np.uint64)
got = C.c_size_t(0)
check(lib().fqz_debug_get_stamps(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), nch, C.byref(got)))
names = ["seq", "qual", "hdr", "plus", "npos", "len"]
order = [0, 1, 2, 3, 4, 10, 11, 12, 13, 14, 5, 6, 7, 8, 9]
labels = ["start", "load+hist", "classify", "ranksort", "huff+depth", "nbits", "weights", "fse:tables", "fse:simulate", "fse:resolve+replay", "tree(FSE/direct)", "codes+clear", "pass1", "hdr+pass2", "copyout"]
for s in range(6):
    sel = buf[buf[:, 15] == s]
    if not len(sel):
        continue
    print("stream %-5s chunks %6d" % (names[s], len(sel)), end="  ")
    full = sel[sel[:, 9] > 0]
    src = full if len(full) else sel
    prev = src[:, 0].astype(np.int64)
    parts = []
    for k, lab in zip(order[1:], labels[1:]):
        cur = src[:, k].astype(np.int64)
        ok = cur > 0
        if ok.sum() == 0:
            continue
        d = np.where(ok, cur - prev, 0)
        parts.append("%s=%.0f" % (lab, d[ok].mean()))
        prev = np.where(ok, cur, prev)
    tot = (src[:, [1,2,3,4,5,6,7,8,9,10,11,12,13,14]].max(axis=1).astype(np.int64) - src[:, 0].astype(np.int64)).mean()
    print("total=%.0f cyc | " % tot + " ".join(parts))
