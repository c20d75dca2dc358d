"""Diagnostic: per-phase cycles of k_seg_encode from in-kernel s_memtime stamps (FQZ_DBG_STAMPS=1)."""
import ctypes as C, os, sys
import numpy as np
os.environ["FQZ_DBG_STAMPS"] = "1"
os.environ["FQZ_ENC_SEG"] = "1"
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
    res = compress.encode_batch_dev(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), ctx=ctx, qual_encoding=fq.ENCODING_PHRED33)
nseg = 2 * (text.size // 65536 + 4096)
buf = np.zeros((nseg, 16), dtype=np.uint64)
got = C.c_size_t(0)
check(lib().fqz_debug_get_stamps(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), nseg, C.byref(got)))
allb = buf[: got.value].astype(np.int64)
half = got.value // 2
buf, ent = allb[:half], allb[half:]
ent = ent[buf[:, 13] > 0]
buf = buf[buf[:, 13] > 0]
labels = ["bitmap", "lines", "records", "piece tables", "split pieces", "hdr/plus/len", "npos", "places+copyout", "qual", "plus", "npos-ent", "len", "end"]
prev = buf[:, 0]
tot = (buf[:, 13] - buf[:, 0]).mean()
print("segments %d  total %.0f cycles (100 MHz ticks x?)" % (len(buf), tot))
for k, lab in enumerate(labels, start=1):
    cur = buf[:, k]
    ok = cur > 0
    d = np.where(ok, cur - prev, 0)
    print("  %-16s %8.0f" % (lab, d[ok].mean() if ok.any() else 0))
    prev = np.where(ok, cur, prev)
span = buf[:, 13].max() - buf[:, 0].min()
print("kernel span %d ticks; sum of segment times / span = %.1f concurrent" % (span, (buf[:, 13] - buf[:, 0]).sum() / span))

# the quality coder's own stamps (fqz_entropy_dev.h DBG_STOP indices)
order = [1, 2, 3, 4, 10, 11, 12, 13, 14, 5, 6, 9]
labels = ["load+hist", "classify", "ranksort", "huff+depth", "nbits", "weights", "fse:tables", "fse:simulate", "fse:resolve+replay", "tree(FSE/direct)", "codes+clear", "both chunks coded"]
prev = buf[:, 8]
print("quality coder:")
for k, lab in zip(order, labels):
    cur = ent[:, k]
    ok = cur > 0
    d = np.where(ok, cur - prev, 0)
    print("  %-20s %8.0f" % (lab, d[ok].mean() if ok.any() else 0))
    prev = np.where(ok, cur, prev)
