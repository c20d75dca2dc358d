"""Diagnostic: where a tile of k_fsplit spends its time (in-kernel s_memtime stamps, FQZ_DBG_FS_STAMPS=1)."""
import ctypes as C, os, sys
import numpy as np
os.environ["FQZ_DBG_FS_STAMPS"] = "1"
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
    res = compress.encode_batch_dev(t.data_ptr(), t.numel(), out.data_ptr(), out.numel(), qual_encoding=0, ctx=ctx)
nt = text.size // 4096 + 8
buf = np.zeros((nt, 8), dtype=np.uint64)
got = C.c_size_t(0)
check(lib().fqz_debug_get_fs_stamps(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), nt, C.byref(got)))
b = buf[: got.value].astype(np.int64)
b = b[b[:, 6] > 0]
labels = ["A:load+list", "B:lookback1+fwd", "C:sizes", "lookback2", "D:offsets", "split", ]
print("tiles", len(b), "s_memtime cycles: total per tile %.1f" % (b[:, 6] - b[:, 0]).mean())
for k, lab in enumerate(labels):
    d = b[:, k + 1] - b[:, k]
    print("  %-16s mean %8.1f  p50 %8.1f  p90 %8.1f" % (lab, d.mean(), np.median(d), np.percentile(d, 90)))
span = b[:, 6].max() - b[:, 0].min()
print("kernel span %.1f ticks; sum of tile times / span = %.1f tiles in flight" % (span, (b[:, 6] - b[:, 0]).sum() / span))
