"""PCIe-inclusive rate of the host-buffer entry points (fqz_compress / fqz_decompress on memory buffers)."""
import sys, time
sys.path.insert(0, ".")
import torch
import fastqpacker_amd as fq
from fastqpacker_amd import compress
text, _ = compress.synth_fastq(2849003)
text = text[:1000000000]
k = bytes(text[-4096:]).rfind(b"\n@SIM:")
text = text[: text.size - 4096 + k + 1]
z = compress.Compress(text)              # warm-up (allocations, pinned staging)
import ctypes as C
import numpy as np
from fastqpacker_amd._lib import lib, check, default_ctx
ctx = default_ctx()
zin = np.frombuffer(z, dtype=np.uint8)
cbuf = np.empty(lib().fqz_encode_bound(text.size) + 10, dtype=np.uint8)
dbuf = np.empty(text.size + 4096, dtype=np.uint8)
cbuf[:] = 0; dbuf[:] = 0                      # touch the pages: the timing below is the library, not the page faults
n = C.c_size_t(0)
def c_compress():
    check(lib().fqz_compress(ctx.handle, text.ctypes.data, text.size, cbuf.ctypes.data, cbuf.size, C.byref(n), None))
def c_decompress():
    check(lib().fqz_decompress(ctx.handle, zin.ctypes.data, zin.size, dbuf.ctypes.data, dbuf.size, C.byref(n), None))
for name, fn in (("fqz_compress", c_compress), ("fqz_decompress", c_decompress)):
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    print("%s: %.1f ms, %.2f GB/s of FASTQ (pageable host buffers in and out)" % (name, best * 1e3, text.size / best / 1e9), flush=True)
assert bytes(dbuf[: n.value]) == bytes(text)
assert compress.Decompress(z) == bytes(text)
print("ratio %.3f" % (text.size / len(z)))
