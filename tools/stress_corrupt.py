"""One-off: random single-byte corruptions of a small .fqz must end in an error or in the original text - never in
garbage, a hang or a fault.  usage: python tools/stress_corrupt.py [flips] [seed] [container version 2|3]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa: F401
import fastqpacker_amd as fq
from fastq_gen import make_fastq

flips = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
text = make_fastq(1500, seed=77, min_len=30, max_len=80, n_frac=0.01)
version = int(sys.argv[3]) if len(sys.argv) > 3 else 2
good = fq.compress.Compress(text, fq.Options(0, 0, version))
assert fq.compress.Decompress(good) == text
n_err = n_same = n_garbage = 0
for k in range(flips):
    bad = bytearray(good)
    at = int(rng.integers(10, len(bad)))
    bad[at] ^= 1 << int(rng.integers(0, 8))
    try:
        out = fq.compress.Decompress(bytes(bad))
    except Exception:
        n_err += 1
        continue
    if out == text:
        n_same += 1
    else:
        n_garbage += 1
        print("GARBAGE at byte %d of %d" % (at, len(bad)), flush=True)
    if k % 50 == 0:
        print("flip %d: errors %d, unchanged %d, garbage %d" % (k, n_err, n_same, n_garbage), flush=True)
print("done: %d flips: %d errors, %d decoded to the original text, %d garbage" % (flips, n_err, n_same, n_garbage))
sys.exit(1 if n_garbage else 0)
