"""FQZ-S1 smoke: GPU .fqz == oracle .fqz for a handful of shapes; both decoders read it.  Run on the GPU box."""
import os, sys, time
os.environ["FQZ_ENC_SEG"] = "1"  # the experimental segment path for every entry point
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import fastqpacker_amd as fq
from fastqpacker_amd import compress
import oracle_lib as O
from fastq_gen import make_fastq

def first_diff(a, b):
    n = min(len(a), len(b))
    x = np.frombuffer(a[:n], dtype=np.uint8) != np.frombuffer(b[:n], dtype=np.uint8)
    i = int(np.argmax(x)) if x.any() else n
    return i

cases = [("tiny", make_fastq(3, seed=1)), ("3000", make_fastq(3000, seed=1)), ("ragged+N phred64", make_fastq(2000, seed=2, min_len=35, max_len=301, n_frac=0.05, phred=64)),
         ("long reads", make_fastq(50, seed=3, min_len=5000, max_len=30000)), ("very long (fallback)", make_fastq(8, seed=3, min_len=60000, max_len=70000)),
         ("short reads", make_fastq(5000, seed=5, min_len=1, max_len=40)), ("empty", b""), ("20000", make_fastq(20000, seed=44))]
bad = 0
for name, t in cases:
    want = O.compress(t, framing=1)
    try:
        got = compress.Compress(t)
    except Exception as e:
        print("%-22s ENCODE FAILED %r" % (name, e)); bad += 1; continue
    ok = got == want
    line = "%-22s %8d -> %7d (oracle %7d) %s" % (name, len(t), len(got), len(want), "same" if ok else "DIFF at %d" % first_diff(got, want))
    if not ok:
        bad += 1
        h1 = [int.from_bytes(got[10 + 4 * i:14 + 4 * i], "little") for i in range(9)] if len(got) >= 46 else []
        h2 = [int.from_bytes(want[10 + 4 * i:14 + 4 * i], "little") for i in range(9)] if len(want) >= 46 else []
        line += "\n    gpu hdr %s\n    ora hdr %s" % (h1, h2)
    try:
        line += "  oracle-dec %s" % (O.decompress(got) == t)
    except Exception as e:
        line += "  oracle-dec ERR %r" % e
    try:
        line += "  gpu-dec %s" % (compress.Decompress(got) == t)
    except Exception as e:
        line += "  gpu-dec ERR %r" % e
    print(line, flush=True)
print("bad", bad)
