"""One-off: records of wildly mixed lengths (20 bp .. 40 kbp, with N, with plus payloads) through Compress / Decompress:
the staged text assembly takes as many records per trip as fit its window and stores oversized ones directly."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa: F401
import fastqpacker_amd as fq
import oracle_lib as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    recs = []
    n = int(rng.integers(50, 3000))
    for i in range(n):
        kind = rng.random()
        L = int(rng.integers(20, 300)) if kind < 0.9 else (int(rng.integers(2000, 9000)) if kind < 0.98 else int(rng.integers(9000, 40000)))
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L, p=[.245, .245, .245, .245, .02]).tobytes()
        qk = rng.random()
        if qk < 0.4:
            q = rng.integers(35, 74, L, dtype=np.uint8).tobytes()
        elif qk < 0.8:
            q = rng.choice(np.array([70, 58, 44, 35], dtype=np.uint8), L, p=[.9, .05, .03, .02]).tobytes()
        else:
            q = bytes([int(rng.integers(35, 74))]) * L
        h = b"r%d/%d len=%d" % (case, i, L)
        plus = h if rng.random() < 0.2 else b""
        recs.append(b"@" + h + b"\n" + seq + b"\n+" + plus + b"\n" + q + b"\n")
    text = b"".join(recs)
    z = fq.compress.Compress(text)
    ok = z == O.compress(text) and fq.compress.Decompress(z) == text and O.decompress(z) == text
    z3 = fq.compress.Compress(text, fq.Options(0, 0, 3))  # container version 3 (rANS-coded qualities)
    ok = ok and z3 == O.compress(text, entropy=2) and fq.compress.Decompress(z3) == text and len(z3) <= len(z) + 64
    print("case %d: %d records, %d bytes -> %d: %s" % (case, n, len(text), len(z), "ok" if ok else "FAIL"), flush=True)
    bad += not ok
sys.exit(1 if bad else 0)
