"""One-off stress of the headers model: random header shapes, GPU bytes == oracle bytes, GPU decode (fast and general path)
== text.  usage: python tools/stress_hdr.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa: F401
import fastqpacker_amd as fq
import oracle_lib as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ALPH = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789:_-/ .", dtype=np.uint8)


def rand_bytes(n, k=None):
    a = ALPH if k is None else ALPH[:k]
    return rng.choice(a, int(n)).tobytes()


def gen_headers(n):
    style = int(rng.integers(0, 8))
    out = []
    if style == 0:      # common prefix + counter
        p = rand_bytes(rng.integers(0, 60))
        out = [p + b"%d" % (i * int(rng.integers(1, 5))) for i in range(n)]
    elif style == 1:    # counter + common tail
        t = rand_bytes(rng.integers(0, 80))
        out = [b"%d" % i + t for i in range(n)]
    elif style == 2:    # prefix + random middle of random length + tail
        p, t = rand_bytes(rng.integers(0, 30)), rand_bytes(rng.integers(0, 30))
        out = [p + rand_bytes(rng.integers(0, 40), 4) + t for _ in range(n)]
    elif style == 3:    # runs of identical headers
        cur = rand_bytes(rng.integers(1, 100))
        for _ in range(n):
            if rng.random() < 0.1:
                cur = rand_bytes(rng.integers(0, 100))
            out.append(cur)
    elif style == 4:    # tiny headers
        out = [rand_bytes(rng.integers(0, 6), 3) for _ in range(n)]
    elif style == 5:    # long headers with shared blocks
        blocks = [rand_bytes(rng.integers(10, 400)) for _ in range(4)]
        out = [b"".join(blocks[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 6)))) + b"%d" % i for i in range(n)]
    elif style == 6:    # one-symbol runs
        out = [bytes([65 + int(rng.integers(0, 2))]) * int(rng.integers(0, 300)) for _ in range(n)]
    else:               # illumina-like with varying field widths
        out = [b"M%d:%d:FC%d:%d:%d:%d:%d %d:N:0:%s" % (rng.integers(0, 3), rng.integers(0, 500), rng.integers(0, 2), rng.integers(1, 9), 1101 + i // 5000,
                                                   rng.integers(0, 30000), rng.integers(0, 30000), rng.integers(1, 3), rand_bytes(8, 4)) for i in range(n)]
    return style, out


bad = 0
for case in range(cases):
    n = int(rng.choice([1, 2, 3, 50, 400, 3000, 12000]))
    style, hs = gen_headers(n)
    L = int(rng.integers(1, 40))
    recs = []
    for h in hs:
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L).tobytes()
        recs.append(b"@" + h.replace(b"\n", b"x") + b"\n" + seq + b"\n+\n" + b"I" * L + b"\n")
    text = b"".join(recs)
    want = O.compress(text)
    got = fq.compress.Compress(text)
    ok = got == want
    dec = fq.compress.Decompress(got) == text if ok else False
    os.environ["FQZ_DEC_GENERAL"] = "1"
    try:
        gen = fq.compress.Decompress(got) == text if ok else False
    finally:
        del os.environ["FQZ_DEC_GENERAL"]
    if not (ok and dec and gen):
        bad += 1
        print("FAIL case %d style %d n %d L %d: enc==oracle %s, decode %s, general %s" % (case, style, n, L, ok, dec, gen), flush=True)
        open("gpurun_out/stress_fail_%d.fq" % case, "wb").write(text)
    elif case % 20 == 0:
        print("case %d ok (style %d, n %d, %d bytes -> %d)" % (case, style, n, len(text), len(got)), flush=True)
print("done: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
