"""Deterministic FASTQ generators for tests (python side; small sizes only)."""
import numpy as np


def illumina_header(i):
    return b"SIM:1:FCX123:1:%d:%d:%d 1:N:0:ATCACG" % (1101 + i // 200000, 1000 + (7919 * i) % 20000, 1000 + (104729 * i) % 20000)


def make_fastq(n_records, seed=1, min_len=150, max_len=150, n_frac=0.0, phred=33, plus_payload=False,
               qual_levels=None, crlf=False):
    rng = np.random.default_rng(seed)
    out = []
    levels = qual_levels or [37, 25, 11, 2]
    eol = b"\r\n" if crlf else b"\n"
    for i in range(n_records):
        L = int(rng.integers(min_len, max_len + 1))
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L)
        if n_frac > 0:
            seq = np.where(rng.random(L) < n_frac, ord("N"), seq).astype(np.uint8)
        q = np.empty(L, dtype=np.uint8)
        cur = 0
        r = rng.random(L)
        for j in range(L):
            if r[j] > 0.9:
                cur = int(rng.integers(0, len(levels)))
            q[j] = levels[cur] + phred
        hdr = illumina_header(i) + (b" length=%d" % L if min_len != max_len else b"")
        plus = hdr if (plus_payload and i % 3 == 0) else b""
        out.append(b"@" + hdr + eol + seq.tobytes() + eol + b"+" + plus + eol + q.tobytes() + eol)
    return b"".join(out)
