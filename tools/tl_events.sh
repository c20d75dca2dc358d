#!/bin/bash
# the library's own timeline (HIP events, no profiler): last encode and last decode of a short bench run   [FQZ_LIB_PATH: a variant build]
FQZ_DBG_TIMELINE=1 python bench.py --steps 3 --warmup 2 --no-cpu --no-v3 --no-supp --decode-steps 2 --inflight 0 2> gpurun_out/tl_events.err > gpurun_out/tl_events.json
python - <<'PY'
blocks=[]; cur=None
for ln in open("gpurun_out/tl_events.err"):
    if "[fqz timeline]" not in ln: continue
    if "launches" in ln:
        cur=[]; blocks.append(cur)
    else: cur.append(ln.rstrip().replace("[fqz timeline] ",""))
enc=[b for b in blocks if any("k_split" in l for l in b)]
dec=[b for b in blocks if any("k_dec_assemble" in l for l in b)]
for name,b in (("encode",enc),("decode",dec)):
    if b:
        print("== last", name); print("\n".join(b[-1]))
PY
