#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_asm
for v in A g16; do
  if [ $v = A ]; then unset FQZ_LIB_PATH; else export FQZ_LIB_PATH=$R/ab_build_g16/libfqzhip.so; fi
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_asm/$v -- python3 bench.py --steps 1 --warmup 0 --no-cpu --decode-steps 1 --profile 0 --inflight 0 > gpurun_out/pmc_asm/$v.log 2>&1 || true
done
python3 - <<'PY'
import csv, glob, collections
for v in ("A", "g16"):
    for f in glob.glob("gpurun_out/pmc_asm/%s/**/*counter_collection.csv" % v, recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")[:24]
            agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
        for k, (s, n) in agg.items():
            if "assemble" in k or "k_split" in k: print(v, k, "per launch KiB", s / n)
PY
