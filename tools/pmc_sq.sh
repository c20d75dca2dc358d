#!/bin/bash
# SQ counters (issue / wait / LDS) per kernel: separate rocprofv3 --pmc passes, program directly after "--"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq
mkdir -p $OUT
cd $R
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_INST_LDS"
rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-v3 --no-supp --decode-steps 1 --profile 0 --inflight 0 > $OUT/p1.log 2>&1 || true
rocprofv3 --pmc $P2 --output-format csv -d $OUT/p2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-v3 --no-supp --decode-steps 1 --profile 0 --inflight 0 > $OUT/p2.log 2>&1 || true
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_sq"
res = collections.defaultdict(dict)
for tag in ("p1", "p2"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r.get("Kernel_Name", "")[:40], r.get("Counter_Name"))
            agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
        for (k, c), (v, n) in agg.items():
            res[k][c] = v / n
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(res, key=lambda k: -res[k].get("SQ_WAVE_CYCLES", 0))[:16]:
        line = "%-40s " % k + " ".join("%s=%.3g" % (c, v) for c, v in sorted(res[k].items()))
        print(line); fo.write(line + "\n")
PY
