#!/bin/bash
# Further counters per kernel (instruction fetch, memory-instruction levels): one more --pmc pass.
# (A pass with the texture-addresser counters - TA_BUSY_sum, TA_ADDR_STALLED_BY_TC_CYCLES_sum, TCP_PENDING_STALL_CYCLES_sum ... - made
#  rocprofv3 abort and the run hang until gpurun's silence limit killed it: not repeated.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_more
mkdir -p $OUT
cd $R
P3="SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
rocprofv3 --pmc $P3 --output-format csv -d $OUT/p3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-v3 --no-supp --decode-steps 1 --profile 0 --inflight 0 > $OUT/p3.log 2>&1 || true
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_more"
res = collections.defaultdict(dict)
for tag in ("p3",):
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
