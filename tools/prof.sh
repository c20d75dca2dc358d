#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats, then HBM byte counters in separate passes.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-v3 --no-supp --decode-steps 1 --inflight 0 > $OUT/bench_trace.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu --decode-steps 0 --profile 0 --inflight 0 --no-v3 --no-supp > $OUT/bench_fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu --decode-steps 0 --profile 0 --inflight 0 --no-v3 --no-supp > $OUT/bench_write.log 2>&1 || true
# the same batch as a version-3 container (FQZ-R1): kernel trace only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_v3 -- python3 bench.py --steps 5 --warmup 1 --no-cpu --no-supp --container 3 --decode-steps 1 --inflight 0 > $OUT/bench_trace_v3.log 2>&1 || true
find $OUT -name "*.csv" | head -20
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:24]:
        print("%-32s calls %6s total_ns %12s avg_ns %10s pct %6s" % (r.get("Name", "")[:32], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
for tag in ("fetch", "write"):
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")[:32]
            agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
        print("== pmc", tag, f)
        for k, (v, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:14]:
            print("%-32s launches %5d sum %14.1f per_launch %12.1f" % (k, n, v, v / n))
PY
