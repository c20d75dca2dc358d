#!/bin/bash
# average duration of one kernel, alone (FQZ_DBG_SERIAL=1), for library variants - from a rocprofv3 kernel trace, so that variants
# that decode garbage on purpose (timing experiments) can be measured too: tools/kstat_variant.sh KERNEL "" ab_build_x/libfqzhip.so ...
K=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  OUT=$R/gpurun_out/kstat_$(basename $(dirname "${lib:-./intree/x}"))
  rm -rf $OUT; mkdir -p $OUT
  cd $R
  if [ -z "$lib" ]; then unset FQZ_LIB_PATH; else export FQZ_LIB_PATH=$R/$lib; fi
  FQZ_DBG_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-v3 --no-supp --decode-steps 3 --inflight 0 --profile 0 > $OUT/bench.log 2>&1
  python3 - "$OUT" "$K" "$lib" <<'PY'
import csv, glob, sys
out, k, lib = sys.argv[1:4]
for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(k): print("[%s] %s calls %s avg %.1f us" % (lib, r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
