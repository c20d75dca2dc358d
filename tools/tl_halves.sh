#!/bin/bash
# kernel timeline of one encode step with the two halves in flight (FQZ_ENC_HALVES=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tlh
mkdir -p $OUT
cd $R
export FQZ_ENC_HALVES=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu --no-v3 --no-supp --decode-steps 0 --inflight 0 --profile 0 > $OUT/bench.log 2>&1
python3 tools/timeline.py $OUT/trace k_line_local 1 > $OUT/timeline.txt 2>&1
tail -3 $OUT/bench.log | cut -c1-300
