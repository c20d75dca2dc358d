#!/bin/bash
# kernel timeline of the last decode of a bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tld
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-v3 --no-supp --decode-steps 2 --inflight 0 --profile 0 > $OUT/bench.log 2>&1
python3 tools/timeline.py $OUT/trace k_dec_blocks 1 > $OUT/timeline.txt 2>&1
head -60 $OUT/timeline.txt
