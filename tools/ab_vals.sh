#!/bin/bash
# A/B of an environment variable's values on the default bench: tools/ab_vals.sh VAR "v1 v2 ..." [steps]   (three rounds, interleaved)
V=$1; VALS=$2; S=${3:-4}
for i in 1 2 3; do
  for x in $VALS; do
    env $V=$x python bench.py --no-supp --no-cpu --no-v3 --inflight 0 --decode-steps 6 --steps $S --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x', 'enc_ms', d['ms_per_step'], 'dec_MBps', d['decode_MBps'], d['roundtrip_bit_exact'])"
  done
done
