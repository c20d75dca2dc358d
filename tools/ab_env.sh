#!/bin/bash
# A/B of an environment switch on the default bench: tools/ab_env.sh VAR [steps]   (VAR=0 against VAR=1, three runs each, interleaved)
V=$1; S=${2:-10}
for i in 1 2 3; do
  for x in 0 1; do
    env $V=$x python bench.py --no-supp --no-cpu --no-v3 --inflight 0 --decode-steps 1 --steps $S --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x', d['ms_per_step'], d['value'], d['roundtrip_bit_exact'])"
  done
done
