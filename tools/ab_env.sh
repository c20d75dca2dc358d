#!/bin/bash
# A/B of environment switches inside ONE gpurun call: tools/ab_env.sh "" "FQZ_X=1" ...   (prints encode ms and decode MB/s)
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --no-cpu --steps 3 --inflight 0 --decode-steps 6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('[$cfg]', 'enc_ms', d['ms_per_step'], 'dec_MBps', d['decode_MBps'])
"
done
done
