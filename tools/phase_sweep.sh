#!/bin/bash
# timing experiment: k_entropy phase costs by early exit (outputs are garbage for FQZ_DBG_STOP != 0)
for s in 0 1 2 3 4 5 6 7 8; do
  FQZ_DBG_STOP=$s python bench.py --steps 3 --warmup 1 --no-cpu --decode-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('stop=$s', 'k_entropy_ms', d['kernel_ms'].get('k_entropy'), 'step_ms', d['ms_per_step'])
"
done
