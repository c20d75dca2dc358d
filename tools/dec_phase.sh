#!/bin/bash
# standalone kernel times of the decode (FQZ_DBG_SERIAL=1) with k_dec_huf cut short (FQZ_DBG_DEC: 2 = parse + weights, 1 = + tables, 0 = all)
for dbg in 0 2 4 3; do
  FQZ_DBG_SERIAL=1 FQZ_DBG_DEC=$dbg python bench.py --no-supp --no-cpu --no-v3 --inflight 0 --decode-steps 3 --steps 2 --warmup 1 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('dbg=$dbg', 'dec_MBps', d['decode_MBps'], {k:v for k,v in d['decode_kernel_ms'].items() if v>0.03}, d['roundtrip_bit_exact'])
except Exception as e:
    print('dbg=$dbg failed:', t[-3:])"
done
