#!/bin/bash
# standalone decode kernel times (FQZ_DBG_SERIAL=1) for library variants: tools/dec_alone.sh "" ab_build_x/libfqzhip.so ...
for lib in "$@"; do
  if [ -z "$lib" ]; then unset FQZ_LIB_PATH; else export FQZ_LIB_PATH=$PWD/$lib; fi
  FQZ_DBG_SERIAL=1 python bench.py --no-supp --no-cpu --no-v3 --inflight 0 --decode-steps 3 --steps 2 --warmup 1 2>&1 | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('[$lib]', {k:v for k,v in d['decode_kernel_ms'].items() if v>0.03}, d['roundtrip_bit_exact'])
except Exception as e:
    print('[$lib] failed:', t[-2:])"
done
