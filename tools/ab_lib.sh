#!/bin/bash
# A/B of library builds inside ONE gpurun call: tools/ab_lib.sh "" ab_build_x/libfqzhip.so ...  ("" = the in-tree build)
for rep in 1 2 3; do
for lib in "$@"; do
  if [ -z "$lib" ]; then unset FQZ_LIB_PATH; else export FQZ_LIB_PATH=$PWD/$lib; fi
  python bench.py --no-cpu --no-v3 --no-supp --steps 6 --warmup 3 --inflight 0 --decode-steps 6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms']; dk = d['decode_kernel_ms']; print('[$lib]', 'enc_ms', d['ms_per_step'], 'k_entropy', k.get('k_entropy'), 'k_split', k.get('k_split'), 'dec_MBps', d['decode_MBps'], 'asm', dk.get('k_dec_assemble'), 'huf', dk.get('k_dec_huf'), 'ok', d['roundtrip_bit_exact'])
"
done
done
