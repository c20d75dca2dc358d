#!/bin/bash
# A/B timing in ONE gpurun call (boxes differ by ~10 %): A = the in-tree library, B, C ... = other builds of libfqzhip.so.
# usage: tools/ab.sh "ab_build_b/libfqzhip.so ab_build_c/libfqzhip.so" [bench.py args]
LIBS=$1; shift
for rep in 1 2; do
  for v in A $LIBS; do
    if [ $v = A ]; then unset FQZ_LIB_PATH; else export FQZ_LIB_PATH=$PWD/$v; fi
    python bench.py --no-cpu --steps 5 --inflight 0 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms']; dk = d['decode_kernel_ms']
        print('$v', 'enc_ms', d['ms_per_step'], 'dec_MBps', d['decode_MBps'], 'ok', d['roundtrip_bit_exact'], {n: k[n] for n in ('k_entropy', 'k_split', 'k_line_local', 'k_compact') if n in k}, {n: dk[n] for n in ('k_dec_assemble', 'k_dec_huf', 'k_dec_frames') if n in dk})
"
  done
done
