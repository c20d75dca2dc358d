for b in 1e9 5e8 2.5e8; do FQZ_DBG_SERIAL=1 python bench.py --bytes $b --no-supp --no-cpu --no-v3 --inflight 0 --decode-steps 3 --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['decode_kernel_ms']; print('$b', {x:k[x] for x in ('k_dec_seq_exec','k_dec_seq_fse','k_dec_huf','k_dec_assemble','k_dec_walk_s','k_dec_sizes')})"; done
