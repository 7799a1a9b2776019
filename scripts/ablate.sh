#!/bin/bash
# GRU timing ablations (results are wrong under ablation; only the class timings matter)
for ab in 0 1 2 4 8 16 3 15 31; do
  for mode in "" "--stepwise"; do
    echo -n "ablate=$ab mode=${mode:-persistent} "
    timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gru-ablate $ab $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['roofline']['classes']
print('ms/step %.2f gru_fwd %.2f gru_bwd %.2f gemm %.2f'%(d['ms_per_step'],c['gru_fwd']['ms_per_step'],c['gru_bwd']['ms_per_step'],c['gemm']['ms_per_step']))"
  done
done
