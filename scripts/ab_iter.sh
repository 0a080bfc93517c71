#!/bin/bash
# A/B of one environment switch on ONE box: bench lines with the switch off / on / off / on (R1 iteration, 10 steps each)
#   bash scripts/ab_iter.sh LCGAN_GZ_DEMOD [extra bench args]
v=$1; shift
for r in 1 2; do
  for on in 0 1; do
    env $v=$on timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cycle "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms',{})
print('$v=$on', round(d['ms_per_step'],2), 'ms', {n: k.get(n) for n in ('conv_igemm','conv_wgrad','act_bwd','rgb')})" || exit 1
  done
done
