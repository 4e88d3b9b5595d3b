#!/bin/bash
# lanes per bucket of k_acc_first (SBN_ACC_G) and segment length (SBN_MSM_SEG) on a 2^22 / 2^21 MSM, one step in flight
for n in ${1:-22}; do for cfg in "- -" "1 -" "2 -" "4 -" "1 64" "2 64" "2 256"; do set -- $cfg
  if [ "$1" = "-" ]; then unset SBN_ACC_G; else export SBN_ACC_G=$1; fi
  if [ "$2" = "-" ]; then unset SBN_MSM_SEG; else export SBN_MSM_SEG=$2; fi
  python bench.py --steps 8 --warmup 2 --blocks none --no-cpu-baseline --log-n $n --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_avg_ms']
print('2^$n G=$1 SEG=$2 ms', d['ms_per_step'], 'acc', k.get('k_acc_first'), 'extra', k.get('k_acc_extra'), 'merge', k.get('k_acc_merge'), 'l1', k.get('k_reduce_l1'))"
done; done
