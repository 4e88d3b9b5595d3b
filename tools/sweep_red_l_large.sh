#!/bin/bash
# buckets per lane of k_reduce_l1 (SBN_RED_L, any value) on large MSMs: the throughput regime of the reduction (millions of buckets)
for n in ${1:-24}; do for L in ${2:-- 8 12 16 20 24 32 48}; do
  if [ "$L" = "-" ]; then unset SBN_RED_L; else export SBN_RED_L=$L; fi
  python bench.py --steps 4 --warmup 1 --blocks none --no-cpu-baseline --log-n $n --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_avg_ms']
print('2^$n L=$L ms', d['ms_per_step'], 'l1', k.get('k_reduce_l1'), 'comb', k.get('k_reduce_combine'))"
done; done
