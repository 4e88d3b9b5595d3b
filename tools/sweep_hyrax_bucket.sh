#!/bin/bash
# the Hyrax derefs commitment WITHOUT the lookup table (bucket method, shared bucket set per row): window bits / lanes per bucket / segment length
for cfg in "- - -" "10 - -" "11 - -" "12 - -" "13 - -" "14 - -" "- 1 -" "- 2 -" "- - 16" "- - 64"; do set -- $cfg
  for v in SBN_MSM_C SBN_ACC_G SBN_MSM_SEG; do unset $v; done
  [ "$1" != "-" ] && export SBN_MSM_C=$1; [ "$2" != "-" ] && export SBN_ACC_G=$2; [ "$3" != "-" ] && export SBN_MSM_SEG=$3
  python bench.py --workload hyrax --precompute-gb 0 --steps 4 --warmup 1 --blocks none --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_avg_ms']
print('C=$1 G=$2 SEG=$3 ms', d['ms_per_step'], {n:v for n,v in k.items() if v>0.2})"
done
