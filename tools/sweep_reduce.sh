# reduction variants of one MSM alone at 2^20: SBN_RED_QUAD = 0 (one wave per chunk everywhere), 1 (quad-cooperative combine level: the default in the latency regime), 2 (quads at level 1 too); SBN_RED_L = buckets per lane
for cfg in "0 4" "1 4" "2 4" "2 8" "2 16" "0 8"; do set -- $cfg
  SBN_RED_QUAD=$1 SBN_RED_L=$2 python bench.py --steps 12 --warmup 3 --blocks none --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['serial_reference']['kernels_avg_ms']
print('quad=$1 L=$2 ms_per_step', d['ms_per_step'], 'l1', k['k_reduce_l1'], 'combine', k['k_reduce_combine'], 'acc', k['k_acc_first'])"
done
