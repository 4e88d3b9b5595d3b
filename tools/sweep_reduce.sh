# level 1 of the bucket reduction of one MSM alone at 2^20: SBN_RED_L = buckets per lane (any value; default: chosen so the chunks spread evenly over
# the SIMDs), SBN_RED_QUAD = 0 / 1 (combine level one wave per group / quad-cooperative: the default in the latency regime).
# First the MSM parity tests under the odd values (ragged last chunk), then the timings.
for L in 3 5 7; do
  SBN_RED_L=$L timeout -k 10 300 python -m pytest tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -1 || exit 1
done
for cfg in "1 -" "1 3" "1 4" "1 5" "1 6" "1 8" "0 5"; do set -- $cfg
  if [ "$2" = "-" ]; then unset SBN_RED_L; else export SBN_RED_L=$2; fi
  SBN_RED_QUAD=$1 python bench.py --steps 12 --warmup 3 --blocks none --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['serial_reference']['kernels_avg_ms']
print('quad=$1 L=$2 ms_per_step', d['ms_per_step'], 'l1', k['k_reduce_l1'], 'combine', k['k_reduce_combine'], 'acc', k['k_acc_first'])"
done
