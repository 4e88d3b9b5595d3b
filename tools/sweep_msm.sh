# one-step-in-flight MSM 2^20: reduction chunking (SBN_RED_L) and window bits (SBN_MSM_C)
run() { python bench.py --inflight 1 --steps 10 --warmup 2 --no-cpu-baseline --blocks none 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); k=d['kernels_avg_ms']
print(os.environ.get('TAG'), 'ms/step', d['ms_per_step'], 'acc', k.get('k_acc_first'), 'l1', k.get('k_reduce_l1'), 'comb', k.get('k_reduce_combine'), 'scatter', k.get('k_scatter_lds'))"; }
for L in 1 2 4 8; do TAG="L=$L" SBN_RED_L=$L run; done
for C in 14 15 16; do TAG="c=$C" SBN_MSM_C=$C run; done
TAG="c=16 L=2" SBN_MSM_C=16 SBN_RED_L=2 run
TAG="c=15 L=2 G=1" SBN_MSM_C=15 SBN_RED_L=2 SBN_ACC_G=1 run
TAG="c=15 G=4" SBN_ACC_G=4 run
