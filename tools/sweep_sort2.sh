# large single MSMs (one step in flight): the two-level sort against the one-level sort, and the window width it unlocks
# usage (GPU box): bash tools/sweep_sort2.sh [quick] > gpurun_out/sweep_sort2.log
run() { python bench.py --log-n $LOGN --inflight 1 --steps $STEPS --warmup 1 --no-cpu-baseline --blocks none 2>gpurun_out/sweep_sort2.err | python -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); k=d['kernels_avg_ms']
a=d['roofline']['alu']
sort=sum(v for n,v in k.items() if n.startswith(('k_s2','k_digits','k_hist','k_scatter','k_block_prefix','k_scan')))
print(os.environ.get('TAG'), 'n=2^'+os.environ['LOGN'], 'c', a['window_bits'], 'W', a['windows'], 'ms/step', d['ms_per_step'], 'pts/s %.3e' % d['value'], 'sort %.3f' % sort, 'acc', k.get('k_acc_first'), 'extra', k.get('k_acc_extra'), 'l1', k.get('k_reduce_l1'), 'comb', k.get('k_reduce_combine'), {n: v for n, v in k.items() if n.startswith(('k_s2', 'k_block_prefix', 'k_scan'))})"; }
export LOGN STEPS
if [ "$1" = quick ]; then
LOGN=26 STEPS=3; TAG="auto" run; TAG="ept16" SBN_SORT2_EPT=16 run; for LO in 9 10; do TAG="lo=$LO" SBN_SORT2_LO=$LO run; done
LOGN=24 STEPS=5; TAG="auto" run; TAG="lo=10" SBN_SORT2_LO=10 run; TAG="c=19" SBN_MSM_C=19 run
LOGN=23 STEPS=5; TAG="auto" run; TAG="c=18" SBN_MSM_C=18 run; TAG="c=20" SBN_MSM_C=20 run
LOGN=22 STEPS=8; TAG="auto" run; TAG="lo=9" SBN_SORT2_LO=9 run; TAG="c=18" SBN_MSM_C=18 run
LOGN=21 STEPS=8; TAG="auto" run; TAG="c=16" SBN_MSM_C=16 run
LOGN=20 STEPS=10; TAG="one-level" SBN_SORT2_MIN=0 run; TAG="two-level (default from 2^20)" run; TAG="two-level c=16" SBN_SORT2_MIN=1024 SBN_MSM_C=16 run; TAG="two-level c=17" SBN_SORT2_MIN=1024 SBN_MSM_C=17 run
LOGN=19 STEPS=10; TAG="one-level (default)" run; TAG="two-level" SBN_SORT2_MIN=1024 run
LOGN=18 STEPS=10; TAG="one-level (default)" run; TAG="two-level" SBN_SORT2_MIN=1024 run
exit 0
fi
LOGN=22 STEPS=8; TAG="one-level" SBN_SORT2_MIN=0 run; TAG="two-level auto" run; for C in 16 17 18; do TAG="two-level c=$C" SBN_MSM_C=$C run; done
LOGN=24 STEPS=5; TAG="one-level" SBN_SORT2_MIN=0 run; TAG="two-level auto" run; for C in 16 18 19 20; do TAG="two-level c=$C" SBN_MSM_C=$C run; done
LOGN=26 STEPS=3; TAG="one-level" SBN_SORT2_MIN=0 run; TAG="two-level auto" run; for C in 16 19 20 21 22; do TAG="two-level c=$C" SBN_MSM_C=$C run; done
