# large single MSMs, one step in flight: window bits around the automatic choice (round 3: after the single-reduction Y3)
# usage (GPU box): bash tools/sweep_msm_large.sh > gpurun_out/sweep_msm_large.log
run() { python bench.py --log-n $LOGN --inflight 1 --steps $STEPS --warmup 1 --no-cpu-baseline --blocks none 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); k=d['serial_reference']['kernels_avg_ms']; a=d['roofline'].get('alu',{})
sort=sum(v for n,v in k.items() if n.startswith(('k_s2','k_digits','k_hist','k_scatter','k_block_prefix','k_scan')))
print(os.environ.get('TAG'), 'n=2^'+os.environ['LOGN'], 'c', a.get('window_bits'), 'W', a.get('windows'), 'ms/step', d['serial_reference']['ms_per_step'], 'sort %.3f' % sort, 'acc', k.get('k_acc_first'), 'extra', k.get('k_acc_extra'), 'merge', k.get('k_acc_merge'), 'l1', k.get('k_reduce_l1'), 'comb', k.get('k_reduce_combine'))"; }
export LOGN STEPS
LOGN=24 STEPS=4; TAG=auto run; for C in 18 19 21 22; do TAG="c=$C" SBN_MSM_C=$C run; done
LOGN=26 STEPS=3; TAG=auto run; for C in 19 21 22; do TAG="c=$C" SBN_MSM_C=$C run; done
LOGN=22 STEPS=6; TAG=auto run; for C in 16 18 19; do TAG="c=$C" SBN_MSM_C=$C run; done
