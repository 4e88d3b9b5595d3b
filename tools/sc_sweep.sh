# sweep of the streaming fused sumcheck round: kernel form / occupancy (SBN_SC_WAVES) x blocks per instance (SBN_SC_GRID)
for w in ${WAVES:-2 3}; do for g in ${GRIDS:-28 56 113 227 455}; do
  SBN_SC_WAVES=$w SBN_SC_GRID=$g python tools/bench_sumcheck.py 21 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d['fused']
print('waves',d['env'].get('SBN_SC_WAVES'),'grid',d['env'].get('SBN_SC_GRID'),'fused_kernel_ms',f['kernel_only_rounds_ms'],'GB/s',f['kernel_only_rounds_GBps'],'ms_total',f['ms_per_sumcheck'], 'eval0', f['kernels_ms_total'].get('k_sc_eval_cubic'))"
done; done
