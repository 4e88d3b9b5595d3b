# where the single-launch sumcheck round (ticket + mailbox) should hand over to the streaming kernel + finish launch
for m in ${MAXES:-16384 8192 4096 2048 1024}; do
  SBN_SC_SINGLE_MAX=$m python tools/bench_sumcheck.py 21 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d['fused']
print('single_max',d['env'].get('SBN_SC_SINGLE_MAX'),'fused_kernel_ms',f['kernel_only_rounds_ms'],'GB/s',f['kernel_only_rounds_GBps'],'ms_total',f['ms_per_sumcheck'], f['kernels_ms_total'], f['kernels_launches'])"
done
