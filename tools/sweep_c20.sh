#!/bin/bash
# round 4: window bits around the automatic choice after the reduction got cheaper (any L, quad combine): pipelined (6 in flight) and one MSM alone
# usage: tools/sweep_c20.sh "<log_n list>" "<c list>"
for n in ${1:-20}; do for c in ${2:-14 15 16 17} auto; do
  if [ $c = auto ]; then unset SBN_MSM_C; else export SBN_MSM_C=$c; fi
  python bench.py --steps 10 --warmup 3 --blocks none --no-cpu-baseline --log-n $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['serial_reference']; k=s['kernels_avg_ms']
print('2^$n c=$c pipelined ms', d['ms_per_step'], 'Mpts/s', round(d['value']/1e6,1), '| alone ms', s['ms_per_step'], 'acc', k.get('k_acc_first'), 'extra', k.get('k_acc_extra'), 'l1', k.get('k_reduce_l1'), 'comb', k.get('k_reduce_combine'), 'sort', round(sum(v for n,v in k.items() if n.startswith('k_s2') or n.startswith('k_hist') or n.startswith('k_scatter') or n.startswith('k_size') or n.startswith('k_digits')),4), 'W', d['roofline']['alu'].get('windows'))"
done; done
