#!/bin/bash
# sweep window size for the bench workload (dev tool):  tools/sweep.sh <log_n> "<c list>"
for c in ${2:-14 15 16 17}; do
  echo "== c=$c"; SBN_MSM_C=$c python bench.py --steps 8 --warmup 2 --no-cpu-baseline --log-n ${1:-20} | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(l['value']/1e6,1),'Mpts/s', l['ms_per_step'],'ms', {k:v for k,v in l['kernels_avg_ms'].items() if v>0.05})"
done
