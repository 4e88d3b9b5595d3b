#!/bin/bash
# 2^20 headline (6 in flight) and one MSM alone under the accumulate kernel's block size / segment length and the sort's split / tile overrides
run() { python bench.py --steps 16 --warmup 4 --blocks none --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['serial_reference']; k=s['kernels_avg_ms']
print(os.environ.get('TAG'), 'pipelined', d['ms_per_step'], 'alone', s['ms_per_step'], 'acc', k['k_acc_first'], 'sort', round(sum(v for n,v in k.items() if n.startswith('k_s2')),4))"; }
TAG=default run
for b in 64 128; do TAG="ACC_BLOCK=$b" SBN_ACC_BLOCK=$b run; done
for s in 64 256 512; do TAG="SEG=$s" SBN_MSM_SEG=$s run; done
for l in 7 9 10; do TAG="SORT2_LO=$l" SBN_SORT2_LO=$l run; done
TAG="SORT2_EPT=16" SBN_SORT2_EPT=16 run
TAG=default run
