#!/bin/bash
# rocprofv3 kernel trace of one batched cubic sumcheck (tools/bench_sumcheck.py): per-dispatch durations of the round kernels
export TMPDIR=/tmp
d=gpurun_out/prof_sc; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 tools/bench_sumcheck.py 21 1 > $d/bench.json 2> $d/err.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_sc/trace/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = open('gpurun_out/prof_sc/dispatches.txt', 'w')
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sbn::', '')
    if not n.startswith('k_sc') and not n.startswith('k_bind'): continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    out.write(f"{n:40s} grid={r.get('Grid_Size_X','?'):>8s}x{r.get('Grid_Size_Y','?'):>4s} wg={r.get('Workgroup_Size_X','?')} vgpr={r.get('VGPR_Count','?')} {dur:10.1f} us\n")
out.close()
PY
tail -70 $d/dispatches.txt
