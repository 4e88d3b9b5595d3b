#!/bin/bash
# effective shader clock of the long dispatches: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch time
# (MI355X_MICROARCH.md, DVFS give-back).  PMC pass and kernel-trace pass are separate runs (pool rule).
export TMPDIR=/tmp
d=gpurun_out/prof_clock; rm -rf $d; mkdir -p $d
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d/pmc_sc -- python3 tools/bench_sumcheck.py 21 1 > $d/sc.json 2> $d/err1.log
rocprofv3 --kernel-trace --output-format csv -d $d/trace_sc -- python3 tools/bench_sumcheck.py 21 1 > /dev/null 2> $d/err2.log
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d/pmc_msm -- python3 bench.py --inflight 1 --steps 5 --warmup 1 --no-cpu-baseline --blocks none > $d/msm.json 2> $d/err3.log
rocprofv3 --kernel-trace --output-format csv -d $d/trace_msm -- python3 bench.py --inflight 1 --steps 5 --warmup 1 --no-cpu-baseline --blocks none > /dev/null 2> $d/err4.log
python3 - <<'PY'
import csv, glob, collections
for tag in ("sc", "msm"):
    pm = glob.glob(f'gpurun_out/prof_clock/pmc_{tag}/**/*counter_collection.csv', recursive=True)
    tr = glob.glob(f'gpurun_out/prof_clock/trace_{tag}/**/*kernel_trace.csv', recursive=True)
    if not pm or not tr:
        print(tag, "missing", pm, tr); continue
    act = collections.defaultdict(list)
    for r in csv.DictReader(open(pm[0])):
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            act[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        dur[r['Kernel_Name'].split('(')[0]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k in act:
        if k in dur and max(dur[k]) > 300000:
            a = max(act[k]); t = max(dur[k])
            print(f"{tag:4s} {k[:50]:50s} longest dispatch {t/1e3:9.1f} us  GUI_ACTIVE {a:12.0f}  -> {a/8/t:5.2f} GHz")
PY
