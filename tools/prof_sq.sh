#!/bin/bash
# SQ counters of the long dispatches of a command: where the wave cycles go (issue, wait on memory, wait on issue)
# usage: tools/prof_sq.sh <tag> <min_us> -- <program and args>   (the program itself follows --, pool rule)
export TMPDIR=/tmp
tag=$1; minus=$2; shift 3
d=gpurun_out/prof_sq_$tag; rm -rf $d; mkdir -p $d
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $d/pmc -- "$@" > $d/out.json 2> $d/err.log
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $d/pmc2 -- "$@" > /dev/null 2> $d/err2.log
python3 - "$d" "$minus" <<'PY'
import csv, glob, collections, sys
d, minus = sys.argv[1], float(sys.argv[2])
rows = collections.OrderedDict()
for sub in ("pmc", "pmc2"):
    fs = glob.glob(f'{d}/{sub}/**/*counter_collection.csv', recursive=True)
    if not fs: print("no counters in", sub); continue
    for r in csv.DictReader(open(fs[0])):
        key = (r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sbn::', ''), r['Dispatch_Id'])
        rows.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
# the two passes are separate processes: match dispatches by (kernel, ordinal)
byk = collections.defaultdict(list)
for (k, did), v in rows.items(): byk[k].append((int(did), v))
out = open(f'{d}/summary.txt', 'w')
for k, lst in byk.items():
    lst.sort()
    a = [v for _, v in lst if 'SQ_WAVE_CYCLES' in v]; b = [v for _, v in lst if 'GRBM_GUI_ACTIVE' in v]
    for i, v in enumerate(a):
        w = v.get('SQ_WAVE_CYCLES', 0)
        if w < minus: continue
        g = b[i].get('GRBM_GUI_ACTIVE', 0) if i < len(b) else 0
        line = (f"{k[:36]:36s} #{i:3d} wave_cyc {w:13.0f} busy {v.get('SQ_BUSY_CYCLES',0):11.0f} wait_any {v.get('SQ_WAIT_ANY',0)/w:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/w:5.2f} "
                f"active_any {v.get('SQ_ACTIVE_INST_ANY',0)/w:5.2f} active_valu {v.get('SQ_ACTIVE_INST_VALU',0)/w:5.2f} insts_valu {v.get('SQ_INSTS_VALU',0):13.0f} waves {v.get('SQ_WAVES',0):8.0f} gui_active {g:11.0f}")
        print(line); out.write(line + "\n")
PY
