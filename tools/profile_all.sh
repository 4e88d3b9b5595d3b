#!/bin/bash
# The rocprofv3 command set behind profiles/ (run on the GPU box from the repo root; raw output under gpurun_out/).
# Kernel-trace and PMC passes are separate runs, and the PMC passes never carry a trace flag (pool rule).
set -e
export TMPDIR=/tmp
for wl in msm hyrax; do
  tag=$([ $wl = msm ] && echo msm20 || echo hyrax)
  d=gpurun_out/prof_$tag; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 bench.py --workload $wl --no-cpu-baseline > $d/bench_default.json 2> $d/err1.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace_serial -- python3 bench.py --workload $wl --inflight 1 --no-cpu-baseline > $d/bench_serial.json 2> $d/err2.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/pmc_fetch -- python3 bench.py --workload $wl --inflight 1 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $d/err3.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/pmc_write -- python3 bench.py --workload $wl --inflight 1 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $d/err4.log
  python3 tools/summarize_prof.py $d r01_$tag $wl
  mkdir -p gpurun_out/profiles_new && cp profiles/r01_${tag}* profiles/pmc_traffic.json gpurun_out/profiles_new/
  echo "== $wl done"
done
