#!/bin/bash
# The rocprofv3 command set behind profiles/ (run on the GPU box from the repo root; raw output under gpurun_out/).
# Kernel-trace and PMC passes are separate runs, the PMC passes never carry a trace flag, and the profiled program follows `--`
# directly (pool rules).  usage: tools/profile_all.sh [tags...]   (default: all)
export TMPDIR=/tmp
R=${R:-r03}
prof() {   # tag workload size pmc? -- program args...
  local tag=$1 wl=$2 size=$3 pmc=$4; shift 5
  local d=gpurun_out/prof_${R}_$tag; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- "$@" > $d/out.json 2> $d/err1.log
  if [ "$pmc" = 1 ]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/pmc_fetch -- "$@" > /dev/null 2> $d/err3.log
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/pmc_write -- "$@" > /dev/null 2> $d/err4.log
  fi
  python3 tools/summarize_prof.py $d ${R}_$tag $wl $size
  echo "== $tag done"
}
want() { [ $# -eq 0 ] && return 0; }
TAGS="$*"
has() { [ -z "$TAGS" ] || [[ " $TAGS " == *" $1 "* ]]; }
B="python3 bench.py --no-cpu-baseline --blocks none"
has msm20 && prof msm20 msm 2^20 0 -- $B --steps 40 --warmup 4
has msm20_serial && prof msm20_serial msm 2^20 1 -- $B --steps 20 --warmup 2 --inflight 1
has msm22 && prof msm22 msm 2^22 1 -- $B --log-n 22 --steps 6 --warmup 1 --inflight 1
has msm24 && prof msm24 msm 2^24 1 -- $B --log-n 24 --steps 4 --warmup 1 --inflight 1
has msm26 && prof msm26 msm 2^26 1 -- $B --log-n 26 --steps 2 --warmup 1 --inflight 1
has hyrax_lookup && prof hyrax_lookup hyrax-lookup 4096x8192 1 -- $B --workload hyrax --steps 8 --warmup 2
has hyrax_bucket && prof hyrax_bucket hyrax-bucket 4096x8192 1 -- $B --workload hyrax --precompute-gb 0 --steps 8 --warmup 2
has sumcheck && prof sumcheck sumcheck 18x2^21 1 -- python3 tools/bench_sumcheck.py 21 2
has bullet && prof bullet bullet 8192 1 -- python3 tools/bench_bullet.py
# round 3: the driver's own command (every block of the bench line) and the compiled prove-stages harness (kernel stats only)
has bench && prof bench bench default 0 -- python3 bench.py
has prove && prof prove prove keyless 0 -- python3 tools/trace_harness.py 2
