#!/bin/bash
# One GPU-box call: the -m gpu test suite, then (unless the tests were killed / timed out) the default bench line.
# usage: tools/gpu_ci.sh <tag> [pytest args...]
tag=${1:-ci}; shift
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/${tag}_pytest.log
tail -5 gpurun_out/${tag}_pytest.log
if [ $rc -le 1 ]; then
  timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
  echo "bench rc=$?"
  tail -c 3000 gpurun_out/${tag}_bench.json
  tail -5 gpurun_out/${tag}_bench.err
fi
exit $rc
