cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04c_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r04c_pytest.log
[ $rc -eq 0 ] || exit $rc
python tools/bench_bullet.py > gpurun_out/r04c_bullet.log 2>&1; tail -5 gpurun_out/r04c_bullet.log
HARNESS_MARKERS=0 python tools/trace_harness.py 3 > gpurun_out/r04c_harness.json 2> gpurun_out/r04c_harness.err; cat gpurun_out/r04c_harness.json
rm -rf gpurun_out/prof_r04_prove; mkdir -p gpurun_out/prof_r04_prove
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_prove/trace -- python3 tools/trace_harness.py 2 > gpurun_out/prof_r04_prove/out.json 2> gpurun_out/prof_r04_prove/err.log
python3 tools/trace_summary.py gpurun_out/prof_r04_prove/trace stages > gpurun_out/r04_prove_trace_summary.txt; python3 tools/trace_summary.py gpurun_out/prof_r04_prove/trace pass > gpurun_out/r04_prove_trace_pass.txt; tail -2 gpurun_out/r04_prove_trace_summary.txt gpurun_out/r04_prove_trace_pass.txt
