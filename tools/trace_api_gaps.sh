# HIP API calls next to the kernel trace of the compiled prove harness: what the host does in the gaps between kernels
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_api
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT -- python3 tools/trace_harness.py 2 > $OUT/out.json 2> $OUT/err.log
ls -R $OUT | head -20
