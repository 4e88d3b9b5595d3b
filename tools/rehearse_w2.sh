#!/bin/bash
# bench.py --gpus 2 as the driver launches it, but with two ranks sharing the one GPU of the box and the 64-byte partials over gloo
# (BENCH_BACKEND=gloo): the N > 1 code path of bench.py (sharded Hyrax rows, strong-scaling 2^26 MSM, parity gates, rank-0 cpu_baseline).
# usage: tools/rehearse_w2.sh <tag>
tag=${1:-w2}; mkdir -p gpurun_out
BENCH_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 5 --warmup 2 --precompute-gb 90 > gpurun_out/${tag}_bench_w2.json 2> gpurun_out/${tag}_bench_w2.err
rc=$?; echo "rc=$rc"; tail -c 2500 gpurun_out/${tag}_bench_w2.json; tail -5 gpurun_out/${tag}_bench_w2.err
exit $rc
