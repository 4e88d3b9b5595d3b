#!/bin/bash
# The collective code path of bench.py on a ONE-GPU box: world size 1, but dist.init_process_group("nccl", device_id=dev)
# is called and every partial sum / row commitment goes through sharding.allgather_fold / gather_rows / allgather_bytes as device uint8
# tensors over RCCL.  Output: gpurun_out/<tag>_rccl_w1.json (+ .err with RCCL's own version / transport lines).
tag=${1:-r04}
mkdir -p gpurun_out
export NCCL_DEBUG=VERSION
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --force-collective --blocks hyrax,sweep > gpurun_out/${tag}_rccl_w1.json 2> gpurun_out/${tag}_rccl_w1.err
rc=$?
echo "rccl world-1 bench rc=$rc"
tail -c 1500 gpurun_out/${tag}_rccl_w1.json
tail -5 gpurun_out/${tag}_rccl_w1.err
exit $rc
