#!/bin/bash
# 2-rank rehearsal of the data-parallel bench on ONE GPU (gloo; RCCL refuses two ranks on one device).  Its timings mean nothing
# (gloo stages through the host); it proves that `bench.py --gpus N` runs end to end with N > 1: arena, in-order buckets, deferred
# weight gradients flushed per bucket, finish(), optimizer step, one JSON line from rank 0, every rank exits 0.
#   tools/dp_rehearsal.sh [workload]  ->  gpurun_out/r03_dp2_gloo_rehearsal_<workload>.json
W=${1:-fcmf}
FCMF_BENCH_SINGLE_DEVICE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29513 bench.py --workload $W --gpus 2 --steps 3 --warmup 1 --backend gloo --no-cpu-baseline \
  > gpurun_out/r03_dp2_gloo_rehearsal_$W.json 2> gpurun_out/r03_dp2_$W.err
echo "rc $?"
