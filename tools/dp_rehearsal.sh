#!/bin/bash
# 2-rank rehearsal of the data-parallel bench on ONE GPU (gloo; RCCL refuses two ranks on one device).  Its timings mean nothing
# (gloo stages through the host); it proves that the BARE command `python bench.py --gpus 2` (no launcher: bench.py starts its own
# ranks) runs end to end: arena, in-order buckets in launch groups, deferred weight gradients flushed per group, finish(), optimizer
# step, one JSON line from rank 0, exit code 0.
#   tools/dp_rehearsal.sh [workload] [tag]  ->  gpurun_out/<tag>_dp2_gloo_rehearsal_<workload>.json
W=${1:-fcmf}; TAG=${2:-r04}
FCMF_BENCH_SINGLE_DEVICE=1 timeout -k 10 500 python bench.py --workload $W --gpus 2 --steps 3 --warmup 1 --backend gloo --no-cpu-baseline \
  > gpurun_out/${TAG}_dp2_gloo_rehearsal_$W.json 2> gpurun_out/${TAG}_dp2_$W.err
echo "rc $?"
