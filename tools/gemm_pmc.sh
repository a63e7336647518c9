#!/bin/bash
# LDS / wait counters of the persistent GEMM kernels on the step's shapes (run on the GPU box from the repo root):
#   tools/gemm_pmc.sh  -> gpurun_out/gemm_pmc_<set>.csv   (rocprofv3 --pmc, one pass per counter set, kernel-trace only)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/gpmc_$i -o g -- $ROOT/tools/bin/gemm_bench 2 0 > $OUT/gemm_pmc_$i.log 2>&1
  f=$(find $OUT/gpmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $OUT/gemm_pmc_$i.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    e = acc[k][r["Counter_Name"]]; e[0] += 1; e[1] += float(r["Counter_Value"])
for k, d in acc.items():
    if "gemm" in k or "splitk" in k:
        print(k, {c: round(v[1] / v[0]) for c, v in d.items()}, "launches", max(v[0] for v in d.values()))
PY
  rm -rf $OUT/gpmc_$i
done
cat $OUT/gemm_pmc_*.txt
