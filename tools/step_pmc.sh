#!/bin/bash
# SQ counters of every kernel of the headline step (run on the GPU box from the repo root):
#   tools/step_pmc.sh -> gpurun_out/step_pmc.txt   (rocprofv3 --pmc passes with kernel-trace only; bench.py --steps 1 --warmup 1)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
: > $OUT/step_pmc.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/spmc_$i -o s -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/step_pmc_$i.log 2>&1
  f=$(find $OUT/spmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" >> $OUT/step_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    e = acc[k][r["Counter_Name"]]; e[0] += 1; e[1] += float(r["Counter_Value"])
keep = ("attn", "add_ln", "colsum", "adamw", "embed", "sum_axis", "box", "gemm_bf16_tile256_kernel<true, true")
for k, d in acc.items():
    if any(x in k for x in keep):
        print(k, {c: round(v[1] / v[0]) for c, v in d.items()}, "launches", max(v[0] for v in d.values()))
PY
  rm -rf $OUT/spmc_$i
done
