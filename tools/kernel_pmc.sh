#!/bin/bash
# SQ counters of the kernels a small script launches (GPU box, repo root):  tools/kernel_pmc.sh <tag> <kernel-substring> python3 tools/attn_bench.py
# One rocprofv3 --pmc pass per counter group (kernel-trace only), summed per kernel name -> gpurun_out/<tag>_pmc.txt
TAG=$1; PAT=$2; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  (cd $ROOT && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/kpmc_$i -o k -- "$@" > $OUT/${TAG}_pmc_pass$i.log 2>&1) || echo "pass $i failed: $grp"
done
python3 - $OUT "$PAT" > $OUT/${TAG}_pmc.txt <<'PY'
import csv, sys, glob, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for cc in glob.glob(out + "/kpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"]
        if pat not in k: continue
        acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k[:60], r["Counter_Name"])].add(r["Dispatch_Id"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s %16.0f   per launch %14.0f" % (c, v, v / max(1, len(n[(k, c)]))))
PY
rm -rf $OUT/kpmc_*
cat $OUT/${TAG}_pmc.txt
