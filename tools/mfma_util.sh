#!/bin/bash
# MFMA utilisation of the step's kernels as rocprofv3 counts it (run on the GPU box from the repo root):
#   TAG=r03 tools/mfma_util.sh -> gpurun_out/r03_mfma_util.txt
# One --pmc pass (kernel-trace only) of bench.py with SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe busy cycles, summed over the
# SIMDs) and GRBM_GUI_ACTIVE (GPU-active cycles at the clock the chip actually held, summed over the 8 XCDs:
# MI355X_MICROARCH.md, DVFS give-back).  Per kernel:
#   MfmaUtil  = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)         -- share of the matrix pipes' cycles that were busy
#   clock     = GRBM_GUI_ACTIVE / 8 / kernel duration                  -- reads high on launches shorter than ~0.3 ms
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfu -o m -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/mfma_util.log 2>&1
python3 - $OUT/mfu > $OUT/${TAG:-r02}_mfma_util.txt <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
acc = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
seen = set()
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"][:64]
    a = acc[k]
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": a[0] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": a[1] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); a[2] += dur.get(r["Dispatch_Id"], 0.0); a[3] += 1
print("# kernel | launches | time ms | MfmaUtil (busy share of the matrix pipes' cycles) | clock GHz (GRBM_GUI_ACTIVE / 8 / time)")
tot_b = tot_g = 0.0
for k, (b, g, t, n) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
    if b <= 0 or g <= 0: continue
    tot_b += b; tot_g += g
    print("%-64s %4d %8.3f   %5.1f %%   %.2f" % (k, n, t * 1e3, 100.0 * b / (g / 8 * 1024), g / 8 / t / 1e9))
print("# all kernels with matrix instructions: MfmaUtil %.1f %%" % (100.0 * tot_b / (tot_g / 8 * 1024)))
PY
rm -rf $OUT/mfu
cat $OUT/r02_mfma_util.txt
