#!/bin/bash
# rocprofv3 kernel statistics of a non-headline bench workload (run on the GPU box from the repo root):
#   tools/profile_workload.sh iaog r02   -> gpurun_out/r02_iaog_kernel_stats.csv + gpurun_out/r02_bench_iaog.json
W=$1; TAG=${2:-r02}; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$W -o $TAG -- python3 $ROOT/bench.py --workload $W --steps 5 --warmup 2 "$@" > $OUT/${TAG}_${W}_prof_bench.json 2> $OUT/${TAG}_${W}_prof.err
cp $(find $OUT/prof_${TAG}_$W -name "${TAG}_kernel_stats.csv" | head -1) $OUT/${TAG}_${W}_kernel_stats.csv
rm -rf $OUT/prof_${TAG}_$W
cd $ROOT
python3 bench.py --workload $W "$@" > $OUT/${TAG}_bench_$W.json 2> $OUT/${TAG}_bench_$W.err
python3 - $OUT/${TAG}_${W}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 7 steps: %.1f  (per step %.2f ms), launches per step %.0f" % (tot / 1e6, tot / 7e6, sum(int(r["Calls"]) for r in rows) / 7))
for r in rows[:14]:
    print("%-90s %6s calls %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
PY
