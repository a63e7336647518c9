#!/bin/bash
# kernel launches PER STEP of a bench workload, initialisation excluded: two rocprofv3 kernel traces with different step counts,
# launches/step = (calls(17 steps) - calls(7 steps)) / 10.   tools/launch_count.sh iaog  ->  gpurun_out/launch_count_<workload>.txt
W=${1:-fcmf}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 5 15; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lc_$n -o lc -- python3 $ROOT/bench.py --workload $W --steps $n --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/lc_$n.err
done
python3 - $OUT > $OUT/launch_count_$W.txt <<'PY'
import csv, glob, sys
out = sys.argv[1]
def load(n):
    f = glob.glob(out + "/lc_%d/**/lc_kernel_stats.csv" % n, recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(5), load(15)
tot = sum(b[k][0] - a.get(k, (0, 0))[0] for k in b) / 10.0
ms = sum(b[k][1] - a.get(k, (0, 0.0))[1] for k in b) / 10.0 / 1e6
print("launches per step %.1f   kernel ms per step %.2f   (initialisation: %d launches)" % (tot, ms, sum(v[0] for v in a.values()) - 7 * tot))
rows = sorted(((b[k][0] - a.get(k, (0, 0))[0]) / 10.0, k) for k in b)
for c, k in rows[::-1][:25]:
    print("%7.1f  %s" % (c, k[:110]))
PY
rm -rf $OUT/lc_5 $OUT/lc_15
cat $OUT/launch_count_$W.txt
