#!/bin/bash
# End-of-round profile of the headline bench (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh r02
# writes gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), gpurun_out/<tag>_pmc_traffic.json
# (two separate --pmc passes, FETCH_SIZE doubled: MI355X_MICROARCH.md) and gpurun_out/<tag>_bench.json.
# Copy what should be judged into profiles/ afterwards.
set -e
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
cp $(find $OUT/prof_$TAG -name "${TAG}_kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcF_$TAG -o $TAG -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmcF.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmcW_$TAG -o $TAG -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmcW.err
python3 $ROOT/tools/pmc_traffic.py $(find $OUT/pmcF_$TAG -name "*counter_collection.csv" | head -1) $(find $OUT/pmcW_$TAG -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_traffic.json
rm -rf $OUT/prof_$TAG $OUT/pmcF_$TAG $OUT/pmcW_$TAG
cd $ROOT
python3 bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo done
