timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
python bench.py --no-cpu-baseline > gpurun_out/b1.json 2>gpurun_out/b.err && python bench.py --no-cpu-baseline > gpurun_out/b2.json 2>>gpurun_out/b.err
cut -c80-200 gpurun_out/b1.json gpurun_out/b2.json
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/abp -o p -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/abp.log 2>&1
