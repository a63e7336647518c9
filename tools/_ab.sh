timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py -q -x -k "adam or optim or parity or golden or step or resume" > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/abp -o p -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/abp.log 2>&1
grep -E "adamw|sumsq|cast_transpose" $ROOT/gpurun_out/abp/p_kernel_stats.csv | cut -c1-140
