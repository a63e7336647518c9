timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py tests/test_abi.py -q -x > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
python bench.py --no-cpu-baseline > gpurun_out/b1.json 2>gpurun_out/b.err && python bench.py --no-cpu-baseline > gpurun_out/b2.json 2>>gpurun_out/b.err
cut -c80-200 gpurun_out/b1.json gpurun_out/b2.json
