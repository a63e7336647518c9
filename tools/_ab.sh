B=tools/bin/gemm_bench
for m in 0 8 0 8; do echo "## XCD_MAP=$m"; FCMF_GEMM_XCD_MAP=$m $B 10 0 "fwd  ffn1"; FCMF_GEMM_XCD_MAP=$m $B 10 0 "dX   ffn2"; done
run() { python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"; }
for m in 0 8 0 8; do echo "step XCD_MAP=$m"; FCMF_GEMM_XCD_MAP=$m run; done
FCMF_GEMM_XCD_MAP=8 timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -x -k gemm 2>&1 | tail -2
