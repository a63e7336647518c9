B=tools/bin/gemm_bench
echo "## KB=64 (product)"; $B 10 0 fwd; $B 10 0 dX; $B 10 0 "sq   NT"
echo "## KB=32"; export FCMF_GEMM_KB=32; $B 10 0 fwd; $B 10 0 dX;  $B 10 0 "sq   NT"
