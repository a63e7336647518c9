B=tools/bin/gemm_bench
echo "## product"; $B 10 0 fwd; $B 10 0 dW
echo "## prio_b"; LD_LIBRARY_PATH=tools/bin/prio_b $B 10 0 fwd; LD_LIBRARY_PATH=tools/bin/prio_b $B 10 0 dW
echo "## product"; $B 10 0 fwd; $B 10 0 dW
