#!/bin/bash
# same-box A/B of two builds of libfcmf_hip.so on the step's forward / dX GEMM shapes: tools/bin/prev vs the in-tree library
B=tools/bin/gemm_bench
for round in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export LD_LIBRARY_PATH=$(pwd)/tools/bin/prev; else unset LD_LIBRARY_PATH; fi
    for s in "fwd  ffn1" "fwd  ffn2" "fwd  qkv" "dX   ffn2" "dX   ffn1" "dX   qkv" "dX   out"; do
      $B 20 0 "$s" | grep -v "^----" | sed "s/^/$which $round  /"
    done
  done
done
unset LD_LIBRARY_PATH
echo "== nt off (plain stores), new build"
for s in "fwd  ffn1" "dX   ffn2" "fwd  qkv"; do FCMF_GEMM_NT_MIN_MB=100000 $B 20 0 "$s" | grep -v "^----" | sed "s/^/plain-stores  /"; done
