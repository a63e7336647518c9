#!/bin/bash
# What the 64-deep k-tile variant of the persistent GEMM buys and why (run on the GPU box from the repo root, after
# `make -C <pkg>/csrc timing ablate ablate2` and the dmaonly_line build): -> gpurun_out/r02_gemm_k64.txt
B=tools/bin/gemm_bench
OUT=gpurun_out/r02_gemm_k64.txt
{
echo "# Persistent GEMM, K-contiguous operands (NT): 64-deep k-tiles (whole 128-B lines per LDS-DMA piece, five-slot operand ring)"
echo "# vs 32-deep k-tiles (64-B row pieces, 4-stage ring).  tools/bin/gemm_bench 10 0 <prefix>; same box, same process order."
echo "## product build, KB = 64 where K % 64 == 0"
$B 10 0 fwd; $B 10 0 dX; $B 10 0 "sq   NT"; $B 10 0 dW
echo "## product build, FCMF_GEMM_KB=32"
FCMF_GEMM_KB=32 $B 10 0 fwd; FCMF_GEMM_KB=32 $B 10 0 dX; FCMF_GEMM_KB=32 $B 10 0 "sq   NT"
echo "## DMA stream alone (no fragment reads, no MFMA), KB = 32: 256 rows x 64 B per operand k-tile"
FCMF_GEMM_KB=32 LD_LIBRARY_PATH=tools/bin/dmaonly $B 10 0 "sq   NT"; FCMF_GEMM_KB=32 LD_LIBRARY_PATH=tools/bin/dmaonly $B 10 0 "fwd  ffn2"
echo "## DMA stream alone, KB = 32 ring but pieces of 8 rows x 128 B (FCMF_GEMM_ABLATE_LINE: same bytes, every line fetched once by one instruction)"
FCMF_GEMM_KB=32 LD_LIBRARY_PATH=tools/bin/dmaonly_line $B 10 0 "sq   NT"; FCMF_GEMM_KB=32 LD_LIBRARY_PATH=tools/bin/dmaonly_line $B 10 0 "fwd  ffn2"
echo "## DMA stream alone, KB = 64 kernels"
LD_LIBRARY_PATH=tools/bin/dmaonly $B 10 0 "sq   NT"; LD_LIBRARY_PATH=tools/bin/dmaonly $B 10 0 "fwd  ffn2"; LD_LIBRARY_PATH=tools/bin/dmaonly $B 10 0 "sq   TN"
echo "## fragment reads + MFMA, no global->LDS traffic (FCMF_GEMM_ABLATE_DMA), KB = 64 / KB = 32 / transposed operands"
LD_LIBRARY_PATH=tools/bin/nodma $B 10 0 "sq   NT"; FCMF_GEMM_KB=32 LD_LIBRARY_PATH=tools/bin/nodma $B 10 0 "sq   NT 4096"; LD_LIBRARY_PATH=tools/bin/nodma $B 10 0 "sq   TN"
echo "## feed only: DMA + fragment reads, no MFMA (FCMF_GEMM_ABLATE_MMA)"
LD_LIBRARY_PATH=tools/bin/nomma $B 10 0 "sq   NT 4096"; LD_LIBRARY_PATH=tools/bin/nomma $B 10 0 "sq   TN 4096"
echo "## in-kernel phase stamps and the clock the main loop ran at (diagnostic build, FCMF_GEMM_TIMING)"
LD_LIBRARY_PATH=tools/bin/timing $B 10 0 "sq   NT 4096"; LD_LIBRARY_PATH=tools/bin/timing $B 10 0 "sq   TN 4096"; LD_LIBRARY_PATH=tools/bin/timing $B 10 0 "fwd  ffn2"; LD_LIBRARY_PATH=tools/bin/timing $B 10 0 "fwd  ffn1+gelu"; LD_LIBRARY_PATH=tools/bin/timing $B 10 0 "dW   3072"
} > $OUT 2>&1
echo done
