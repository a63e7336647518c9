"""Calibration only (never on the product path): the step's big GEMM shapes through torch.matmul (hipBLASLt / rocBLAS)
and through fcmf_gemm (weight gradients: accumulate = 1, as the step issues them), same operands, HIP-event timed.  Tells how far the hand-written kernels are from the vendor's
best on THESE shapes.  Run on the GPU box: python tools/blaslt_calib.py > gpurun_out/blaslt_calib.txt"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multimodal-aspect-category-sentiment-analysis_amd"))
ops = importlib.import_module("fcmf_framework.ops")

dev = torch.device("cuda:0")
T = 49152
SHAPES = [  # name, M, N, K, ta, tb   (C[M,N] = op(A) op(B); ta: A stored [K,M]; tb = 0: B stored [N,K], 1: [K,N])
    ("fwd  ffn1  Y=X W^T", T, 3072, 768, 0, 0),
    ("fwd  ffn2  Y=H W^T", T, 768, 3072, 0, 0),
    ("fwd  proj  Y=X W^T", T, 768, 768, 0, 0),
    ("fwd  qkv   Y=X W^T", T, 2304, 768, 0, 0),
    ("dX   ffn1  dX=dY Wt^T", T, 768, 3072, 0, 0),
    ("dX   ffn2  dH=dY Wt^T", T, 3072, 768, 0, 0),
    ("dW   ffn1  dW=dY^T X", 3072, 768, T, 1, 1),
    ("dW   ffn2  dW=dY^T H", 768, 3072, T, 1, 1),
    ("dW   proj  dW=dY^T X", 768, 768, T, 1, 1),
    ("dW   qkv   dW=dY^T X", 2304, 768, T, 1, 1),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, M, N, K, ta, tb in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=dev, dtype=torch.bfloat16)
    C = torch.empty((M, N), device=dev, dtype=torch.float32 if ta else torch.bfloat16)
    Cb = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    opA = A.t() if ta else A
    opB = B if tb else B.t()
    t_lt = timeit(lambda: torch.matmul(opA, opB, out=Cb))
    # weight gradients as the step issues them: accumulate into an f32 gradient (split-K through the registered workspace)
    t_my = timeit(lambda: ops.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), C.stride(0), bool(ta), bool(tb), acc=bool(ta)))
    if ta:
        C.zero_()
        ops.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), C.stride(0), True, bool(tb), acc=True)
    ref = torch.matmul(opA.float()[:256], opB.float())
    err = (C[:256].float() - ref).abs().max().item() / ref.abs().max().item()
    fl = 2.0 * M * N * K
    print("%-24s M=%6d N=%5d K=%6d  torch.matmul %.3f ms %6.0f TF | fcmf_gemm %.3f ms %6.0f TF | ratio %.2f  relerr %.1e" %
          (name, M, N, K, t_lt, fl / t_lt / 1e9, t_my, fl / t_my / 1e9, t_lt / t_my, err), flush=True)
