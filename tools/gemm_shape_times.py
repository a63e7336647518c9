"""Diagnostic: every distinct (kernel, M, N, K, epilogue) GEMM of one benchmark step with its launches and summed HIP-event time
(the weight-gradient GEMMs that go through ops.deferred_dw are launched by its flush, not here: see bench.py's per_kernel line).
python tools/gemm_shape_times.py [bench args]  ->  stderr"""
import importlib
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
import bench  # noqa: E402  (sets up sys.path for the package)

ops = importlib.import_module("fcmf_framework.ops")
H = importlib.import_module("fcmf_framework._hip")
orig = ops.gemm
rec = []
on = [False]


def gemm(A, B, C, M, N, K, lda, ldb, ldc, ta, tb, bias=None, aux=None, epi=0, acc=False, colsum=None):
    if not on[0]:
        return orig(A, B, C, M, N, K, lda, ldb, ldc, ta, tb, bias=bias, aux=aux, epi=epi, acc=acc, colsum=colsum)
    nq = len(ops.deferred_dw.q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(A, B, C, M, N, K, lda, ldb, ldc, ta, tb, bias=bias, aux=aux, epi=epi, acc=acc, colsum=colsum)
    e1.record()
    if len(ops.deferred_dw.q) == nq:            # (launched, not queued)
        rec.append(((H.last_gemm_kernel(), M, N, K, int(ta), int(tb), int(epi), int(acc), colsum is not None), e0, e1))


ops.gemm = gemm
orig_timed = bench.timed_loop


def timed(step, args, world, dev, trace=True):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    on[0] = True
    step()
    torch.cuda.synchronize()
    on[0] = False
    agg = {}
    for key, e0, e1 in rec:
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    tot = sum(a[1] for a in agg.values())
    print(f"# {len(rec)} launched GEMMs, {tot:.2f} ms by HIP events (epi: 0 none 1 gelu 2 tanh 3 gelu' 4 tanh' 5 add)", file=sys.stderr)
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fl = 2.0 * k[1] * k[2] * k[3] * n
        print("%-46s M=%6d N=%5d K=%6d ta=%d tb=%d epi=%d acc=%d cs=%d  x%3d %8.3f ms %7.1f us/call %7.1f TFLOP/s"
              % (*k, n, ms, ms / n * 1e3, fl / (ms * 1e-3) / 1e12), file=sys.stderr)
    return orig_timed(step, args, world, dev, trace)


bench.timed_loop = timed
sys.argv = ["bench.py"] + sys.argv[1:] + ["--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
bench.main()
