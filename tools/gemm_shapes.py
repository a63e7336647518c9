"""Prints every distinct GEMM of one benchmark step with the kernel that served it (diagnostic)."""
import importlib
import os
import sys

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402  (sets up sys.path for the package)

ops = importlib.import_module("fcmf_framework.ops")
H = importlib.import_module("fcmf_framework._hip")
orig = ops.gemm
seen = {}


def gemm(A, B, C, M, N, K, lda, ldb, ldc, ta, tb, bias=None, aux=None, epi=0, acc=False, colsum=None):
    orig(A, B, C, M, N, K, lda, ldb, ldc, ta, tb, bias=bias, aux=aux, epi=epi, acc=acc, colsum=colsum)
    key = (H.last_gemm_kernel(), M, N, K, int(ta), int(tb), int(acc))
    seen[key] = seen.get(key, 0) + 1


ops.gemm = gemm
for m in list(sys.modules.values()):
    if getattr(m, "__name__", "").startswith("fcmf_framework") and getattr(m, "ops", None) is ops:
        pass
sys.argv = ["bench.py"] + sys.argv[1:] + ["--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
bench.main()
for k, n in sorted(seen.items(), key=lambda kv: (kv[0][0], -kv[0][1] * kv[0][2] * kv[0][3])):
    print("GEMM %-44s M=%6d N=%5d K=%6d ta=%d tb=%d acc=%d  x%d" % (*k, n), file=sys.stderr)
