"""Diagnostic: every ATen operator one benchmark step dispatches OUTSIDE the C-ABI library (the "torch glue"), with the
innermost frame of this package that asked for it.  Usage (GPU box):  python tools/op_census.py [bench.py args]
A TorchDispatchMode sees forward and backward (the mode travels to the autograd thread with the thread-local state)."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.getcwd())
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench  # noqa: E402

SKIP = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::t", "aten::transpose", "aten::permute", "aten::select",
        "aten::slice", "aten::detach", "aten::alias", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::as_strided",
        "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::unbind", "aten::split", "aten::narrow", "aten::chunk",
        "aten::new_empty", "aten::_local_scalar_dense", "aten::is_pinned", "aten::lift_fresh", "aten::unflatten", "aten::flatten")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if name not in SKIP:
            site = ""
            for fr in reversed(traceback.extract_stack(limit=24)):
                f = fr.filename
                if ("fcmf_framework" in f or f.endswith("bench.py") or "_amd/" in f) and "op_census" not in f:
                    site = "%s:%d %s" % (os.path.basename(f), fr.lineno, fr.name)
                    break
            shp = ""
            for a in args:
                if torch.is_tensor(a):
                    shp = "%s %s" % (tuple(a.shape), str(a.dtype).replace("torch.", ""))
                    break
            self.cnt[(name, site, shp)] += 1
        return func(*args, **(kwargs or {}))


orig_timed = bench.timed_loop


def timed(step, args, world, dev):
    step(); step()
    torch.cuda.synchronize()
    with Census() as c:
        step()
        torch.cuda.synchronize()
    tot = sum(c.cnt.values())
    print("# %d dispatched operators in one step (views / empties not counted)" % tot, file=sys.stderr)
    for (name, site, shp), v in c.cnt.most_common(120):
        print("%4d  %-28s %-44s %s" % (v, name, site, shp), file=sys.stderr)
    return orig_timed(step, args, world, dev)


bench.timed_loop = timed
sys.argv = ["bench.py"] + (sys.argv[1:] or []) + ["--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
bench.main()
