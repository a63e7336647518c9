"""Diagnostic: where the HOST time of one benchmark step goes (cProfile around the step function, GPU work asynchronous).
python tools/host_profile.py [workload]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.getcwd())
import torch
import bench  # noqa: E402

orig_timed = bench.timed_loop


def timed(step, args, world, dev):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    with torch.autograd.set_multithreading_enabled(False):      # backward in THIS thread, so that cProfile sees it
        pr.enable()
        for _ in range(3):
            step()
        pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("tottime").print_stats(45)
    return orig_timed(step, args, world, dev)


bench.timed_loop = timed
wl = sys.argv[1] if len(sys.argv) > 1 else "iaog"
sys.argv = ["bench.py", "--workload", wl, "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
bench.main()
