"""Test infrastructure (not collected by pytest): times the CPU oracle at several thread counts on the GPU box's host
to pick the thread count of bench.py's cpu_baseline leg (16 was fastest: the box exposes 256 logical CPUs but a
16-core share).  Usage: python tests/oracle_thread_scan.py"""
import sys, time, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-aspect-category-sentiment-analysis_amd'))
import synthetic_data as synth
from oracle import fcmf_oracle as O
cfg, NI, NR = synth.BASE_CFG, 7, 36
P = {k: v.requires_grad_(True) for k, v in synth.synth_params(synth.fcmf_param_shapes(cfg)).items()}
def step(bsz, seed):
    batch = synth.synth_batch(bsz, cfg, S=128, num_imgs=NI, num_roi=NR, seed=seed)
    for p in P.values(): p.grad = None
    loss, _ = O.fcmf_step_loss(P, cfg, batch, NI, NR, training=True)
    loss.backward()
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    step(1, 1)
    t0 = time.perf_counter(); step(2, 2); dt = time.perf_counter() - t0
    print(f"threads {th}: B=2 fwd+bwd {dt:.1f} s -> {2/dt:.3f} samples/s", flush=True)
