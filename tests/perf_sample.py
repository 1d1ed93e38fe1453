"""Sampling micro-benchmark (not a test): DDIM + CFG steps of the CIFAR-10 model, for rocprofv3.
    python tests/perf_sample.py [steps] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
import torch
import v_diffusion
from bench import build_model

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
model = build_model(dev).eval()
gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine"), T, "v", "fixed_medium", "snr_trunc", "mse", intp_frac=0.3,
                                   w_guide=1.0)
lab = torch.randint(1, 11, (B,), device=dev).float()
for mode in (False, True, None):
    gd.p_sample(model, (B, 3, 32, 32), label=lab, device=dev, seed=1, use_ddim=True, use_graph=mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gd.p_sample(model, (B, 3, 32, 32), label=lab, device=dev, seed=1, use_ddim=True, use_graph=mode)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B} T={T} use_graph={mode}: {dt:.3f} s  {dt / T * 1e3:.2f} ms/step  {B * T / dt / 50:.2f} img/s at 50 steps "
          f"(graphs cached: {len(getattr(gd, '_graphs', {}))})", flush=True)
