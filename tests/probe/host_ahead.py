"""How far the host runs ahead of the GPU in the train step (not a test): wall time of HotPathTrainer.step calls that return without a
device synchronisation against the GPU time of the same steps.   python tests/probe/host_ahead.py [cifar10|celeba]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
import bench                                        # noqa: E402
import v_diffusion                                   # noqa: E402
from v_diffusion.trainer import HotPathTrainer       # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
B = 128
dev = torch.device("cuda", 0)
W = bench.WORKLOADS[wl]
model = bench.build_model(dev, cfg=W["cfg"]).train()
diffusion = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc",
                                          "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.1)
trainer = HotPathTrainer(model, diffusion, lr=2e-4, weight_decay=0.001, warmup=1000, grad_norm=1.0, ema_decay=0.9999, use_ema=True)
x = torch.rand((B, 3, W["res"], W["res"]), device=dev) * 2 - 1
y = (torch.rand((B, 40), device=dev) < 0.2).float() if wl == "celeba" else torch.randint(1, 11, (B,), device=dev).float()
for _ in range(3):
    trainer.step(x, y)
torch.cuda.synchronize()
N = 10
host = []
t0 = time.perf_counter()
for _ in range(N):
    h0 = time.perf_counter()
    trainer.step(x, y)
    host.append((time.perf_counter() - h0) * 1e3)
t_enq = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) * 1e3
print(f"{wl}: {N} steps: host enqueue {t_enq / N:.2f} ms/step (per call: {' '.join(f'{h:.1f}' for h in host)}), with the final synchronize {t_all / N:.2f} ms/step")
# one step alone from an idle GPU: the host's lead is zero at its start
torch.cuda.synchronize()
t0 = time.perf_counter(); trainer.step(x, y); h = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize(); a = (time.perf_counter() - t0) * 1e3
print(f"single step from an idle GPU: host {h:.2f} ms, done after {a:.2f} ms")
