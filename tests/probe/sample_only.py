"""DDIM + CFG sampling alone (for rocprofv3 --kernel-trace --stats / --pmc): python tests/probe/sample_only.py [steps] [B] [cifar10|celeba] [passes]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
import bench                                        # noqa: E402
import v_diffusion                                   # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
WL = sys.argv[3] if len(sys.argv) > 3 else "cifar10"
PASSES = int(sys.argv[4]) if len(sys.argv) > 4 else 2
RES = bench.WORKLOADS[WL]["res"]
dev = torch.device("cuda", 0)
model = bench.build_model(dev, cfg=bench.WORKLOADS[WL]["cfg"]).eval()
diffusion = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), T, "v", "fixed_medium", "snr_trunc",
                                          "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.1)
y = (torch.rand((B, 40), device=dev) < 0.2).float() if WL == "celeba" else torch.randint(1, 11, (B,), device=dev).float()
for it in range(PASSES):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = diffusion.p_sample(model, shape=(B, 3, RES, RES), label=y, device=dev, seed=131071, use_ddim=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"pass {it}: {T} steps, {dt / T * 1e3:.2f} ms/step, finite={bool(torch.isfinite(out).all())}")
