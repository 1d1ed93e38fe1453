"""The single-GPU slice of BASELINE configs[4], timed once (not part of the default bench: ~2 min): CelebA 64x64 (merged config), EMA
weights (`with trainer.ema_weights()`, reference train_utils.py:171-185), DDIM-250, guidance w = 3, sample batch 256 = 512 UNet rows per
reverse step (reference generate.py:138-150, diffusion.py:394-414).  Prints one JSON line: images/s, seconds per batch, algorithmic and
executed TFLOP/s (executed share taken from the launches two recorded reverse steps make).
    python tests/probe/celeba_ddim250.py [--steps 250] [--batch 256] > profiles/r04_celeba_ddim250.json"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
import bench                                        # noqa: E402
import v_diffusion                                   # noqa: E402
from v_diffusion import _hip                         # noqa: E402
from v_diffusion.trainer import HotPathTrainer       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=250)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--w", type=float, default=3.0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
W = bench.WORKLOADS["celeba"]
model = bench.build_model(dev, cfg=W["cfg"]).eval()
gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), a.steps, "v", "fixed_medium", "snr_trunc", "mse",
                                   intp_frac=0.3, w_guide=a.w, p_uncond=0.1)
tr = HotPathTrainer(model, gd, use_ema=True)
g = torch.Generator(dev).manual_seed(4321)
labels = (torch.rand((a.batch, 40), device=dev, generator=g) < 0.2).float()
with tr.ema_weights():
    gd.p_sample(model, (8, 3, 64, 64), label=labels[:8], device=dev, seed=131071, use_ddim=True)             # warm-up (short chain is fine: same kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = gd.p_sample(model, (a.batch, 3, 64, 64), label=labels, device=dev, seed=131071, use_ddim=True)
    torch.cuda.synchronize()
    ds = time.perf_counter() - t0
    xt = torch.randn((a.batch, 3, 64, 64), device=dev)
    _hip.PROFILE = []
    with torch.inference_mode():
        for st in (a.steps - 1, a.steps // 2):
            xt = gd.p_sample_step(model, xt, torch.full((a.batch,), st, device=dev), labels.clone(), use_ddim=True)
    torch.cuda.synchronize()
    rec, _hip.PROFILE = _hip.PROFILE, None
fl = sum(r[1] for r in rec if not r[0].startswith("hbm:"))
fe = sum(r[1] * bench.executed_share(r[0]) for r in rec if not r[0].startswith("hbm:"))
alg = a.steps * 2 * W["fwd_gflop"] * a.batch / ds / 1e3
print(json.dumps({"metric": f"ddim{a.steps}_cfg_samples_per_sec", "value": round(a.batch / ds, 3), "unit": "images/s", "seconds_per_batch": round(ds, 2),
                  "config": {"workload": "BASELINE configs[4], 1-GPU slice: " + W["name"] + f", EMA weights, DDIM-{a.steps}, w = {a.w}, sample batch {a.batch}",
                             "unet_rows_per_step": 2 * a.batch},
                  "roofline": {"bound": "mfma", "achieved": round(alg * fe / fl, 2), "peak": bench.PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(alg * fe / fl / bench.PEAK_FP32_MFMA_TFLOPS, 4), "algorithmic_tflops": round(alg, 2),
                               "frac_vs_direct_roofline": round(alg / bench.PEAK_FP32_MFMA_TFLOPS, 4),
                               "executed_share_of_algorithmic_flops": round(fe / fl, 4),
                               "recorded_gflop_per_unet_row": round(fl / 2 / (2 * a.batch) / 1e9, 2)},
                  "hbm_peak_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2), "finite": bool(torch.isfinite(out).all())}))
