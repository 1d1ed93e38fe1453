"""GroupNorm backward micro-benchmark (not a test).  python tests/perf_gn.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
import torch
from v_diffusion import _hip as H

DEV = "cuda"
for (B, R, C, film, drop) in [(128, 32, 256, True, 0.2), (128, 32, 256, False, 0.0), (128, 16, 256, True, 0.2), (128, 32, 512, False, 0.0),
                              (128, 8, 256, True, 0.2)]:
    x = torch.randn(B, R, R, C, device=DEV)
    dy = torch.randn(B, R, R, C, device=DEV)
    dx = torch.empty_like(x)
    gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    fl = torch.randn(B, 2 * C, device=DEV) * 0.1 if film else None
    dfl = torch.empty(B, 2 * C, device=DEV) if film else None
    stats, coef = torch.empty(B, 32, 2, device=DEV), torch.empty(B, 4, C, device=DEV)
    H.gn_stats(x, C, B, R * R, C, stats)
    y = torch.empty_like(x)
    H.gn_apply(x, C, stats, gamma, beta, fl, 1, drop, 123, H.RS_NONE, y, C, B, R, R, C, coef)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    fn = lambda: H.gn_apply_bwd(dy, C, x, C, coef, gamma, beta, fl, 1, drop, 123, H.RS_NONE, None, 0, dx, C, False, dfl, dg, db, False, B, R, R, C)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = 3 * x.numel() * 4 / 1e6
    print(f"gn bwd B={B} {R}x{R} C={C} film={film} drop={drop}: {us:8.1f} us  {mb / us * 1e-3 * 1e3:7.2f} GB/s (3 tensors)", flush=True)
