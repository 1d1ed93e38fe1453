"""GroupNorm backward (vd_gn_apply_bwd) timing at the benchmark shapes; env knobs VD_GN_TPB / VD_GN_TWO_PASS select
variants (read once per process).   python tests/perf_gn.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H        # noqa: E402

dev = torch.device("cuda", 0)


def case(B, R, C, film, p_drop):
    x = torch.randn(B, R, R, C, device=dev)
    dy = torch.randn(B, R, R, C, device=dev)
    dx = torch.empty_like(x)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    fl = torch.randn(B, 2 * C, device=dev) * 0.1 if film else None
    dfilm = torch.empty(B, 2 * C, device=dev) if film else None
    stats = torch.empty(B, 32, 2, device=dev)
    coef = torch.empty(B, 4, C, device=dev)
    H.gn_stats(x, C, B, R * R, C, stats)
    y = torch.empty_like(x)
    H.gn_apply(x, C, stats, gamma, beta, fl, 1, p_drop, 1234, H.RS_NONE, y, C, B, R, R, C, coef)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)

    def fwd():
        H.gn_apply(x, C, None, gamma, beta, fl, 1, p_drop, 1234, H.RS_NONE, y, C, B, R, R, C, coef)
    fwd(); torch.cuda.synchronize()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    for _ in range(20):
        fwd()
    f1.record(); torch.cuda.synchronize()
    tf = f0.elapsed_time(f1) / 20
    print(f"B={B} {R}x{R} C={C} film={int(film)} p={p_drop}: apply {tf * 1e3:7.1f} us  {2 * 4.0 * B * R * R * C / tf / 1e9:6.2f} TB/s (x in, y out)   checksum {float(y.double().sum()):.6e}")

    def run():
        H.gn_apply_bwd(dy, C, x, C, coef, gamma, beta, fl, 1, p_drop, 1234, H.RS_NONE, None, 0, dx, C, False, dfilm, dg, db, False, B, R, R, C)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n
    nbytes = 3 * 4.0 * B * R * R * C
    print(f"B={B} {R}x{R} C={C} film={int(film)} p={p_drop}: {t * 1e3:7.1f} us  {nbytes / t / 1e9:6.2f} TB/s (x + dy in, dx out)   checksum {float(dx.double().sum()):.6e}", flush=True)


if __name__ == "__main__":
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("VD_GN")})
    case(128, 32, 256, False, 0.0)
    case(128, 32, 256, True, 0.2)
    case(128, 32, 512, False, 0.0)
    case(128, 16, 256, True, 0.2)
    case(128, 8, 256, True, 0.2)
    case(128, 64, 192, True, 0.1)
    case(128, 32, 384, True, 0.1)
    case(128, 64, 384, False, 0.0)
    case(128, 32, 576, True, 0.1)
    case(128, 16, 768, True, 0.1)
    case(128, 8, 768, True, 0.1)
    case(128, 8, 1536, False, 0.0)
    case(128, 16, 1536, False, 0.0)
