"""Same-box A/B of the two input-gradient kernels (not a test): Winograd F(4x4,3x3) (csrc/wino43.hip) against F(2x2,3x3) (wino.hip) at
the benchmark's layer shapes, plus the error of each against fp64.   python tests/perf_wino43.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H

DEV = "cuda"


def timeit(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def conv_fp64(x, w):
    B, Hh, Ww, Cin = x.shape
    xp = F.pad(x.double(), (0, 0, 1, 1, 1, 1))
    out = torch.zeros(B, Hh, Ww, w.shape[0], dtype=torch.float64, device=DEV)
    for ky in range(3):
        for kx in range(3):
            out += xp[:, ky:ky + Hh, kx:kx + Ww, :] @ w[:, :, ky, kx].double().T
    return out


SHAPES = ((128, 32, 32, 256, 256), (128, 32, 32, 512, 256), (128, 64, 64, 192, 192), (128, 32, 32, 384, 384),
          (128, 16, 16, 256, 256), (128, 16, 16, 512, 256), (128, 16, 16, 576, 576))
for nimg, Hh, Ww, Cin, Cout in SHAPES[:int(os.environ.get("VD_PERF_SHAPES", len(SHAPES)))]:
    g = torch.Generator(DEV).manual_seed(1)
    dy = torch.randn((nimg, Hh, Ww, Cout), device=DEV, generator=g)
    w = torch.randn((Cout, Cin, 3, 3), device=DEV, generator=g) * (9 * Cin) ** -0.5
    ud = torch.empty(16, Cin, Cout, device=DEV)
    H.wino_pack(w, Cout, Cin, ud=ud)
    u43 = torch.empty(H.lib().vd_wino43_u_floats(Cout, Cin), device=DEV)
    H.wino43_pack(w, Cout, Cin, u43)
    dx2, dx4 = torch.empty(nimg, Hh, Ww, Cin, device=DEV), torch.full((nimg, Hh, Ww, Cin), 7.0, device=DEV)
    f2 = lambda: H.conv3x3_wino(dy, Cout, ud, None, dx2, Cin, nimg, Hh, Ww, Cout, Cin)
    f4 = lambda: H.conv3x3_dgrad_wino43(dy, Cout, u43, dx4, Cin, nimg, Hh, Ww, Cin, Cout)
    t2, t4 = timeit(f2), timeit(f4)
    ref = conv_fp64(dy, w.flip(2, 3).transpose(0, 1).contiguous())
    sc = ref.abs().max().item()
    e2, e4 = (dx2.double() - ref).abs().max().item(), (dx4.double() - ref).abs().max().item()
    r2, r4 = ((dx2.double() - ref).norm() / ref.norm()).item(), ((dx4.double() - ref).norm() / ref.norm()).item()
    fl = 2.0 * nimg * Hh * Ww * Cout * 9 * Cin
    # forward pass (round 4): F(2x2,3x3) with GroupNorm partials + bias + residual against the F(4x4,3x3) forward kernel, same operands
    x = torch.nn.functional.silu(torch.randn((nimg, Hh, Ww, Cin), device=DEV, generator=g))
    bias = torch.randn((Cout,), device=DEV, generator=g)
    res = torch.randn((nimg, Hh, Ww, Cout), device=DEV, generator=g)
    uf = torch.empty(16, Cout, Cin, device=DEV)
    H.wino_pack(w, Cout, Cin, uf=uf)
    u43f = torch.empty(H.lib().vd_wino43_u_floats(Cout, Cin), device=DEV)
    H.wino43_pack_fwd(w, Cout, Cin, u43f)
    y2, y4 = torch.empty(nimg, Hh, Ww, Cout, device=DEV), torch.empty(nimg, Hh, Ww, Cout, device=DEV)
    part = torch.empty(H.stats_part_numel(nimg, Hh * Ww, Cout), device=DEV)
    g2 = lambda: H.conv3x3_wino(x, Cin, uf, bias, y2, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout, stats_part=part)
    g4 = lambda: H.conv3x3_wino43_fwd(x, Cin, u43f, bias, y4, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout, stats_part=part)
    g4n = lambda: H.conv3x3_wino43_fwd(x, Cin, u43f, bias, y4, Cout, nimg, Hh, Ww, Cin, Cout, stats_part=part)
    tf2, tf4, tf4n = timeit(g2), timeit(g4), timeit(g4n)
    reff = conv_fp64(x, w) + bias.double() + res.double()
    ef2, ef4 = (y2.double() - reff).abs().max().item(), 0.0
    g4(); torch.cuda.synchronize()
    ef4 = (y4.double() - reff).abs().max().item()
    print(f"{nimg}x{Hh}x{Ww} {Cin}->{Cout} FORWARD: F(2,3) {tf2:.3f} ms err {ef2:.2e} | F(4,3) fwd {tf4:.3f} ms ({fl / 4 / tf4 / 1e9:.0f} exe) err {ef4:.2e}, "
          f"without residual {tf4n:.3f} ms | input gradient {t4:.3f} ms | x{tf2 / tf4:.2f}")
    print(f"{nimg}x{Hh}x{Ww} {Cout}->{Cin}: F(2,3) {t2:.3f} ms ({fl / t2 / 1e9:.0f} alg TF) err {e2:.2e} rel-L2 {r2:.2e} | "
          f"F(4,3) {t4:.3f} ms ({fl / t4 / 1e9:.0f} alg TF, {fl / 4 / t4 / 1e9:.0f} exe) err {e4:.2e} rel-L2 {r4:.2e} | scale {sc:.1f} | x{t2 / t4:.2f}")

# ---- localisation aid when a kernel change breaks the result: one image, one K tile, delta kernels
if os.environ.get("VD_W43_DEBUG"):
    nimg, Hh, Ww, Cin, Cout = 1, 32, 32, 32, 8
    dy = torch.randn((nimg, Hh, Ww, Cout), device=DEV)
    for tap in ((1, 1), (0, 0), (2, 1)):
        w = torch.zeros((Cout, Cin, 3, 3), device=DEV)
        w[3, 5, tap[0], tap[1]] = 1.0
        u43 = torch.empty(H.lib().vd_wino43_u_floats(Cout, Cin), device=DEV)
        H.wino43_pack(w, Cout, Cin, u43)
        dx4 = torch.full((nimg, Hh, Ww, Cin), 7.0, device=DEV)
        H.conv3x3_dgrad_wino43(dy, Cout, u43, dx4, Cin, nimg, Hh, Ww, Cin, Cout)
        ref = conv_fp64(dy, w.flip(2, 3).transpose(0, 1).contiguous())
        err = (dx4.double() - ref).abs()
        print("tap", tap, "max err", err.max().item(), "per-channel max err (nonzero):",
              {c: round(err[..., c].max().item(), 3) for c in range(Cin) if err[..., c].max().item() > 1e-4})
        e5 = err[0, :, :, 5]
        print("  channel 5 error by (y % 4, x % 4):", [[round(e5[u::4, v::4].max().item(), 3) for v in range(4)] for u in range(4)])
