"""Same-box A/B of the two Winograd weight gradients (not a test): F(4x4,3x3) unfused (transforms + 36 grouped GEMMs + finish) against
the fused F(2x2,3x3) kernel of wino.hip, with the error of each against fp64.   python tests/perf_wgrad43.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H

DEV = "cuda"


def timeit(fn, n=10):
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


for nimg, Hh, Ww, Cin, Cout in ((128, 32, 32, 256, 256), (128, 32, 32, 512, 256), (128, 16, 16, 256, 256), (128, 16, 16, 512, 256),
                                (128, 8, 8, 256, 256), (128, 64, 64, 192, 192), (128, 32, 32, 384, 384),
                                (128, 8, 8, 768, 768), (128, 8, 8, 1536, 768), (128, 16, 16, 576, 576), (128, 16, 16, 1152, 576)):
    g = torch.Generator(DEV).manual_seed(1)
    x = F.silu(torch.randn((nimg, Hh, Ww, Cin), device=DEV, generator=g))
    dy = torch.randn((nimg, Hh, Ww, Cout), device=DEV, generator=g) * 0.05
    dw2, dw4 = torch.empty(Cout, Cin, 3, 3, device=DEV), torch.empty(Cout, Cin, 3, 3, device=DEV)
    db2, db4 = torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
    f2 = lambda: H.conv3x3_wgrad_wino(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw2, Cin, Cout, dbias=db2)
    f4 = lambda: H.conv3x3_wgrad_wino43(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw4, Cin, Cout, dbias=db4)
    t2 = timeit(f2)
    if not H.lib().vd_conv3x3_wgrad_wino43_supported(nimg, Hh, Ww, Cin, Cout, Cin, Cout):
        print(f"{nimg}x{Hh}x{Ww} {Cin}->{Cout}: F(2,3) {t2:.3f} ms | F(4,3) not served")
        continue
    t4 = timeit(f4)
    H.PROFILE = []
    f4()
    torch.cuda.synchronize()
    ph = [e0.elapsed_time(e1) for _, _, e0, e1 in H.PROFILE]
    H.PROFILE = None
    xp = F.pad(x.double(), (0, 0, 1, 1, 1, 1))
    d2 = dy.double().reshape(-1, Cout)
    ref = torch.empty(Cout, Cin, 3, 3, dtype=torch.float64, device=DEV)
    for ky in range(3):
        for kx in range(3):
            ref[:, :, ky, kx] = d2.T @ xp[:, ky:ky + Hh, kx:kx + Ww, :].reshape(-1, Cin)
    r2, r4 = ((dw2.double() - ref).norm() / ref.norm()).item(), ((dw4.double() - ref).norm() / ref.norm()).item()
    eb = (db4.double() - d2.sum(0)).abs().max().item() / max(d2.sum(0).abs().max().item(), 1e-30)
    fl = 2.0 * nimg * Hh * Ww * Cout * 9 * Cin
    print(f"{nimg}x{Hh}x{Ww} {Cin}->{Cout}: F(2,3) {t2:.3f} ms ({fl / t2 / 1e9:.0f} alg TF) rel-L2 {r2:.2e} | F(4,3) {t4:.3f} ms ({fl / t4 / 1e9:.0f} alg TF) "
          f"[transform {ph[0]:.3f} + gemm {ph[1]:.3f} + finish {ph[2]:.3f}] rel-L2 {r4:.2e} dbias rel {eb:.1e} | x{t2 / t4:.2f}")
