"""Winograd vs direct 3x3 convolution at the benchmark's layer shapes (run on the GPU box): ms per launch, TFLOP/s on the
direct convolution's algorithmic FLOPs.  python tests/perf_wino.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H

DEV = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
SHAPES = [(128, 32, 32, 256, 256), (128, 32, 32, 512, 256), (128, 16, 16, 256, 256), (128, 16, 16, 512, 256), (128, 8, 8, 256, 256),
          (128, 8, 8, 512, 256), (256, 32, 32, 256, 256), (128, 64, 64, 192, 192), (128, 32, 32, 384, 384), (128, 16, 16, 576, 576),
          (128, 8, 8, 768, 768)]
for nimg, Hh, Ww, Cin, Cout in SHAPES:
    x = torch.randn(nimg, Hh, Ww, Cin, device=DEV)
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) * (9 * Cin) ** -0.5
    b = torch.randn(Cout, device=DEV)
    res = torch.randn(nimg, Hh, Ww, Cout, device=DEV)
    wf = torch.empty(Cout, 9, Cin, device=DEV)
    H.pack_conv3x3(w, Cout, Cin, wf=wf, Cin_p=Cin)
    uf = torch.empty(16, Cout, Cin, device=DEV)
    H.wino_pack(w, Cout, Cin, uf=uf)
    y = torch.empty(nimg, Hh, Ww, Cout, device=DEV)
    y2 = torch.empty_like(y)
    part = torch.empty(H.stats_part_numel(nimg, Hh * Ww, Cout), device=DEV)
    fl = 2.0 * nimg * Hh * Ww * Cout * 9 * Cin
    out = []
    for name, fn in (("direct", lambda: H.conv3x3(x, Cin, wf, b, y, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout, stats_part=part)),
                     ("wino", lambda: H.conv3x3_wino(x, Cin, uf, b, y2, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout, stats_part=part)),
                     ("wino-plain", lambda: H.conv3x3_wino(x, Cin, uf, None, y2, Cout, nimg, Hh, Ww, Cin, Cout))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out.append(f"{name} {ms:.3f} ms {fl / ms / 1e9:.1f} TF")
    H.conv3x3(x, Cin, wf, b, y, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout)
    H.conv3x3_wino(x, Cin, uf, b, y2, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout)
    err = (y - y2).abs().max().item()
    print(f"{nimg}x{Hh}x{Ww} {Cin}->{Cout}: " + " | ".join(out) + f" | max|direct-wino| {err:.2e}", flush=True)

print("---- weight gradient")
for nimg, Hh, Ww, Cin, Cout in SHAPES:
    x = torch.randn(nimg, Hh, Ww, Cin, device=DEV)
    dy = torch.randn(nimg, Hh, Ww, Cout, device=DEV)
    dw, dw2 = torch.empty(Cout, Cin, 3, 3, device=DEV), torch.empty(Cout, Cin, 3, 3, device=DEV)
    db = torch.empty(Cout, device=DEV)
    fl = 2.0 * nimg * Hh * Ww * Cout * 9 * Cin
    out = []
    for name, fn in (("direct", lambda: H.conv3x3_wgrad(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw, Cin, Cout, dbias=db, direct=True)),
                     ("wino", lambda: H.conv3x3_wgrad_wino(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw2, Cin, Cout, dbias=db))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out.append(f"{name} {ms:.3f} ms {fl / ms / 1e9:.1f} TF")
    rel = ((dw - dw2).norm() / dw.norm()).item()
    print(f"{nimg}x{Hh}x{Ww} {Cin}->{Cout}: " + " | ".join(out) + f" | rel-L2 direct vs wino {rel:.2e}", flush=True)
