"""Timing experiments on the Winograd convolution (WRONG results by construction): which part of a K tile costs what.
VD_WINO_EXP is read once per process: run   for e in 0 1 2 3 4; do VD_WINO_EXP=$e python tests/probe/wino_exp.py; done"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa: F401,E402  (loads libvdiff_hip_probe.so: the product library has no probe code)
from v_diffusion import _hip as H

DEV = "cuda"
names = {0: "full kernel", 1: "no transform arithmetic", 2: "+ no patch reads", 3: "+ no U-fragment reads", 4: "full, no tile barrier", 5: "MFMA + epilogue skeleton (no LDS reads, no DMA)"}
e = int(os.environ.get("VD_WINO_EXP", "0"))
for nimg, Hh, Ww, Cin, Cout in ((128, 32, 32, 256, 256), (128, 32, 32, 512, 256)):
    x = torch.randn(nimg, Hh, Ww, Cin, device=DEV)
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) * (9 * Cin) ** -0.5
    uf = torch.empty(16, Cout, Cin, device=DEV)
    H.wino_pack(w, Cout, Cin, uf=uf)
    y = torch.empty(nimg, Hh, Ww, Cout, device=DEV)
    fn = lambda: H.conv3x3_wino(x, Cin, uf, None, y, Cout, nimg, Hh, Ww, Cin, Cout)
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * nimg * Hh * Ww * Cout * 9 * Cin
    print(f"exp {e} ({names[e]}): {Cin}->{Cout}@{Hh}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF algorithmic, {fl * 4 / 9 / ms / 1e9:.1f} executed")
