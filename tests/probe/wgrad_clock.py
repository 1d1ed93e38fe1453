"""Shader clock and cycle count of the Winograd weight-gradient kernel (stamps at kernel start / end, vd_wino_set_probe).
python tests/probe/wgrad_clock.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa: F401,E402  (loads libvdiff_hip_probe.so: the product library has no probe code)
from v_diffusion import _hip as H

DEV = "cuda"
# (the fused F(2x2,3x3) kernel itself: H.conv3x3_wgrad routes layers of this size to the unfused F(4x4,3x3) path since round 3)
for nimg, Hh, Ww, Cin, Cout in ((128, 32, 32, 256, 256), (128, 32, 32, 512, 256), (128, 16, 16, 256, 256), (128, 8, 8, 256, 256)):
    x = torch.randn(nimg, Hh, Ww, Cin, device=DEV)
    dy = torch.randn(nimg, Hh, Ww, Cout, device=DEV)
    dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
    db = torch.empty(Cout, device=DEV)
    for _ in range(3):
        H.conv3x3_wgrad_wino(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw, Cin, Cout, dbias=db)
    buf = torch.zeros(4 * 4096, dtype=torch.int64, device=DEV)
    H.lib().vd_wino_set_probe(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    H.conv3x3_wgrad_wino(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw, Cin, Cout, dbias=db)
    e1.record()
    torch.cuda.synchronize()
    H.lib().vd_wino_set_probe(None)
    t = buf.view(-1, 4).double().cpu()
    t = t[t[:, 3] > 0]
    clk = ((t[:, 2] - t[:, 0]) / (t[:, 3] - t[:, 1])).median() * 100
    span = (t[:, 3].max() - t[:, 1].min()) / 100
    cyc = (t[:, 2] - t[:, 0]).median()
    flops = 2.0 * nimg * Hh * Ww * Cout * 9 * Cin
    # MFMA-bound cycles of a workgroup (one per CU): its share of the 16 Winograd GEMMs (4/9 of the direct FLOPs) at 256 FLOP per cycle and CU
    mf = flops * 4 / 9 / t.shape[0] / 256
    print(f"{nimg}x{Hh}x{Ww} {Cin}->{Cout}: {t.shape[0]} workgroups, kernel {span:.1f} us (events incl. reduce {e0.elapsed_time(e1) * 1e3:.1f}), "
          f"shader clock {clk:.0f} MHz, cycles per workgroup {cyc:.0f} vs MFMA-bound {mf:.0f} ({mf / cyc:.2f})")
