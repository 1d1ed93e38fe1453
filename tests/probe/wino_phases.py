"""Where a Winograd workgroup spends its cycles (vd_wino_set_probe): prologue / K loop / epilogue, cycles parked at the K-tile
barrier, first K tile.  python tests/probe/wino_phases.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa: F401,E402  (loads libvdiff_hip_probe.so: the product library has no probe code)
from v_diffusion import _hip as H

DEV = "cuda"
for nimg, Hh, Ww, Cin, Cout, stats in ((128, 32, 32, 256, 256, False), (128, 32, 32, 256, 256, True), (128, 16, 16, 512, 256, False)):
    x = torch.randn(nimg, Hh, Ww, Cin, device=DEV)
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) * (9 * Cin) ** -0.5
    uf = torch.empty(16, Cout, Cin, device=DEV)
    H.wino_pack(w, Cout, Cin, uf=uf)
    y = torch.empty(nimg, Hh, Ww, Cout, device=DEV)
    res = torch.randn_like(y)
    b = torch.randn(Cout, device=DEV)
    part = torch.empty(H.stats_part_numel(nimg, Hh * Ww, Cout), device=DEV)
    TPW = 64 if os.environ.get("VD_WINO_WIDE") == "0" or Hh < 16 else 128     # tiles per item (the wide form serves 16x16 and larger)
    nwg = min(256, (Cout // 32) * (nimg * Hh * Ww // 4 // TPW))      # persistent workgroups: one per CU
    nitems = (Cout // 32) * (nimg * Hh * Ww // 4 // TPW)
    buf = torch.zeros(nwg * 64, dtype=torch.int64, device=DEV)
    for _ in range(3):
        H.conv3x3_wino(x, Cin, uf, b if stats else None, y, Cout, nimg, Hh, Ww, Cin, Cout, res=res if stats else None, ldres=Cout,
                       stats_part=part if stats else None)
    H.lib().vd_wino_set_probe(buf.data_ptr())
    H.conv3x3_wino(x, Cin, uf, b if stats else None, y, Cout, nimg, Hh, Ww, Cin, Cout, res=res if stats else None, ldres=Cout,
                   stats_part=part if stats else None)
    torch.cuda.synchronize()
    H.lib().vd_wino_set_probe(None)
    t = buf.view(nwg, 8, 8).double().cpu()
    pro, loop, epi, wait, nkt = (t[..., 1] - t[..., 0]), (t[..., 2] - t[..., 1]), (t[..., 3] - t[..., 2]), t[..., 4], t[..., 6]
    rt = t[..., 7]
    per = nitems / nwg
    tot = t[..., 3] - t[..., 0]
    print(f"{nimg}x{Hh}x{Ww} {Cin}->{Cout} stats={stats}: workgroups {nwg} x {per:.0f} items, K tiles per item {int(nkt.max())}")
    print(f"  cycles per wave (median): whole kernel {tot.median():.0f} = {tot.median() / per:.0f} per item = {tot.median() / per / nkt.max():.0f} per K tile "
          f"(MFMA-bound: {4096 * TPW // 64}); parked at barriers {wait.median():.0f} ({100 * wait.median() / tot.median():.1f} %); prologue of the first item {pro.median():.0f}")
    print(f"  span of workgroup end times: {(rt.max() - rt.min()) / 100:.1f} us (100 MHz clock)")
    rt0 = t[..., 5]
    clk = (tot / (rt - rt0)).median() * 100.0                # s_memtime ticks per 100 MHz s_memrealtime tick
    print(f"  s_memtime ticks per microsecond while this kernel runs: {clk:.0f} (MHz if s_memtime counts shader clocks); kernel {(rt.max() - rt0.min()) / 100:.1f} us")
