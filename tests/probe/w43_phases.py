"""Per-item phase budget of wino43_conv_kernel from in-kernel stamps (probe library, vd_wino43_set_probe): where a persistent workgroup's
cycles go between the K loop and everything around it, and how long the waves sit at the K-tile barrier.
    python tests/probe/w43_phases.py            (GPU box; prints the table committed as profiles/r06_wino43_item_budget.txt)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _probe_lib  # noqa: F401,E402  (loads libvdiff_hip_probe.so: the product library has no probe code)
from v_diffusion import _hip as H

DEV = "cuda"
NAMES = ["first-stage DMA wait + barrier (the DMA was issued inside the previous item's epilogue)", "first row pass", "K loop", "post-loop barrier",
         "output transform pass 0 + exchange stores (+ residual issue)", "exchange barrier", "pass 1 + partner add + barrier (exchange area read)",
         "next item's first-stage DMA issue + bias / residual / statistics + 16 output stores issued", "statistics barrier + write"]


def time_plain(fn, n=10):
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


def report(name, fn, nwg, nkt, rounds):
    ms = time_plain(fn)
    buf = torch.zeros(nwg * 8 * 4 * 16, dtype=torch.int64, device=DEV)
    H.lib().vd_wino43_set_probe(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    ms_probe = time_plain(fn, 4)
    H.lib().vd_wino43_set_probe(None)
    t = buf.view(nwg, 8, 4, 16).double().cpu()
    print(f"== {name}: {ms:.3f} ms per launch without stamps, {ms_probe:.3f} with; {nwg} workgroups x {rounds} item rounds, {nkt} K tiles per item")
    r = min(rounds, 4)
    for half, waves in (("waves 0-3 (xi rows 0-2)", slice(0, 4)), ("waves 4-7 (xi rows 3-5)", slice(4, 8))):
        tt = t[:, waves, 1:r] if r > 1 else t[:, waves, 0:1]          # steady-state rounds (skip the first: cold start)
        d = [(tt[..., i + 1] - tt[..., i]).median().item() for i in range(9)]
        item = (tt[..., 9] - tt[..., 0]).median().item()
        print(f"  {half}: item {item:.0f} cycles = {item / nkt:.0f} per K tile all-in; K loop {d[2]:.0f} = {d[2] / nkt:.0f} per K tile (MFMA-bound 4608 per SIMD pair)")
        for i, nm in enumerate(NAMES):
            print(f"      {d[i]:9.0f}  {100 * d[i] / item:5.1f} %  {nm}")
        vm, bar = tt[..., 10].median().item(), tt[..., 11].median().item()
        print(f"      inside the K loop: waiting for own DMA (vmcnt) {vm:.0f} = {vm / nkt:.0f} per tile, parked at the tile barrier {bar:.0f} = {bar / nkt:.0f} per tile")
    # gap between items of one wave, and the chip-wide phase of item starts (100 MHz clock)
    if r > 1:
        gap = (t[:, :, 1:r, 0] - t[:, :, 0:r - 1, 9]).median().item()
        rt = t[:, 0, 1, 12]
        print(f"  gap between an item's last stamp and the next item's first: {gap:.0f} cycles; start of item round 1 across workgroups: spread {(rt.max() - rt.min()) / 100:.2f} us "
              f"(std {rt.std() / 100:.2f} us)")
    clk = ((t[:, 0, r - 1, 0] - t[:, 0, 0, 0]) / (t[:, 0, r - 1, 12] - t[:, 0, 0, 12]).clamp(min=1)).median().item() * 100 if r > 1 else 0
    print(f"  shader clock over this ONE stamped launch: {clk:.0f} MHz (s_memtime ticks per s_memrealtime microsecond; a handful of launches after idle -- "
          f"the clock the chip holds under sustained load is profiles/r05_clock_by_kernel.txt: 2.37 GHz)")


for nimg, Hh, Ww, Cin, Cout in ((128, 32, 32, 256, 256), (128, 32, 32, 512, 256), (128, 16, 16, 256, 256), (128, 64, 64, 192, 192)):
    g = torch.Generator(DEV).manual_seed(1)
    x = torch.nn.functional.silu(torch.randn((nimg, Hh, Ww, Cin), device=DEV, generator=g))
    w = torch.randn((Cout, Cin, 3, 3), device=DEV, generator=g) * (9 * Cin) ** -0.5
    bias = torch.randn((Cout,), device=DEV, generator=g)
    res = torch.randn((nimg, Hh, Ww, Cout), device=DEV, generator=g)
    u43f = torch.empty(H.lib().vd_wino43_u_floats(Cout, Cin), device=DEV)
    H.wino43_pack_fwd(w, Cout, Cin, u43f)
    y = torch.empty(nimg, Hh, Ww, Cout, device=DEV)
    part = torch.empty(H.stats_part_numel(nimg, Hh * Ww, Cout), device=DEV)
    grp = nimg // 4 if Ww == 16 else nimg * (Hh // 16 if Ww == 64 else 1)
    items = grp * (Cout // 32)
    nwg = min(256, items)
    rounds = (items + nwg - 1) // nwg
    report(f"forward {Cin}->{Cout} @{Hh}x{Ww} B={nimg}, bias + residual + statistics",
           lambda: H.conv3x3_wino43_fwd(x, Cin, u43f, bias, y, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout, stats_part=part), nwg, Cin // 8, rounds)
    report(f"forward {Cin}->{Cout} @{Hh}x{Ww} B={nimg}, bias + statistics, no residual",
           lambda: H.conv3x3_wino43_fwd(x, Cin, u43f, bias, y, Cout, nimg, Hh, Ww, Cin, Cout, stats_part=part), nwg, Cin // 8, rounds)
    if Cin == Cout:
        dy = torch.randn((nimg, Hh, Ww, Cout), device=DEV, generator=g)
        u43 = torch.empty(H.lib().vd_wino43_u_floats(Cout, Cin), device=DEV)
        H.wino43_pack(w, Cout, Cin, u43)
        dx = torch.empty(nimg, Hh, Ww, Cin, device=DEV)
        report(f"input gradient {Cout}->{Cin} @{Hh}x{Ww} B={nimg}", lambda: H.conv3x3_dgrad_wino43(dy, Cout, u43, dx, Cin, nimg, Hh, Ww, Cin, Cout), nwg, Cout // 8, rounds)
