"""Tail / prologue experiments for the conv kernel (not a test).  python tests/perf_tail.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
import torch
from v_diffusion import _hip as H
from perf_kernels import timeit

DEV = "cuda"


def conv(B, Hh, Cin, Cout, tag=""):
    x = torch.randn(B, Hh, Hh, Cin, device=DEV)
    w = torch.randn(Cout, 9, Cin, device=DEV) * 0.02
    bias = torch.randn(Cout, device=DEV)
    y = torch.empty(B, Hh, Hh, Cout, device=DEV)
    fl = 2.0 * B * Hh * Hh * Cout * 9 * Cin
    timeit(lambda: H.conv3x3(x, Cin, w, bias, y, Cout, B, Hh, Hh, Cin, Cout), fl,
           f"{tag} conv {Cin}->{Cout} @{Hh} B={B} blocks={B * Hh * Hh // 128 * (Cout // 128)} tile={H.lib().vd_gemm_last_tile()}")


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "blocks"):
    for B in (64, 80, 128, 160, 240, 256, 320):
        conv(B, 32, 256, 256, "blocks")
if which in ("all", "k"):
    for Cin in (128, 256, 512, 1024, 2048):
        conv(128, 32, Cin, 256, "k")
