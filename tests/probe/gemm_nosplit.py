"""Timing of the 1x1-convolution GEMM shapes of the CIFAR-10 step on random data, looped alone (not a test):
   VDIFF_HIP_LIB=<library> python tests/probe/gemm_nosplit.py
Round 6 used it on timing-only builds of gemm.hip whose operand pieces were the raw bits (no split8) and on a build that read pre-split bf16
weight images: profiles/r06_presplit_price.txt (the first looked 13-22 % faster only because garbage operands lower the power draw; the
real pre-split form was 1-5 % slower: the split forms are power-bound, not bound by the split's vector instructions)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H                    # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [  # (label, M, N, K, a_kind, b_kind)
    ("fwd qkv  32x32", 131072, 768, 256, H.ROW, H.ROW),
    ("fwd proj 32x32", 131072, 256, 256, H.ROW, H.ROW),
    ("fwd skip 32x32", 131072, 256, 512, H.ROW, H.ROW),
    ("fwd qkv  16x16", 32768, 768, 256, H.ROW, H.ROW),
    ("dgrad qkv 32x32", 131072, 256, 768, H.ROW, H.COL),
    ("dgrad proj 32x32", 131072, 256, 256, H.ROW, H.COL),
    ("dgrad skip 32x32", 131072, 512, 256, H.ROW, H.COL),
    ("dgrad qkv 16x16", 32768, 256, 768, H.ROW, H.COL),
]
print("library:", H.LIB_PATH)
for label, M, N, K, ak, bk in SHAPES:
    A = torch.randn(M, K, device=dev)
    Bm = torch.randn((N, K) if bk == H.ROW else (K, N), device=dev) * 0.05
    Cm = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    kw = dict(a_kind=ak, b_kind=bk, lda=K, ldb=K if bk == H.ROW else N, ldc=N, bias=bias)
    for _ in range(20):
        H.gemm(A, Bm, Cm, M, N, K, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            H.gemm(A, Bm, Cm, M, N, K, **kw)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    print(f"{label:18s} M={M:6d} N={N:4d} K={K:4d}: {best * 1e3:7.1f} us  {2.0 * M * N * K / best / 1e9:6.1f} TFLOP/s fp32-equivalent")
