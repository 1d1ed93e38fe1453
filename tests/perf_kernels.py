"""Kernel micro-benchmark (not a test): times the matmul-shaped kernels at the CIFAR bs=128 shapes with HIP events.
    python tests/perf_kernels.py [filter]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "v-diffusion-torch_amd")]
import torch
from v_diffusion import _hip as H

DEV = "cuda"
PEAK = 157.3


def timeit(fn, flops, name, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = flops / ms / 1e9
    print(f"{name:58s} {ms:8.3f} ms  {tf:7.1f} TFLOP/s  {100 * tf / PEAK:5.1f}%", flush=True)
    return ms


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    B = 128
    for (Hh, Cin, Cout) in [(32, 256, 256), (16, 256, 256), (8, 256, 256), (32, 512, 256), (16, 512, 256), (8, 512, 256)]:
        x = torch.randn(B, Hh, Hh, Cin, device=DEV)
        w = torch.randn(Cout, 9, Cin, device=DEV) * 0.02
        wd = torch.randn(Cin, 9, Cout, device=DEV) * 0.02
        bias = torch.randn(Cout, device=DEV)
        y = torch.empty(B, Hh, Hh, Cout, device=DEV)
        dx = torch.empty(B, Hh, Hh, Cin, device=DEV)
        dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
        fl = 2.0 * B * Hh * Hh * Cout * 9 * Cin
        if "fwd".startswith(flt) or flt in "fwd":
            timeit(lambda: H.conv3x3(x, Cin, w, bias, y, Cout, B, Hh, Hh, Cin, Cout), fl, f"conv3x3 fwd   {Cin}->{Cout} @{Hh}x{Hh} B={B}")
        if flt in "dgrad":
            timeit(lambda: H.conv3x3(y, Cout, wd, None, dx, Cin, B, Hh, Hh, Cout, Cin), fl, f"conv3x3 dgrad {Cout}->{Cin} @{Hh}x{Hh} B={B}")
        if flt in "wgrad":
            timeit(lambda: H.conv3x3_wgrad(x, Cin, y, Cout, B, Hh, Hh, Cin, Cout, dw, Cin, Cout), fl, f"conv3x3 wgrad {Cin}->{Cout} @{Hh}x{Hh} B={B}")
    if flt in "gemm":
        for (M, N, K, ak, bk, nm) in [(B * 1024, 768, 256, 0, 0, "qkv 1x1 @32"), (B * 256, 768, 256, 0, 0, "qkv 1x1 @16"),
                                      (B * 256, 256, 768, 0, 1, "qkv dgrad @16"), (768, 256, B * 256, 1, 1, "qkv wgrad @16")]:
            A = torch.randn((M, K) if ak == 0 else (K, M), device=DEV)
            Bm = torch.randn((N, K) if bk == 0 else (K, N), device=DEV)
            C = torch.empty(M, N, device=DEV)
            sk = 1 if ak == 0 else max(1, min(64, 512 // (((M + 63) // 64) * ((N + 63) // 64)), K // 256))
            timeit(lambda: H.gemm(A, Bm, C, M, N, K, a_kind=ak, b_kind=bk, lda=A.shape[1], ldb=Bm.shape[1], ldc=N, splitk=sk),
                   2.0 * M * N * K, f"gemm {nm} M={M} N={N} K={K} kinds=({ak},{bk}) splitk={sk}")
        for L, hd in [(256, 256), (64, 256), (1024, 256)]:
            qkv = torch.randn(B, L, 768, device=DEV)
            S = torch.empty(B, 1, L, L, device=DEV)
            O = torch.empty(B, L, 256, device=DEV)
            timeit(lambda: H.gemm(qkv, qkv[0, 0, 256:], S, L, L, hd, lda=768, ldb=768, ldc=L, batch=B, nh=1, sA=(L * 768, hd),
                                  sB=(L * 768, hd), sC=(L * L, L * L), alpha=1 / 16), 2.0 * B * L * L * hd, f"attn QK^T L={L} hd={hd}")
            timeit(lambda: H.gemm(S, qkv[0, 0, 512:], O, L, hd, L, a_kind=0, b_kind=1, lda=L, ldb=768, ldc=256, batch=B, nh=1,
                                  sA=(L * L, L * L), sB=(L * 768, hd), sC=(L * 256, hd)), 2.0 * B * L * L * hd, f"attn PV   L={L} hd={hd}")


if __name__ == "__main__":
    main()
