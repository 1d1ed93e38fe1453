"""Same-box A/B: fused attention kernels (csrc/attn.hip) against the three-launch path (vd_gemm -> vd_softmax_rows -> vd_gemm and its
five-launch backward) at the attention shapes of the two benchmark models.   python tests/perf_attn.py"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "v-diffusion-torch_amd"))
from v_diffusion import _hip as H        # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(B, nh, L, hd, bwd=True):
    hid, ld = nh * hd, 3 * nh * hd
    qkv = torch.randn(B, L, ld, device=dev)
    f = qkv.reshape(-1)
    q, k, v = f[0:], f[hid:], f[2 * hid:]
    alpha = 1.0 / math.sqrt(hd)
    O1, O2 = torch.empty(B, L, hid, device=dev), torch.empty(B, L, hid, device=dev)
    lse, delta = torch.empty(B * nh * L, device=dev), torch.empty(B * nh * L, device=dev)
    S = torch.empty(B, nh, L, L, device=dev)
    sP, sQ, sO = (nh * L * L, L * L), (L * ld, hd), (L * hid, hd)

    def unfused_fwd():
        H.gemm(q, k, S, L, L, hd, a_kind=H.ROW, b_kind=H.ROW, lda=ld, ldb=ld, ldc=L, batch=B * nh, nh=nh, sA=sQ, sB=sQ, sC=sP, alpha=alpha)
        H.softmax_rows(S, B * nh * L, L)
        H.gemm(S, v, O1, L, hd, L, a_kind=H.ROW, b_kind=H.COL, lda=L, ldb=ld, ldc=hid, batch=B * nh, nh=nh, sA=sP, sB=sQ, sC=sO)

    def fused_fwd():
        H.attn_fwd(q, k, v, ld, O2, hid, lse, B, nh, L, hd, alpha)

    tu, tf = timeit(unfused_fwd), timeit(fused_fwd)
    err = (O1 - O2).abs().max().item()
    fl = 4.0 * B * nh * L * L * hd
    line = f"B={B} nh={nh} L={L} hd={hd}: fwd unfused {tu:7.3f} ms | fused {tf:7.3f} ms ({fl / tf / 1e9:6.1f} TF) | max|diff| {err:.2e}"
    if bwd and H.attn_supported(L, hd, True):
        dO = torch.randn(B, L, hid, device=dev)
        dqkv1, dqkv2 = torch.empty_like(qkv), torch.empty_like(qkv)
        d1, d2 = dqkv1.reshape(-1), dqkv2.reshape(-1)
        dP = torch.empty(B, nh, L, L, device=dev)

        def unfused_bwd():
            H.gemm(S, dO, d1[2 * hid:], L, hd, L, a_kind=H.COL, b_kind=H.COL, lda=L, ldb=hid, ldc=ld, batch=B * nh, nh=nh, sA=sP, sB=sO, sC=sQ)
            H.gemm(dO, v, dP, L, L, hd, a_kind=H.ROW, b_kind=H.ROW, lda=hid, ldb=ld, ldc=L, batch=B * nh, nh=nh, sA=sO, sB=sQ, sC=sP)
            H.softmax_rows_bwd(S, dP, B * nh * L, L, alpha)
            H.gemm(dP, k, d1[0:], L, hd, L, a_kind=H.ROW, b_kind=H.COL, lda=L, ldb=ld, ldc=ld, batch=B * nh, nh=nh, sA=sP, sB=sQ, sC=sQ)
            H.gemm(dP, q, d1[hid:], L, hd, L, a_kind=H.COL, b_kind=H.COL, lda=L, ldb=ld, ldc=ld, batch=B * nh, nh=nh, sA=sP, sB=sQ, sC=sQ)

        def fused_bwd():
            H.attn_bwd(q, k, v, ld, O2, hid, dO, hid, lse, delta, d2[0:], d2[hid:], d2[2 * hid:], ld, B, nh, L, hd, alpha)

        unfused_fwd(); fused_fwd()
        tub, tfb = timeit(unfused_bwd), timeit(fused_bwd)
        errb = (dqkv1 - dqkv2).abs().max().item() / dqkv1.abs().max().item()
        line += f" || bwd unfused {tub:7.3f} ms | fused {tfb:7.3f} ms ({2 * fl / tfb / 1e9:6.1f} TF alg) | rel diff {errb:.2e}"
    print(line, flush=True)


if __name__ == "__main__":
    print("---- CelebA 64x64 model (hd 64), batch 128")
    for L in (64, 256, 1024, 4096):
        case(128, 1, L, 64)
    print("---- CIFAR-10 model (hd 256): sampling (256 rows, forward) and training (128 rows, forward + backward)")
    for L in (64, 256, 1024):
        case(256, 1, L, 256, bwd=False)
    for L in (64, 256, 1024):
        case(128, 1, L, 256)
    print("---- hd 128")
    case(64, 2, 1024, 128)
