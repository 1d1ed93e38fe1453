"""Per-kernel parity tests of libvdiff_hip.so, called through the C ABI (ctypes), against plain PyTorch CPU
references of the same op (fp64 as truth, torch fp32 as the natural-noise yardstick).  Need an MI355X."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from v_diffusion import _hip
    _hip.lib()
    return _hip


def rnd(*shape, seed=0, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed + 7919 * len(shape) + sum(shape))
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def close(got, ref64, ref32=None, slack=4.0, floor=2e-6, name=""):
    """|got - ref64| must be within `slack` x the error torch's own fp32 result has (plus a relative floor)."""
    got = got.detach().cpu().double()
    ref64 = ref64.detach().double()
    scale = max(ref64.abs().max().item(), 1e-30)
    err = (got - ref64).abs().max().item()
    nat = 0.0 if ref32 is None else (ref32.detach().double() - ref64).abs().max().item()
    tol = slack * nat + floor * scale
    assert math.isfinite(err) and err <= tol, f"{name}: max err {err:.3e} > tol {tol:.3e} (natural fp32 noise {nat:.3e}, scale {scale:.3e})"
    return err


# ------------------------------------------------------------------------------------------------ GEMM engine
GEMM_SHAPES = [(128, 128, 32), (64, 64, 64), (200, 72, 100), (256, 512, 1024), (1024, 256, 2304), (36, 4, 8), (132, 260, 36)]


@pytest.mark.parametrize("a_kind,b_kind", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("tile", [0, 64, 128, 12864, 64128])
def test_gemm_kinds(H, a_kind, b_kind, M, N, K, tile):
    A = rnd(M, K, seed=1)
    B = rnd(N, K, seed=2)
    bias = rnd(N, seed=3)
    R = rnd(M, N, seed=4)
    ref64 = 0.5 * (A.double() @ B.double().T) + bias.double() + R.double()
    ref32 = 0.5 * (A @ B.T) + bias + R
    Ad = (A if a_kind == 0 else A.T.contiguous()).to(DEV)
    Bd = (B if b_kind == 0 else B.T.contiguous()).to(DEV)
    Cd = torch.full((M, N + 4), 7.0, device=DEV)                # ldc > N: untouched padding must survive
    H.gemm(Ad, Bd, Cd, M, N, K, a_kind=a_kind, b_kind=b_kind, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N + 4,
           bias=bias.to(DEV), R=R.to(DEV), ldr=N, alpha=0.5, tile=tile)
    torch.cuda.synchronize()
    close(Cd[:, :N], ref64, ref32, name=f"gemm{a_kind}{b_kind}")
    assert (Cd[:, N:] == 7.0).all()


def test_gemm_batched_heads_accumulate(H):
    Bz, nh, L, hd = 3, 2, 64, 32
    qkv = rnd(Bz, L, 3 * nh * hd, seed=5)
    q = qkv[..., : nh * hd].reshape(Bz, L, nh, hd)
    k = qkv[..., nh * hd: 2 * nh * hd].reshape(Bz, L, nh, hd)
    ref64 = torch.einsum("blnc,bmnc->bnlm", q.double(), k.double()) / math.sqrt(hd)
    qd = qkv.to(DEV)
    S = torch.ones(Bz, nh, L, L, device=DEV)
    ld = 3 * nh * hd
    H.gemm(qd, qd[0, 0, nh * hd:], S, L, L, hd, a_kind=0, b_kind=0, lda=ld, ldb=ld, ldc=L, batch=Bz * nh, nh=nh,
           sA=(L * ld, hd), sB=(L * ld, hd), sC=(nh * L * L, L * L), alpha=1 / math.sqrt(hd), accumulate=True)
    torch.cuda.synchronize()
    close(S, ref64 + 1.0, None, floor=3e-6, name="batched QK^T")


@pytest.mark.parametrize("M,N,K,splitk", [(64, 96, 4096, 8), (256, 36, 128, 1), (768, 256, 8192, 6), (12, 260, 1000, 3)])
@pytest.mark.parametrize("tile", [0, 128, 64, 12864, 64128])
def test_gemm_colsum_rides_on_wgrad(H, M, N, K, splitk, tile):
    """bias gradient = column sums of the COL-kind A operand, produced by the weight-gradient GEMM itself"""
    A, B = rnd(K, M, seed=8), rnd(K, N, seed=9)
    Cd = torch.zeros(M, N, device=DEV)
    cs = torch.ones(M, device=DEV)
    H.gemm(A.to(DEV), B.to(DEV), Cd, M, N, K, a_kind=1, b_kind=1, lda=M, ldb=N, ldc=N, splitk=splitk, tile=tile, colsum=cs,
           colsum_accumulate=True)
    torch.cuda.synchronize()
    close(Cd, A.double().T @ B.double(), A.T @ B, name="wgrad gemm")
    close(cs, A.double().sum(0) + 1.0, A.sum(0) + 1.0, name="colsum")


@pytest.mark.parametrize("kinds", [(1, 1), (0, 0), (0, 1)])
def test_gemm_splitk(H, kinds):
    a_kind, b_kind = kinds
    M, N, K = 64, 96, 4096
    A, B = rnd(M, K, seed=6), rnd(N, K, seed=7)
    ref64, ref32 = A.double() @ B.double().T, A @ B.T
    Ad = (A if a_kind == 0 else A.T.contiguous()).to(DEV)
    Bd = (B if b_kind == 0 else B.T.contiguous()).to(DEV)
    Cd = torch.ones(M, N, device=DEV)
    H.gemm(Ad, Bd, Cd, M, N, K, a_kind=a_kind, b_kind=b_kind, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, splitk=8, accumulate=True)
    torch.cuda.synchronize()
    close(Cd, ref64 + 1.0, ref32 + 1.0, name="splitk")


def test_gemm_rejects_bad_input(H):
    A = torch.zeros(8, 6, device=DEV)
    with pytest.raises(RuntimeError):
        H.gemm(A, A, A, 8, 8, 6, lda=6, ldb=6, ldc=8)          # lda not a multiple of 4
    with pytest.raises(RuntimeError):
        H.gemm(torch.zeros(4, 4), A, A, 4, 4, 4, lda=4, ldb=4, ldc=4)   # CPU tensor: no fallback


# ------------------------------------------------------------------------------------------------ conv 3x3
def nhwc(x, ld=None):
    """NCHW cpu -> NHWC device tensor with per-pixel stride ld (extra channels filled with NaN-free junk)."""
    B, Cc, Hh, Ww = x.shape
    ld = ld or Cc
    out = torch.full((B, Hh, Ww, ld), 3.0)
    out[..., :Cc] = x.permute(0, 2, 3, 1)
    return out.to(DEV)


def from_nhwc(y, Cc):
    return y[..., :Cc].permute(0, 3, 1, 2).cpu()


CONV_CASES = [  # nimg, H, W, Cin, Cout, ldx_extra, ldy_extra
    (4, 32, 32, 192, 192, 0, 0), (2, 16, 16, 576, 320, 0, 0), (64, 16, 16, 64, 192, 0, 0),
    (2, 8, 8, 32, 64, 0, 0), (3, 16, 16, 64, 32, 32, 64), (1, 32, 32, 256, 256, 0, 0), (2, 16, 16, 4, 32, 0, 0),
    (2, 8, 8, 32, 3, 0, 1), (2, 4, 4, 96, 160, 0, 0), (5, 8, 8, 36, 68, 4, 4), (1, 2, 2, 8, 8, 0, 0), (2, 64, 64, 32, 32, 0, 0)]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_forward(H, case):
    nimg, Hh, Ww, Cin, Cout, ex, ey = case
    x, w, b = rnd(nimg, Cin, Hh, Ww, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=3)
    res = rnd(nimg, Cout, Hh, Ww, seed=4)
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    ref32 = F.conv2d(x, w, b, padding=1) + res
    xd, rd = nhwc(x, Cin + ex), nhwc(res, Cout + ey)
    wf = torch.empty(Cout, 9, Cin, device=DEV)
    H.pack_conv3x3(w.to(DEV), Cout, Cin, wf=wf, Cin_p=Cin)
    y = torch.full((nimg, Hh, Ww, Cout + ey), 5.0, device=DEV)
    H.conv3x3(xd, Cin + ex, wf, b.to(DEV), y, Cout + ey, nimg, Hh, Ww, Cin, Cout, res=rd, ldres=Cout + ey)
    torch.cuda.synchronize()
    close(from_nhwc(y, Cout), ref64, ref32, name="conv fwd")
    if ey:
        assert (y[..., Cout:] == 5.0).all()


@pytest.mark.parametrize("case", [(2, 8, 8, 32, 64), (2, 16, 16, 64, 32), (1, 32, 32, 256, 256), (2, 8, 8, 32, 4)])
def test_conv3x3_dgrad(H, case):
    nimg, Hh, Ww, Cin, Cout = case
    Cw = 3 if Cout == 4 else Cout                # out_conv: 3 real output channels padded to 4
    x = rnd(nimg, Cin, Hh, Ww, seed=1).double().requires_grad_(True)
    w = rnd(Cw, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    dy = rnd(nimg, Cw, Hh, Ww, seed=3)
    F.conv2d(x, w.double(), padding=1).backward(dy.double())
    x32 = x.detach().float().requires_grad_(True)
    F.conv2d(x32, w, padding=1).backward(dy)
    wd = torch.empty(Cin, 9, Cout, device=DEV)
    H.pack_conv3x3(w.to(DEV), Cw, Cin, wd=wd, Cout_p=Cout)
    dyp = torch.zeros(nimg, Cout, Hh, Ww)
    dyp[:, :Cw] = dy
    dx = torch.empty(nimg, Hh, Ww, Cin, device=DEV)
    H.conv3x3(nhwc(dyp), Cout, wd, None, dx, Cin, nimg, Hh, Ww, Cout, Cin)
    torch.cuda.synchronize()
    close(from_nhwc(dx, Cin), x.grad, x32.grad, name="conv dgrad")


@pytest.mark.parametrize("case", [(2, 8, 8, 32, 64, 32, 64), (4, 16, 16, 64, 32, 64, 32), (4, 32, 32, 192, 192, 192, 192),
                                  (2, 16, 16, 576, 192, 576, 192), (2, 16, 16, 192, 384, 192, 384), (2, 32, 32, 256, 256, 256, 256),
                                  (3, 16, 16, 4, 32, 3, 32), (3, 8, 8, 32, 4, 32, 3), (8, 32, 32, 32, 32, 32, 32),
                                  (64, 32, 32, 128, 128, 128, 128)])    # last: >= 64 Ki pixels -> KT = 16 split-K kernel
def test_conv3x3_wgrad(H, case):
    nimg, Hh, Ww, Cin, Cout, Cin_w, Cout_w = case
    x = torch.zeros(nimg, Cin, Hh, Ww)
    x[:, :Cin_w] = rnd(nimg, Cin_w, Hh, Ww, seed=1)
    dy = torch.zeros(nimg, Cout, Hh, Ww)
    dy[:, :Cout_w] = rnd(nimg, Cout_w, Hh, Ww, seed=2)
    w64 = torch.zeros(Cout_w, Cin_w, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x[:, :Cin_w].double(), w64, padding=1).backward(dy[:, :Cout_w].double())
    w32 = torch.zeros(Cout_w, Cin_w, 3, 3, requires_grad=True)
    F.conv2d(x[:, :Cin_w], w32, padding=1).backward(dy[:, :Cout_w])
    dw = torch.ones(Cout_w, Cin_w, 3, 3, device=DEV)
    db = torch.ones(Cout_w, device=DEV)
    H.conv3x3_wgrad(nhwc(x), Cin, nhwc(dy), Cout, nimg, Hh, Ww, Cin, Cout, dw, Cin_w, Cout_w, accumulate=True, dbias=db, direct=True)
    torch.cuda.synchronize()
    close(dw, w64.grad + 1.0, w32.grad + 1.0, name="conv wgrad")
    close(db, dy[:, :Cout_w].double().sum((0, 2, 3)) + 1.0, dy[:, :Cout_w].sum((0, 2, 3)) + 1.0, name="conv dbias")


@pytest.mark.parametrize("case", [(3, 8, 8, 32, 3), (2, 16, 12, 64, 6), (1, 4, 4, 32, 1)])
def test_thin_output_conv_fwd_bwd(H, case):
    """hid -> 3 convolution as GEMM + 9-tap gather (vd_tap_gather / vd_tap_spread / vd_thin_wgrad_finish) == F.conv2d"""
    nimg, Hh, Ww, Cin, Cout = case
    x, w, bias = rnd(nimg, Cin, Hh, Ww, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=0.2), rnd(Cout, seed=3)
    dy = rnd(nimg, Cout, Hh, Ww, seed=4)
    leaves = [t.double().requires_grad_(True) for t in (x, w, bias)]
    F.conv2d(leaves[0], leaves[1], leaves[2], padding=1).backward(dy.double())
    l32 = [t.clone().requires_grad_(True) for t in (x, w, bias)]
    y32 = F.conv2d(l32[0], l32[1], l32[2], padding=1)
    y32.backward(dy)
    P, nz, cop = nimg * Hh * Ww, (9 * Cout + 3) // 4 * 4, (Cout + 3) // 4 * 4
    wz = torch.zeros(nz, Cin, device=DEV)
    H.pack_conv3x3(w.to(DEV), Cout, Cin, wf=wz, Cin_p=Cin)
    xa = nhwc(x)
    z = torch.empty(P, nz, device=DEV)
    H.gemm(xa, wz, z, P, 9 * Cout, Cin, a_kind=H.ROW, b_kind=H.ROW, lda=Cin, ldb=Cin, ldc=nz)
    y = torch.zeros(nimg, Hh, Ww, cop, device=DEV)
    H.tap_gather(z, nz, bias.to(DEV), y, cop, nimg, Hh, Ww, Cout)
    close(from_nhwc(y, Cout), F.conv2d(x.double(), w.double(), bias.double(), padding=1), y32, name="thin-out fwd")
    assert float(y[..., Cout:].abs().max()) == 0.0 if cop > Cout else True
    dyp = torch.zeros(nimg, cop, Hh, Ww)
    dyp[:, :Cout] = dy
    dz = torch.full((P, nz), 7.0, device=DEV)
    H.tap_spread(nhwc(dyp), cop, dz, nz, nimg, Hh, Ww, Cout)
    gz, cs = torch.empty(nz, Cin, device=DEV), torch.empty(nz, device=DEV)
    H.gemm(dz, xa, gz, nz, Cin, P, a_kind=H.COL, b_kind=H.COL, lda=nz, ldb=Cin, ldc=Cin, splitk=2 if P >= 512 else 1, colsum=cs)
    dw, db = torch.ones(Cout, Cin, 3, 3, device=DEV), torch.ones(Cout, device=DEV)
    H.thin_wgrad_finish(gz, Cout, Cin, Cin, dw, accumulate=True, colsum=cs, cs_stride=9, cs_off=4, dbias=db)
    close(dw, leaves[1].grad + 1, l32[1].grad + 1, name="thin-out dw")
    close(db, leaves[2].grad + 1, l32[2].grad + 1, name="thin-out db")
    dx = torch.empty(nimg, Hh, Ww, Cin, device=DEV)
    H.gemm(dz, wz, dx, P, Cin, nz, a_kind=H.ROW, b_kind=H.COL, lda=nz, ldb=Cin, ldc=Cin)
    close(from_nhwc(dx, Cin), leaves[0].grad, l32[0].grad, name="thin-out dx")


@pytest.mark.parametrize("case", [(3, 8, 8, 3, 32), (2, 16, 12, 4, 64), (1, 4, 4, 1, 32)])
def test_thin_input_conv_fwd_wgrad(H, case):
    """3 -> hid convolution as im2col (vd_im2col3x3) + K = 36 GEMM == F.conv2d, and its weight/bias gradient"""
    nimg, Hh, Ww, Cin, Cout = case
    cip = (Cin + 3) // 4 * 4
    x, w, bias = rnd(nimg, Cin, Hh, Ww, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=0.3), rnd(Cout, seed=3)
    dy = rnd(nimg, Cout, Hh, Ww, seed=4)
    leaves = [t.double().requires_grad_(True) for t in (w, bias)]
    y64 = F.conv2d(x.double(), leaves[0], leaves[1], padding=1)
    y64.backward(dy.double())
    l32 = [t.clone().requires_grad_(True) for t in (w, bias)]
    y32 = F.conv2d(x, l32[0], l32[1], padding=1)
    y32.backward(dy)
    P = nimg * Hh * Ww
    xp = torch.zeros(nimg, cip, Hh, Ww)
    xp[:, :Cin] = x
    xc = torch.full((P, 9 * cip), 5.0, device=DEV)
    H.im2col3x3(nhwc(xp), cip, xc, nimg, Hh, Ww, cip)
    wf = torch.empty(Cout, 9, cip, device=DEV)
    H.pack_conv3x3(w.to(DEV), Cout, Cin, wf=wf, Cin_p=cip)
    y = torch.empty(nimg, Hh, Ww, Cout, device=DEV)
    H.gemm(xc, wf, y, P, Cout, 9 * cip, a_kind=H.ROW, b_kind=H.ROW, lda=9 * cip, ldb=9 * cip, ldc=Cout, bias=bias.to(DEV))
    close(from_nhwc(y, Cout), y64.detach(), y32, name="thin-in fwd")
    gw, db = torch.empty(Cout, 9 * cip, device=DEV), torch.empty(Cout, device=DEV)
    H.gemm(nhwc(dy), xc, gw, Cout, 9 * cip, P, a_kind=H.COL, b_kind=H.COL, lda=Cout, ldb=9 * cip, ldc=9 * cip, colsum=db)
    dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
    H.thin_wgrad_finish(gw, Cout, cip, Cin, dw)
    close(dw, leaves[0].grad, l32[0].grad, name="thin-in dw")
    close(db, leaves[1].grad, l32[1].grad, name="thin-in db")


def test_conv3x3_wgrad_phases_equal_the_single_call(H):
    """vd_conv3x3_wgrad_phase(1) + (2) (what bench.py times separately) == vd_conv3x3_wgrad, bit for bit"""
    nimg, Hh, Ww, Cin, Cout = 4, 16, 16, 64, 32
    x, dy = nhwc(rnd(nimg, Cin, Hh, Ww, seed=1)), nhwc(rnd(nimg, Cout, Hh, Ww, seed=2))
    dw1, db1 = torch.zeros(Cout, Cin, 3, 3, device=DEV), torch.zeros(Cout, device=DEV)
    dw2, db2 = torch.ones(Cout, Cin, 3, 3, device=DEV), torch.ones(Cout, device=DEV)
    H.conv3x3_wgrad(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw1, Cin, Cout, dbias=db1, direct=True)
    H.PROFILE = []
    try:
        H.conv3x3_wgrad(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw2, Cin, Cout, dbias=db2, direct=True)
        names = [r[0] for r in H.PROFILE]
    finally:
        H.PROFILE = None
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)
    assert len(names) == 2 and names[0].startswith("gemm_") and names[1] == "reduce_slabs_oihw_kernel"


# ------------------------------------------------------------------------------------------------ GroupNorm family
def ref_gn_block(x, gamma, beta, film, act, resample, mask=None):
    B, Cc = x.shape[:2]
    h = F.group_norm(x, 32, gamma, beta, 1e-6)
    if film is not None:
        shift, scale = film[:, :Cc, None, None], film[:, Cc:, None, None]
        h = (1 + scale) * h + shift
    if act:
        h = F.silu(h)
    if mask is not None:
        h = h * mask
    if resample == 1:
        h = F.avg_pool2d(h, 2)
    elif resample == 2:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    return h


GN_CASES = [  # nimg, C, H, W, film, act, resample
    (2, 64, 8, 8, False, True, 0), (3, 96, 8, 8, True, True, 0), (2, 256, 16, 16, True, True, 0), (2, 64, 8, 8, False, True, 1),
    (2, 64, 8, 8, False, True, 2), (2, 128, 4, 4, False, False, 0), (1, 1152, 8, 8, True, True, 0), (2, 192, 32, 32, False, True, 1),
    # 64x64 images and wide slabs (round 5): an image slab is shared by sibling workgroups that exchange their partial sums (SPLIT form of
    # gn_bwd_fused_kernel) wherever its pixel rows are whole 128-byte lines -- C = 256 (32-channel slabs, 4 siblings), C = 384 / 768 (the
    # 24-channel slab widened to 96 channels: 16 siblings at 64x64, 4 at 32x32, none at 16x16), unit counts that are not multiples of 8;
    # 96-byte rows (C = 192, 96) and the 64x64 down-sampling norm keep the two-pass form
    (3, 192, 64, 64, True, True, 0), (3, 96, 64, 64, False, True, 0), (2, 384, 64, 64, False, False, 0), (5, 256, 64, 64, True, True, 0),
    (2, 192, 64, 64, False, True, 1), (3, 384, 32, 32, True, True, 0), (2, 768, 16, 16, True, True, 0), (2, 1536, 8, 8, False, True, 0)]


@pytest.mark.parametrize("case", GN_CASES)
def test_gn_forward_backward(H, case):
    nimg, Cc, Hh, Ww, use_film, act, rs = case
    x = rnd(nimg, Cc, Hh, Ww, seed=1) * 1.5 + 0.3
    gamma, beta = 1 + 0.1 * rnd(Cc, seed=2), 0.1 * rnd(Cc, seed=3)
    film = 0.3 * rnd(nimg, 2 * Cc, seed=4) if use_film else None
    Ho, Wo = (Hh // 2, Ww // 2) if rs == 1 else ((Hh * 2, Ww * 2) if rs == 2 else (Hh, Ww))
    dy = rnd(nimg, Cc, Ho, Wo, seed=5)
    add = rnd(nimg, Cc, Hh, Ww, seed=6)

    def run(dt):
        xs, g, b_ = x.to(dt).requires_grad_(True), gamma.to(dt).requires_grad_(True), beta.to(dt).requires_grad_(True)
        fl = None if film is None else film.to(dt).requires_grad_(True)
        out = ref_gn_block(xs, g, b_, fl, act, rs)
        out.backward(dy.to(dt))
        return out.detach(), xs.grad + add.to(dt), g.grad, b_.grad, (None if fl is None else fl.grad)
    o64, dx64, dg64, db64, df64 = run(torch.float64)
    o32, dx32, dg32, db32, df32 = run(torch.float32)

    ldx = Cc + 8
    xd = nhwc(x, ldx)
    stats = torch.empty(nimg, 32, 2, device=DEV)
    H.gn_stats(xd, ldx, nimg, Hh * Ww, Cc, stats)
    coef = torch.empty(nimg, 4, Cc, device=DEV)
    y = torch.empty(nimg, Ho, Wo, Cc, device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    fd = None if film is None else film.to(DEV)
    H.gn_apply(xd, ldx, stats, gd, bd, fd, act, 0.0, 0, rs, y, Cc, nimg, Hh, Ww, Cc, coef)
    torch.cuda.synchronize()
    mean64 = x.double().reshape(nimg, 32, -1).mean(-1)
    close(stats[..., 0], mean64, None, floor=1e-6, name="gn mean")
    close(from_nhwc(y, Cc), o64, o32, name="gn fwd")

    dx = torch.full((nimg, Hh, Ww, Cc), 1.0, device=DEV)
    dfilm = None if film is None else torch.empty(nimg, 2 * Cc, device=DEV)
    dgam, dbet = torch.ones(Cc, device=DEV), torch.ones(Cc, device=DEV)
    H.gn_apply_bwd(nhwc(dy), Cc, xd, ldx, coef, gd, bd, fd, act, 0.0, 0, rs, nhwc(add), Cc, dx, Cc, True, dfilm, dgam, dbet,
                   True, nimg, Hh, Ww, Cc)
    torch.cuda.synchronize()
    # which form ran: siblings per (image, slab) of the SPLIT form, the two-pass form (-1), or the plain single-pass form on a 96-channel slab
    k = H.lib().vd_gn_bwd_last_kernel()
    want = {(256, 4096, 0): 4, (384, 4096, 0): 16, (384, 1024, 0): 4, (192, 4096, 0): -1, (96, 4096, 0): -1, (192, 4096, 1): -1}.get((Cc, Hh * Ww, rs))
    if want is not None:
        assert (k == -1) if want == -1 else (k // 100000000 == want and k % 10000 == 1024), (k, want)
    if (Cc, Hh * Ww) in ((768, 256), (1536, 64)):
        assert 0 < k < 100000000 and k >= 1000000, k            # plain form, non-temporal (whole-line) slab
    close(from_nhwc(dx, Cc), dx64 + 1.0, dx32 + 1.0, slack=6, floor=5e-6, name="gn dx")
    close(dgam, dg64 + 1.0, dg32 + 1.0, slack=6, floor=5e-6, name="dgamma")
    close(dbet, db64 + 1.0, db32 + 1.0, slack=6, floor=5e-6, name="dbeta")
    if film is not None:
        close(dfilm, df64, df32, slack=6, floor=5e-6, name="dfilm")


@pytest.mark.parametrize("case", [(3, 8, 8, 64, 96, 0), (2, 16, 16, 32, 256, 0), (2, 32, 32, 64, 64, 64), (1, 8, 16, 32, 192, 32)])
def test_gn_stats_from_producer_epilogues(H, case):
    """GroupNorm statistics assembled from the partial sums the producing conv / 1x1 GEMM epilogues leave behind, for a
    single source and for the channel concatenation of two sources (the virtual concat of unet.py:315)"""
    nimg, Hh, Ww, Cin, C1, C2 = case
    HW = Hh * Ww
    x = rnd(nimg, Cin, Hh, Ww, seed=1)
    w1, b1 = rnd(C1, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5), rnd(C1, seed=3)
    res = rnd(nimg, C1, Hh, Ww, seed=4)
    y1 = F.conv2d(x.double(), w1.double(), b1.double(), padding=1) + res.double()
    wf = torch.empty(C1, 9, Cin, device=DEV)
    H.pack_conv3x3(w1.to(DEV), C1, Cin, wf=wf, Cin_p=Cin)
    Ct = C1 + C2
    buf = torch.zeros(nimg, Hh, Ww, Ct, device=DEV)
    part1 = torch.full((H.stats_part_numel(nimg, HW, C1),), 7.0, device=DEV)
    H.conv3x3(nhwc(x), Cin, wf, b1.to(DEV), buf[..., :C1], Ct, nimg, Hh, Ww, Cin, C1, res=nhwc(res), ldres=C1, stats_part=part1)
    parts = [(part1, C1, HW // (H.last_row_tile() // 2))]
    ref = y1
    if C2:
        w2, b2 = rnd(C2, Cin, seed=5, scale=Cin ** -0.5), rnd(C2, seed=6)          # second source: a 1x1 conv (plain GEMM)
        y2 = F.conv2d(x.double(), w2.double()[:, :, None, None], b2.double())
        part2 = torch.full((H.stats_part_numel(nimg, HW, C2),), 7.0, device=DEV)
        H.gemm(nhwc(x), w2.to(DEV), buf[0, 0, 0, C1:], nimg * HW, C2, Cin, lda=Cin, ldb=Cin, ldc=Ct, bias=b2.to(DEV), stats=part2,
               stats_hw=HW)
        parts.append((part2, C2, HW // (H.last_row_tile() // 2)))
        ref = torch.cat([y1, y2], 1)
    stats = torch.empty(nimg, 32, 2, device=DEV)
    H.gn_stats_from_partials(parts, nimg, HW, stats)
    direct = torch.empty(nimg, 32, 2, device=DEV)
    H.gn_stats(buf, Ct, nimg, HW, Ct, direct)
    g = ref.reshape(nimg, 32, -1)
    mean, var = g.mean(-1), g.var(-1, unbiased=False)
    close(stats[..., 0], mean, None, floor=2e-6, name="mean")
    close(stats[..., 1], 1 / torch.sqrt(var + 1e-6), None, floor=5e-6, name="rstd")
    close(direct[..., 1], 1 / torch.sqrt(var + 1e-6), None, floor=5e-6, name="rstd direct")
    # partials -> coefficient table in one kernel (vd_gn_coef_from_partials) == statistics + vd_gn_apply's own table
    gamma, beta, film = rnd(Ct, seed=7).to(DEV), rnd(Ct, seed=8).to(DEV), (rnd(nimg, 2 * Ct, seed=9) * 0.1).to(DEV)
    y_a, y_b = torch.empty_like(buf), torch.empty_like(buf)
    coef_a, coef_b = torch.empty(nimg, 4, Ct, device=DEV), torch.full((nimg, 4, Ct), 3.0, device=DEV)
    H.gn_apply(buf, Ct, stats, gamma, beta, film, 1, 0.0, 0, H.RS_NONE, y_a, Ct, nimg, Hh, Ww, Ct, coef_a)
    H.gn_coef_from_partials(parts, nimg, HW, gamma, beta, film, coef_b)
    H.gn_apply(buf, Ct, None, gamma, beta, film, 1, 0.0, 0, H.RS_NONE, y_b, Ct, nimg, Hh, Ww, Ct, coef_b)
    assert torch.equal(coef_a, coef_b) and torch.equal(y_a, y_b)
    # ... and both in ONE launch (vd_gn_apply_from_partials: every apply workgroup finalises the statistics of its own groups; the sums
    # run in fp64 in another order than the wave butterfly, so the fp32 results may differ in the last bit)
    y_c, coef_c = torch.empty_like(buf), torch.full((nimg, 4, Ct), 5.0, device=DEV)
    H.gn_apply_from_partials(buf, Ct, parts, gamma, beta, film, 1, 0.0, 0, H.RS_NONE, y_c, Ct, nimg, Hh, Ww, Ct, coef_c)
    assert (coef_c - coef_a).abs().max().item() <= 2e-6 * coef_a.abs().max().item()
    assert (y_c - y_a).abs().max().item() <= 4e-6 * max(y_a.abs().max().item(), 1.0)
    for rs in (H.RS_DOWN, H.RS_UP):          # the resampling forms of the apply pass, table not requested
        Ho, Wo = (Hh // 2, Ww // 2) if rs == H.RS_DOWN else (Hh * 2, Ww * 2)
        y_d, y_e = torch.empty(nimg, Ho, Wo, Ct, device=DEV), torch.empty(nimg, Ho, Wo, Ct, device=DEV)
        H.gn_apply(buf, Ct, None, gamma, beta, film, 1, 0.0, 0, rs, y_d, Ct, nimg, Hh, Ww, Ct, coef_b)
        H.gn_apply_from_partials(buf, Ct, parts, gamma, beta, film, 1, 0.0, 0, rs, y_e, Ct, nimg, Hh, Ww, Ct, None)
        assert (y_e - y_d).abs().max().item() <= 4e-6 * max(y_d.abs().max().item(), 1.0)


@pytest.mark.parametrize("rs", [0, 1, 2])
def test_plain_resample_and_backward(H, rs):
    nimg, Cc, Hh, Ww = 2, 64, 8, 8
    x = rnd(nimg, Cc, Hh, Ww, seed=1)
    ref = F.avg_pool2d(x, 2) if rs == 1 else (F.interpolate(x, scale_factor=2, mode="nearest") if rs == 2 else x)
    Ho, Wo = ref.shape[2:]
    y = torch.empty(nimg, Ho, Wo, Cc, device=DEV)
    H.gn_apply(nhwc(x), Cc, None, None, None, None, 0, 0.0, 0, rs, y, Cc, nimg, Hh, Ww, Cc, None)
    close(from_nhwc(y, Cc), ref.double(), ref, name="resample")
    dy = rnd(nimg, Cc, Ho, Wo, seed=2)
    xs = x.double().requires_grad_(True)
    r = F.avg_pool2d(xs, 2) if rs == 1 else (F.interpolate(xs, scale_factor=2, mode="nearest") if rs == 2 else xs * 1)
    r.backward(dy.double())
    dx = torch.empty(nimg, Hh, Ww, Cc, device=DEV)
    H.gn_apply_bwd(nhwc(dy), Cc, None, 0, None, None, None, None, 0, 0.0, 0, rs, None, 0, dx, Cc, False, None, None, None, False,
                   nimg, Hh, Ww, Cc)
    close(from_nhwc(dx, Cc), xs.grad, xs.grad.float(), name="resample bwd")


@pytest.mark.parametrize("geom", [(4, 128, 16, 16), (3, 192, 64, 64)])
def test_dropout_mask_is_consistent_and_bernoulli(H, geom):
    """(64x64: the SPLIT single-pass backward regenerates the mask per sibling workgroup from the element index of its pixel range)"""
    (nimg, Cc, Hh, Ww), p = geom, 0.2
    x = rnd(nimg, Cc, Hh, Ww, seed=1) + 3.0            # keep activations away from 0
    gamma, beta = torch.ones(Cc), torch.zeros(Cc)
    xd = nhwc(x)
    stats = torch.empty(nimg, 32, 2, device=DEV)
    H.gn_stats(xd, Cc, nimg, Hh * Ww, Cc, stats)
    coef = torch.empty(nimg, 4, Cc, device=DEV)
    y0, y1, y2 = (torch.empty(nimg, Hh, Ww, Cc, device=DEV) for _ in range(3))
    H.gn_apply(xd, Cc, stats, gamma.to(DEV), beta.to(DEV), None, 0, 0.0, 0, 0, y0, Cc, nimg, Hh, Ww, Cc, coef)
    H.gn_apply(xd, Cc, stats, gamma.to(DEV), beta.to(DEV), None, 0, p, 1234, 0, y1, Cc, nimg, Hh, Ww, Cc, coef)
    H.gn_apply(xd, Cc, stats, gamma.to(DEV), beta.to(DEV), None, 0, p, 1235, 0, y2, Cc, nimg, Hh, Ww, Cc, coef)
    keep = (y1 != 0)
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 0.01, frac
    torch.testing.assert_close(y1[keep], (y0 / (1 - p))[keep], rtol=1e-6, atol=1e-6)
    assert ((y2 != 0) != keep).float().mean().item() > 0.2           # a different seed gives a different mask
    # replay the mask through the torch reference: backward must use the same mask
    mask = from_nhwc(keep.float() / (1 - p), Cc).double()
    xs = x.double().requires_grad_(True)
    dy = rnd(nimg, Cc, Hh, Ww, seed=2)
    (F.group_norm(xs, 32, gamma.double(), beta.double(), 1e-6) * mask).backward(dy.double())
    dx = torch.empty(nimg, Hh, Ww, Cc, device=DEV)
    dg, db = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    H.gn_apply_bwd(nhwc(dy), Cc, xd, Cc, coef, gamma.to(DEV), beta.to(DEV), None, 0, p, 1234, 0, None, 0, dx, Cc, False, None,
                   dg, db, False, nimg, Hh, Ww, Cc)
    close(from_nhwc(dx, Cc), xs.grad, None, floor=2e-5, name="dropout bwd")


# ------------------------------------------------------------------------------------------------ small kernels
def test_colsum_axpby_silu(H):
    x = rnd(1000, 72, seed=1)
    xd = torch.zeros(1000, 80, device=DEV)
    xd[:, :72] = x.to(DEV)
    out = torch.ones(72, device=DEV)
    H.colsum(xd, 80, 1000, 72, out, accumulate=True)
    close(out, x.double().sum(0) + 1, x.sum(0) + 1, name="colsum")
    y = rnd(1000, 72, seed=2)
    yd = y.to(DEV)
    H.axpby(xd, 80, 0.5, yd, 72, 2.0, 1000, 72)
    close(yd, 0.5 * x.double() + 2.0 * y.double(), None, name="axpby")
    v = rnd(5000, seed=3) * 3
    vd, od = v.to(DEV), torch.empty(5000, device=DEV)
    H.silu(vd, od)
    close(od, F.silu(v.double()), F.silu(v), name="silu")
    vs = v.double().requires_grad_(True)
    g = rnd(5000, seed=4)
    F.silu(vs).backward(g.double())
    gd = torch.ones(5000, device=DEV)
    H.silu_bwd(vd, g.to(DEV), gd, accumulate=True)
    close(gd, vs.grad + 1, None, floor=3e-6, name="silu bwd")


@pytest.mark.parametrize("L", [64, 256, 1000, 4096])
def test_softmax_rows(H, L):
    rows = 37
    s = rnd(rows, L, seed=1) * 4
    p64 = torch.softmax(s.double(), -1)
    sd = s.to(DEV)
    H.softmax_rows(sd, rows, L)
    close(sd, p64, torch.softmax(s, -1), name="softmax")
    dp = rnd(rows, L, seed=2)
    ss = s.double().requires_grad_(True)
    torch.softmax(0.25 * ss, -1).backward(dp.double())           # ds = alpha * p*(dp - sum(dp*p))
    pd = torch.softmax(0.25 * s.double(), -1).float().to(DEV)
    dpd = dp.to(DEV)
    H.softmax_rows_bwd(pd, dpd, rows, L, 0.25)
    close(dpd, ss.grad, None, floor=3e-6, name="softmax bwd")


ATTN_CASES = [(2, 2, 256, 64), (1, 1, 1024, 64), (3, 1, 64, 64), (2, 1, 128, 128), (1, 2, 320, 128), (2, 1, 64, 256), (1, 1, 1024, 256)]


def _attn_ref(qkv, B, nh, L, hd, dtype):
    """softmax(q k^T / sqrt(hd)) v per (image, head) on the packed [B, L, 3 nh hd] projection (unet.py:55-64)"""
    x = qkv.to(dtype).reshape(B, L, 3, nh, hd)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))          # [B, nh, L, hd]
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, L, nh * hd)


@pytest.mark.parametrize("B,nh,L,hd", ATTN_CASES)
def test_attn_fused_forward(H, B, nh, L, hd):
    assert H.attn_supported(L, hd, False)
    hid = nh * hd
    qkv = rnd(B, L, 3 * hid, seed=L + hd) * 1.5                            # logits of std ~2: a softmax with real contrast
    ref64, ref32 = _attn_ref(qkv, B, nh, L, hd, torch.float64), _attn_ref(qkv, B, nh, L, hd, torch.float32)
    qd = qkv.to(DEV)
    o = torch.full((B, L, hid + 4), 7.0, device=DEV)                        # ldo > hid: the padding must survive
    lse = torch.empty(B * nh * L, device=DEV)
    flat = qd.reshape(-1)
    H.attn_fwd(flat[0:], flat[hid:], flat[2 * hid:], 3 * hid, o, hid + 4, lse, B, nh, L, hd, 1.0 / math.sqrt(hd))
    torch.cuda.synchronize()
    close(o[..., :hid], ref64, ref32, floor=3e-6, name="attn fwd")
    assert (o[..., hid:] == 7.0).all()
    # log-sum-exp (log2 domain) against the definition
    x = qkv.double().reshape(B, L, 3, nh, hd)
    s = (x[:, :, 0].permute(0, 2, 1, 3) @ x[:, :, 1].permute(0, 2, 3, 1)) / math.sqrt(hd)
    close(lse.reshape(B, nh, L), torch.logsumexp(s, -1) * math.log2(math.e), None, floor=3e-6, name="attn lse")
    # lse = NULL (no backward follows) gives the same output, bit for bit, and so does a second run
    o2 = torch.full((B, L, hid + 4), 7.0, device=DEV)
    H.attn_fwd(flat[0:], flat[hid:], flat[2 * hid:], 3 * hid, o2, hid + 4, None, B, nh, L, hd, 1.0 / math.sqrt(hd))
    assert torch.equal(o, o2)


@pytest.mark.parametrize("B,nh,L,hd", ATTN_CASES + [(2, 1, 1024, 256), (3, 1, 64, 256)])      # (round 3: head dim 256 too)
def test_attn_fused_backward(H, B, nh, L, hd):
    assert H.attn_supported(L, hd, True)
    hid = nh * hd
    qkv = rnd(B, L, 3 * hid, seed=L + hd + 1) * 1.5
    do = rnd(B, L, hid, seed=3)
    g64 = qkv.double().requires_grad_(True)
    _attn_ref(g64, B, nh, L, hd, torch.float64).backward(do.double())
    g32 = qkv.clone().requires_grad_(True)
    _attn_ref(g32, B, nh, L, hd, torch.float32).backward(do)
    qd, dod = qkv.to(DEV), do.to(DEV)
    flat = qd.reshape(-1)
    o = torch.empty(B, L, hid, device=DEV)
    lse, delta = torch.empty(B * nh * L, device=DEV), torch.empty(B * nh * L, device=DEV)
    sc = 1.0 / math.sqrt(hd)
    H.attn_fwd(flat[0:], flat[hid:], flat[2 * hid:], 3 * hid, o, hid, lse, B, nh, L, hd, sc)
    dqkv = torch.full((B, L, 3 * hid), 5.0, device=DEV)
    dflat = dqkv.reshape(-1)
    H.attn_bwd(flat[0:], flat[hid:], flat[2 * hid:], 3 * hid, o, hid, dod, hid, lse, delta, dflat[0:], dflat[hid:], dflat[2 * hid:],
               3 * hid, B, nh, L, hd, sc)
    torch.cuda.synchronize()
    close(dqkv, g64.grad, g32.grad, floor=3e-6, name="attn bwd")
    dqkv2 = torch.empty_like(dqkv)
    d2 = dqkv2.reshape(-1)
    H.attn_bwd(flat[0:], flat[hid:], flat[2 * hid:], 3 * hid, o, hid, dod, hid, lse, delta, d2[0:], d2[hid:], d2[2 * hid:], 3 * hid,
               B, nh, L, hd, sc)
    assert torch.equal(dqkv, dqkv2), "fused attention backward is not bitwise reproducible"


def test_attn_unsupported_shapes_are_refused(H):
    assert not H.attn_supported(100, 64, False) and not H.attn_supported(256, 32, False) and not H.attn_supported(256, 512, True)
    assert H.attn_supported(256, 256, True)             # round 3: the backward kernels are built for head dim 256 as well
    # which shapes TRAINING takes fused is a measured policy (tests/perf_attn.py), inference always does
    assert H.attn_use_fused(1024, 256, 256, False) and not H.attn_use_fused(1024, 256, 128, True)
    assert H.attn_use_fused(4096, 64, 128, True) and not H.attn_use_fused(256, 64, 128, True)
    assert H.attn_use_fused(4096, 256, 128, True)       # 8.6 GB per materialised map: fused whatever the head dim
    x = torch.zeros(1, 100, 192, device=DEV)
    with pytest.raises(H.HipError):
        H.attn_fwd(x, x, x, 192, x, 192, None, 1, 1, 100, 64, 0.125)


def test_layout_roundtrip(H):
    x = rnd(3, 3, 8, 8, seed=1)
    y = torch.full((3, 8, 8, 4), 9.0, device=DEV)
    H.nchw_to_nhwc(x.to(DEV), y, 3, 3, 8, 8, 4)
    assert torch.equal(y[..., :3].cpu(), x.permute(0, 2, 3, 1)) and (y[..., 3] == 0).all()
    z = torch.empty(3, 3, 8, 8, device=DEV)
    H.nhwc_to_nchw(y, 4, z, 3, 3, 8, 8)
    assert torch.equal(z.cpu(), x)


def test_uint8_image_formats(H):
    """the data formats either side of the path: generate.py:149 (export) and datasets.py:115-120 (ingest)"""
    from v_diffusion.functions import to_uint8_images, from_uint8_images
    x = rnd(5, 3, 8, 12, seed=1).clamp(-1.3, 1.3)
    ref = (x * 127.5 + 127.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    got = to_uint8_images(x.to(DEV))
    assert got.dtype == torch.uint8 and torch.equal(got.cpu(), ref)
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (5, 8, 12, 3), generator=g, dtype=torch.uint8)
    flip = torch.tensor([1, 0, 1, 0, 0], dtype=torch.bool)
    t = u8.permute(0, 3, 1, 2).float() / 255.0                       # ToTensor
    t = torch.where(flip[:, None, None, None], t.flip(-1), t)        # RandomHorizontalFlip (applied before ToTensor; commutes)
    ref = (t - 0.5) / 0.5                                            # Normalize(0.5, 0.5)
    got = from_uint8_images(u8.to(DEV), flip.to(DEV))
    assert torch.equal(got.cpu(), ref)
    assert torch.equal(from_uint8_images(u8.to(DEV)).cpu(), (u8.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5)


def test_timestep_embedding_matches_oracle(H):
    from oracle.unet_ref import timestep_embedding
    t = torch.tensor([0.0, 1e-3, 0.25, 0.5, 0.999, 1.0, 0.123456789], dtype=torch.float64)
    for dim in (256, 192, 33):
        out = torch.empty(len(t), dim, device=DEV)
        H.timestep_embedding(t.to(DEV), out, len(t), dim)
        ref = timestep_embedding(t, dim)
        assert (out.cpu() - ref).abs().max().item() <= 2e-7


def test_class_embed_and_multitag(H):
    n, emb, ncls = 9, 64, 10
    y = torch.tensor([0., 1., 10., 3., 3., 0., 7., 10., 5.])
    w, b, te = rnd(emb, ncls, seed=1), rnd(emb, seed=2), rnd(n, emb, seed=3)
    oh = F.one_hot((y.long() - 1).clamp(min=0), ncls).double() * (y != 0).double()[:, None]
    ws = w.double().requires_grad_(True)
    bs = b.double().requires_grad_(True)
    ref = te.double() + F.linear(oh, ws, bs)
    g = rnd(n, emb, seed=4)
    ref.backward(g.double())
    ted = te.to(DEV)
    H.class_embed(y.to(DEV), w.to(DEV), b.to(DEV), ted, n, emb, ncls)
    close(ted, ref.detach(), None, name="class embed")
    dw, db = torch.ones(emb, ncls, device=DEV), torch.ones(emb, device=DEV)
    H.class_embed_bwd(y.to(DEV), g.to(DEV), dw, db, n, emb, ncls, accumulate=True)
    close(dw, ws.grad + 1, None, name="class dw")
    close(db, bs.grad + 1, None, name="class db")
    tags = (rnd(6, 40, seed=5) > 0.8).float()
    tags[0] = 0
    out = torch.empty(6, 40, device=DEV)
    H.multitag_norm(tags.to(DEV), out, 6, 40)
    ref = tags / torch.count_nonzero(tags, dim=1).clamp(min=1.0).sqrt().unsqueeze(1)
    close(out, ref.double(), ref, name="multitag")


# ------------------------------------------------------------------------------------------------ diffusion kernels
@pytest.mark.parametrize("mot,rw", [("v", "snr_trunc"), ("v", "snr_1plus"), ("v", "constant"), ("v", "snr"), ("x0", "snr_trunc"),
                                    ("eps", "snr_trunc"), ("both", "snr_trunc"), ("eps", "snr"), ("x0", "constant")])
def test_qsample_loss_fwd_bwd(H, mot, rw):
    from oracle import diffusion_ref as dref
    n, Cc, R = 5, 3, 8
    Co = 6 if mot == "both" else 3
    x0, eps = rnd(n, Cc, R, R, seed=1).clamp(-1, 1), rnd(n, Cc, R, R, seed=2)
    t = torch.tensor([0.02, 0.3, 0.5, 0.8, 0.97], dtype=torch.float64)
    logsnr = dref.make_schedule("cosine")(t).float()
    out = rnd(n, Co, R, R, seed=3)
    o64 = out.double().requires_grad_(True)
    loss64 = dref.train_loss(lambda a, b, c: o64, lambda tt: logsnr.double(), x0.double(), t, None, eps.double(), mot, rw)
    gl = rnd(n, seed=4)
    (loss64 * gl.double()).sum().backward()
    loss32 = dref.train_loss(lambda a, b, c: out, lambda tt: logsnr, x0, t, None, eps, mot, rw)
    xt = torch.empty(n, Cc, R, R, device=DEV)
    H.q_sample(x0.to(DEV), eps.to(DEV), logsnr.to(DEV), xt, n, Cc, R * R)
    ref_xt = dref.q_sample(x0.double(), logsnr.double()[:, None, None, None], eps.double())
    close(xt, ref_xt, None, floor=1e-6, name="q_sample")
    od = out.to(DEV)
    loss, aux = torch.empty(n, device=DEV), torch.empty(n, 2, device=DEV)
    H.loss_fwd(x0.to(DEV), eps.to(DEV), xt, od, logsnr.to(DEV), H.OUT_TYPES[mot], H.REWEIGHTS[rw], loss, aux, n, Cc, R * R)
    close(loss, loss64.detach(), loss32, floor=1e-5, name="loss")
    dout = torch.full((n, Co, R, R), 5.0, device=DEV)
    H.loss_bwd(x0.to(DEV), eps.to(DEV), xt, od, logsnr.to(DEV), aux, gl.to(DEV), H.OUT_TYPES[mot], H.REWEIGHTS[rw], dout,
               n, Cc, R * R)
    close(dout, o64.grad, None, floor=2e-5, name="dloss")


@pytest.mark.parametrize("mot,clip", [("v", False), ("v", True), ("both", False), ("eps", True), ("x0", False)])
def test_bpd_terms_fwd_bwd(H, mot, clip):
    """vd_bpd_terms / vd_bpd_bwd (KL and discretised decoder NLL, reference diffusion.py:446-464) against an fp64 autograd
    evaluation of the oracle's expressions, at mid-chain variances where the tanh-CDF difference is well conditioned"""
    import math as _m
    from oracle import diffusion_ref as dref
    n, Cc, R = 6, 3, 8
    Co = 6 if mot == "both" else 3
    x0 = ((rnd(n, Cc, R, R, seed=1).clamp(-1, 1) * 127.5).round() / 127.5)
    x0[0, :, :2], x0[1, :, :2] = 1.0, -1.0
    eps = rnd(n, Cc, R, R, seed=2)
    sched = dref.make_schedule("cosine")
    s = torch.tensor([0.3, 0.4, 0.5, 0.6, 0.7, 0.8], dtype=torch.float64)
    ls, lt = sched(s).float().reshape(-1, 1, 1, 1), sched(s + 0.125).float().reshape(-1, 1, 1, 1)
    xt = dref.q_sample(x0, lt, eps)
    out = rnd(n, Co, R, R, seed=3) * 0.7

    def terms(o, dt):
        c1, c2, tlv = dref.ddpm_coefs(ls, lt, "fixed_small")
        _, _, lv = dref.ddpm_coefs(ls, lt, "fixed_medium", 0.3)
        c1, c2, tlv, lv = (v.to(dt) for v in (c1, c2, tlv, lv))
        px0 = dref.predictions(mot, xt.to(dt), o, lt.to(dt))[0]
        if clip:
            px0 = px0.clamp(-1.0, 1.0)
        kl = dref.normal_kl(c1 * xt.to(dt) + c2 * x0.to(dt), tlv, c1 * xt.to(dt) + c2 * px0, lv).flatten(1).mean(1) / _m.log(2.0)
        xc, inv = x0.to(dt) - px0, torch.exp(-0.5 * lv)
        cdf = lambda z: 0.5 * (1.0 + torch.tanh(_m.sqrt(2.0 / _m.pi) * (z + 0.044715 * z ** 3)))
        cu = torch.where(x0 > 0.999, torch.ones((), dtype=dt), cdf(inv * (xc + 1.0 / 255)))
        cl = torch.where(x0 < -0.999, torch.zeros((), dtype=dt), cdf(inv * (xc - 1.0 / 255)))
        nll = (-torch.log(torch.clamp(cu - cl - 1e-12, min=0) + 1e-12)).flatten(1).mean(1) / _m.log(2.0)
        return kl, nll, px0

    o64 = out.double().requires_grad_(True)
    kl64, nll64, p64 = terms(o64, torch.float64)
    o32 = out.clone().requires_grad_(True)
    kl32, nll32, _ = terms(o32, torch.float32)
    use_kl = torch.tensor([1.0, 0.0, 1.0, 0.0, 1.0, 1.0])
    gl = rnd(n, seed=4)
    (torch.where(use_kl != 0, kl64, nll64) * gl.double()).sum().backward()
    (torch.where(use_kl != 0, kl32, nll32) * gl).sum().backward()
    kl32, nll32 = kl32.detach(), nll32.detach()
    # coefficient table exactly as GaussianDiffusion._bpd_coefs builds it
    import v_diffusion
    gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine"), 8, mot, "fixed_medium", "snr_trunc", "kl", intp_frac=0.3)
    coef = gd._bpd_coefs(ls.to(DEV), lt.to(DEV))
    kl, nll, mse = (torch.empty(n, device=DEV) for _ in range(3))
    pred = torch.empty(n, Cc, R, R, device=DEV)
    H.bpd_terms(x0.to(DEV), xt.to(DEV), out.to(DEV), coef, H.OUT_TYPES[mot], clip, kl, nll, pred, mse, n, Cc, R * R)
    close(kl, kl64.detach(), kl32, floor=2e-5, name="kl")
    close(nll, nll64.detach(), nll32, floor=1e-4, name="nll")          # log of a difference of fp32 CDFs: ~4 digits
    close(pred, p64.detach(), None, floor=2e-6, name="pred")
    close(mse, ((p64.detach() - x0.double()) ** 2).flatten(1).mean(1), None, floor=1e-5, name="mse")
    dout = torch.full((n, Co, R, R), 5.0, device=DEV)
    H.bpd_bwd(x0.to(DEV), xt.to(DEV), out.to(DEV), coef, use_kl.to(DEV), gl.to(DEV), H.OUT_TYPES[mot], clip, dout, n, Cc, R * R)
    rows = use_kl != 0
    close(dout[rows.to(DEV)], o64.grad[rows], None, floor=2e-5, name="d kl")
    # decoder-NLL rows: where a pixel's two fp32 tanh-CDFs saturate to the same number the clamp kills the gradient (in the
    # reference's fp32 autograd too, not in fp64) -- compare with the fp32 autograd of the same expression, in L2
    got, ref = dout[(~rows).to(DEV)].cpu().double(), o32.grad[~rows].double()
    assert (got - ref).norm().item() <= 3e-2 * ref.norm().item(), "d nll"


def test_sumsq_adamw_ema(H):
    n = 100003
    p, g = rnd(n, seed=1), rnd(n, seed=2) * 0.1
    pd, gd = torch.zeros(n + 1, device=DEV)[:n], g.to(DEV)
    pd.copy_(p)
    ss = torch.empty(1, device=DEV)
    H.sumsq(gd, ss)
    close(ss, (g.double() ** 2).sum().reshape(1), None, floor=1e-6, name="sumsq")
    ref = torch.nn.Parameter(p.clone().double())
    ref.grad = g.clone().double()
    opt = torch.optim.AdamW([ref], lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ema_ref = p.clone().double()
    m, v, ema = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), p.clone().to(DEV)
    for step in range(1, 4):
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        ema_ref += (1 - 0.99) * (ref.detach() - ema_ref)
        ref.grad = g.clone().double()
        H.sumsq(gd, ss)
        H.adamw_ema(pd, gd, m, v, ema, ss, 1.0, 2e-4, 0.9, 0.999, 1e-8, 0.01, 1 - 0.9 ** step, 1 - 0.999 ** step, 0.99)
    close(pd, ref.detach(), None, floor=2e-6, name="adamw p")
    close(ema, ema_ref, None, floor=2e-6, name="ema")


@pytest.mark.parametrize("knobs", [{"VD_GEMM_KT": "16", "VD_GEMM_TILE": "128"}, {"VD_GEMM_LEGACY": "1"}, {"VD_GEMM_TR": "0"},
                                   {"VD_GEMM_KT": "16", "VD_GEMM_TILE": "128", "VD_GEMM_TR": "0"},
                                   {"VD_GEMM_SPLIT": "0", "VD_GEMM_KT": "16", "VD_GEMM_TILE": "128"}])
def test_kernel_variants_in_subprocess(H, knobs):
    """Instantiations the plain run does not select: the KT = 16 form of the 128x128 LDS-DMA kernel (picked for launches
    of >= 1024 workgroups), the register-staged fallback kernel (picked when a leading dimension exceeds the 32-bit
    buffer-offset range), and the column-per-lane epilogue for launches that the plain run sends to the transposed-
    accumulator form (VD_GEMM_TR=0), and -- since the split-operand forms became the default in round 5 -- the fp32-MFMA forms of the
    128-row tiles (VD_GEMM_SPLIT=0).  Force each through its knob and re-run the GEMM / conv parity cases in a child
    process (the knobs are read once per process)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, **knobs)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_kernels_gpu.py"), "-q", "-x", "--no-header",
                        "-p", "no:cacheprovider", "-k",
                        # (the fp32-MFMA row: the operand kinds that HAVE split forms -- im2col addressing never takes them)
                        "gemm_kinds or batched_heads or test_gemm_splitk or gemm_grouped" if knobs.get("VD_GEMM_SPLIT") == "0" else
                        "gemm_kinds or conv3x3_forward or conv3x3_dgrad or conv3x3_wgrad or batched_heads or test_gemm_splitk or colsum or gn_stats_from_producer_epilogues"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("wide", ["1", "0"])
def test_wino_forms_in_subprocess(H, wide):
    """The launcher picks between the 64-tile and the 128-tile ("wide") form of the Winograd convolution by how the items fill
    the residency rounds, so a plain run sends the small parity cases to the 64-tile form and most benchmark shapes to the wide
    one.  VD_WINO_WIDE forces one form for every geometry it serves; both selected tests call vd_conv3x3_wino and assert the
    instantiation through vd_wino_last_kernel (tests/test_bench_shapes_gpu.py::_expected_wino mirrors the forced choice):
      1: the ragged / multi-image / partial-channel-block cases and 576->576 @16x16 at B = 128 through the wide kernel;
      0: the benchmark launches (B = 128: 32x32x256, 64x64x192, ...) through the 64-tile kernel's multi-round item loop."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, VD_WINO_WIDE=wide)
    here = os.path.dirname(os.path.abspath(__file__))
    files = [os.path.join(here, "test_kernels_gpu.py"), os.path.join(here, "test_bench_shapes_gpu.py")]
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-q", "-x", "--no-header", "-p", "no:cacheprovider", "-k",
                        "conv3x3_wino_forward_stats_dgrad or wino_conv_at_bench_launches"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "27 passed" in r.stdout, r.stdout[-500:]      # 12 geometry cases + 7 + 8 bench launches (WINO_BENCH + CELEBA_MIXED)


SPLIT_CHILD = r"""
import math, os, sys, json
sys.path[:0] = [%r]
import torch
from v_diffusion import _hip as H
DEV = "cuda"
out = {}
g = torch.Generator(DEV).manual_seed(5)
def rel(a, b): return ((a.double() - b).norm() / b.norm()).item()
# every operand kind of the 128-row tiles, K tiles of 16 and 32, plain and transposed epilogues, batched heads, split-K, grouped
for name, (ak, bk, M, N, K, kw) in {
    "RR": (0, 0, 4096, 256, 512, {}), "RC": (0, 1, 4096, 512, 256, {}), "CC": (1, 1, 1024, 256, 1024, {}), "CR": (1, 0, 256, 384, 640, {}),
    "RR_small": (0, 0, 1024, 256, 256, {}), "RR_ragged": (0, 0, 1000, 260, 100, {}), "CC_splitk": (1, 1, 256, 256, 8192, {"splitk": 6}),
    "RR_big": (0, 0, 131072, 256, 256, {}), "RC_big": (0, 1, 65536, 512, 256, {}),
}.items():
    A = torch.randn((M, K), device=DEV, generator=g); B = torch.randn((N, K), device=DEV, generator=g)
    bias = torch.randn((N,), device=DEV, generator=g); R = torch.randn((M, N), device=DEV, generator=g)
    Ad = A if ak == 0 else A.T.contiguous(); Bd = B if bk == 0 else B.T.contiguous()
    C = torch.empty((M, N), device=DEV)
    extra = {} if kw else {"bias": bias, "R": R, "ldr": N}
    H.gemm(Ad, Bd, C, M, N, K, a_kind=ak, b_kind=bk, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, tile=128, **extra, **kw)
    torch.cuda.synchronize()
    ref = A.double() @ B.double().T + (0 if kw else bias.double() + R.double())
    out[name] = (H.lib().vd_gemm_last_tile(), rel(C, ref), (C.double() - ref).abs().max().item() / ref.abs().max().item())
# positive operands: every product has the same sign, a truncating accumulator would show as a bias
A = torch.rand((2048, 2048), device=DEV, generator=g) + 0.5; B = torch.rand((256, 2048), device=DEV, generator=g) + 0.5
C = torch.empty((2048, 256), device=DEV)
H.gemm(A, B, C, 2048, 256, 2048, a_kind=0, b_kind=0, lda=2048, ldb=2048, ldc=256, tile=128)
ref = A.double() @ B.double().T
out["positive"] = (H.lib().vd_gemm_last_tile(), rel(C, ref), ((C.double() - ref) / ref).mean().item())
# operands spread over 2^+-20: the three pieces of a value keep its own exponent (no shared scale)
A = torch.randn((1024, 512), device=DEV, generator=g) * torch.exp2(torch.randint(-20, 21, (1024, 512), device=DEV, generator=g).float())
B = torch.randn((256, 512), device=DEV, generator=g) * torch.exp2(torch.randint(-20, 21, (256, 512), device=DEV, generator=g).float())
C = torch.empty((1024, 256), device=DEV)
H.gemm(A, B, C, 1024, 256, 512, a_kind=0, b_kind=0, lda=512, ldb=512, ldc=256, tile=128)
ref = A.double() @ B.double().T
out["wide_range"] = (H.lib().vd_gemm_last_tile(), rel(C, ref), 0.0)
ents, refs = [], []
for e in range(9):
    dy = torch.randn((8192, 256), device=DEV, generator=g); x = torch.randn((8192, 256), device=DEV, generator=g)
    ents.append((dy, x, torch.empty((256, 256), device=DEV), torch.empty((256,), device=DEV)))
    refs.append(dy.double().T @ x.double())
H.gemm_grouped_wgrad(ents, 256, 256, 8192, 256, 256, 256, 7)
torch.cuda.synchronize()
out["grouped"] = (H.lib().vd_gemm_last_tile(), max(rel(e[2], r) for e, r in zip(ents, refs)), 0.0)
print("RESULT " + json.dumps(out))
"""


def test_split_operand_gemm_forms_in_subprocess(H):
    """VD_GEMM_SPLIT=1 runs the 128-row GEMM tiles on the 16-bit matrix cores with every fp32 operand split exactly into three bf16
    pieces and the six products that reach 2^-24 of a.b kept (csrc/gemm.hip, SPL).  The same child script runs with the switch off and on:
    the split forms must actually be the ones that ran (instantiation code), and against fp64 their error must not exceed the fp32 MFMA
    chain's by more than 25 % on any case (measured: 0.6-0.9 of it on normal operands, 1.12 with magnitudes spread over 2^+-20, where the
    dropped 2^-24 terms show), and the mean signed error on all-positive operands must stay under 1e-7 of the result (measured -4e-8)."""
    import json
    import os
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "v-diffusion-torch_amd")
    res = {}
    for flag in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", SPLIT_CHILD % pkg], env=dict(os.environ, VD_GEMM_SPLIT=flag), capture_output=True, text=True,
                           timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
        res[flag] = json.loads(line[0][7:])
    for name, (code1, err1, aux1) in res["1"].items():
        code0, err0, aux0 = res["0"][name]
        assert (code1 // 10 ** 6) // 100 >= 2 and (code0 // 10 ** 6) // 100 < 2, f"{name}: instantiation codes {code0} / {code1}"
        assert err1 <= 1.25 * err0 + 1e-9, f"{name}: split-operand rel-L2 {err1:.3e} vs fp32 MFMA {err0:.3e}"
    # all-positive operands: the 16-bit pipe's adder leaves a systematic error of -4e-8 of the result (fp32 MFMA chain: < 1e-9) -- under one
    # fp32 ulp (6e-8), bounded here at 1e-7
    assert abs(res["1"]["positive"][2]) <= 1e-7, res["1"]["positive"]
    print("split-operand / fp32 MFMA rel-L2 against fp64: " + ", ".join(f"{k} {res['1'][k][1]:.2e}/{res['0'][k][1]:.2e}" for k in res["1"]))


SPLIT_DOMAIN_CHILD = r"""
import sys, json, torch
sys.path.insert(0, %r)
from v_diffusion import _hip as H
M, N, K = 32768, 256, 512         # (a shape the launcher gives 128-row tiles: the split forms exist for those)
g = torch.Generator("cuda").manual_seed(3)
A = torch.randn((M, K), device="cuda", generator=g)
B = torch.randn((N, K), device="cuda", generator=g) * 1e-3
A[5, 7] = 3.0e38          # finite in fp32 and in bf16 (largest bf16: 3.39e38): inside the documented domain
A[9, 1] = 3.4e38          # finite in fp32, rounds to Inf in bf16: outside
A[13, 2] = float("inf")
C = torch.empty((M, N), device="cuda")
H.gemm(A, B, C, M, N, K, a_kind=0, b_kind=0, lda=K, ldb=K, ldc=N)
torch.cuda.synchronize()
ref = A.double() @ B.double().T
rows = [r for r in range(M) if r not in (9, 13)]
rel = float(((C[rows].double() - ref[rows]).norm() / ref[rows].norm()))
print("RESULT " + json.dumps(dict(code=H.lib().vd_gemm_last_tile(), rel=rel, finite5=bool(torch.isfinite(C[5]).all()),
                                  nan9=int(torch.isnan(C[9]).sum()), inf9=int(torch.isinf(C[9]).sum()),
                                  nan13=int(torch.isnan(C[13]).sum()), inf13=int(torch.isinf(C[13]).sum()))))
"""


def test_split_operand_gemm_domain(H):
    """the stated domain of the opt-in split-operand GEMM forms (round-4 advice; include/vdiff_hip.h, csrc/gemm.hip): operands up to the
    largest bf16 (3.39e38) behave as in the fp32 form; a finite fp32 value beyond it, or an Inf, turns its output row into NaN under
    VD_GEMM_SPLIT=1 where the fp32 MFMA form gives finite values / Inf -- documented, not hidden; all other rows are unaffected."""
    import json
    import os
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "v-diffusion-torch_amd")
    res = {}
    for flag in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", SPLIT_DOMAIN_CHILD % pkg], env=dict(os.environ, VD_GEMM_SPLIT=flag), capture_output=True,
                           text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
        res[flag] = json.loads(line[0][7:])
    f32, spl = res["0"], res["1"]
    assert (spl["code"] // 10 ** 6) // 100 >= 2 and (f32["code"] // 10 ** 6) // 100 < 2, (f32["code"], spl["code"])
    for r in (f32, spl):
        assert r["finite5"] and r["rel"] <= 2e-6, r                    # inside the domain: both forms, every ordinary row
    assert f32["nan9"] == 0 and f32["inf13"] > 0                       # fp32 MFMA: a huge finite operand stays finite / Inf propagates as Inf
    assert spl["nan9"] > 0 and spl["nan13"] > 0, spl                   # split forms: NaN (the documented difference)
    print("split-operand domain:", json.dumps(res))


# ------------------------------------------------------------------------------------------------ Winograd F(2x2,3x3) convolution
WINO_CASES = [  # nimg, H, W, Cin, Cout, ldx_extra, ldy_extra      (geometries: every patch-image variant of csrc/wino.hip)
    (2, 32, 32, 64, 64, 0, 0),       # 16 tiles per row: one workgroup = 4 tile rows of one image
    (3, 16, 16, 32, 96, 32, 64),     # 8 tiles per row, one image per workgroup, ld > C on both sides, partial 2nd channel block
    (5, 8, 8, 48, 36, 0, 4),         # 4 tiles per row: 4 images per workgroup, last workgroup has 1 valid image, Cout % 32 != 0
    (1, 64, 64, 16, 32, 0, 0),       # 32 tiles per row: half a tile row per wave
    (2, 16, 32, 32, 32, 0, 0),       # non-square
    (1, 8, 128, 16, 20, 0, 0),       # 64 tiles per row (largest patch image)
    (4, 32, 32, 256, 256, 0, 0), (8, 16, 16, 512, 256, 0, 0), (16, 8, 8, 256, 256, 0, 0), (1, 64, 64, 192, 192, 0, 0),
    (10, 32, 32, 32, 96, 0, 0),      # 40 tile groups x 3 channel blocks = 120 work items (one ragged round of persistent workgroups)
    (9, 32, 32, 16, 288, 0, 0)]      # 36 x 9 = 324 items: one full round of 256 (XCD-permuted ids) + a ragged one (plain ids)


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv3x3_wino_forward_stats_dgrad(H, case):
    """vd_conv3x3_wino == F.conv2d (+bias +residual), its GroupNorm partial sums == statistics of the output, and the same
    kernel on the rotated pack == the input gradient.  Error budget: Winograd F(2x2,3x3) in fp32 carries ~2x the rounding error
    of a direct fp32 sum (measured against fp64), so the slack over torch's own fp32 result is 8 instead of 4."""
    nimg, Hh, Ww, Cin, Cout, ex, ey = case
    assert H.lib().vd_conv3x3_wino_supported(nimg, Hh, Ww, Cin, Cout, Cin + ex, Cout + ey, Cout + ey) == 1
    x, w, b = rnd(nimg, Cin, Hh, Ww, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=3)
    res = rnd(nimg, Cout, Hh, Ww, seed=4)
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    ref32 = F.conv2d(x, w, b, padding=1) + res
    xd, rd = nhwc(x, Cin + ex), nhwc(res, Cout + ey)
    uf, ud = torch.empty(16, Cout, Cin, device=DEV), torch.empty(16, Cin, Cout, device=DEV)
    H.wino_pack(w.to(DEV), Cout, Cin, uf=uf, ud=ud)
    y = torch.full((nimg, Hh, Ww, Cout + ey), 5.0, device=DEV)
    HW = Hh * Ww
    part = torch.full((H.stats_part_numel(nimg, HW, Cout),), 7.0, device=DEV)
    H.conv3x3_wino(xd, Cin + ex, uf, b.to(DEV), y, Cout + ey, nimg, Hh, Ww, Cin, Cout, res=rd, ldres=Cout + ey, stats_part=part)
    torch.cuda.synchronize()
    close(from_nhwc(y, Cout), ref64, ref32, slack=8.0, floor=4e-6, name="wino fwd")
    if ey:
        assert (y[..., Cout:] == 5.0).all(), "padding channels of the output were written"
    assert H.last_row_tile() == 128
    if Cout % 32 == 0:
        stats = torch.empty(nimg, 32, 2, device=DEV)
        H.gn_stats_from_partials([(part, Cout, HW // 64)], nimg, HW, stats)
        g = ref64.reshape(nimg, 32, -1)
        close(stats[..., 0], g.mean(-1), None, floor=4e-6, name="wino stats mean")
        close(stats[..., 1], 1 / torch.sqrt(g.var(-1, unbiased=False) + 1e-6), None, floor=1e-5, name="wino stats rstd")
    # without statistics / residual / bias: bitwise the same convolution result
    y2 = torch.empty(nimg, Hh, Ww, Cout, device=DEV)
    H.conv3x3_wino(xd, Cin + ex, uf, None, y2, Cout, nimg, Hh, Ww, Cin, Cout)
    y3 = torch.empty_like(y2)
    H.conv3x3_wino(xd, Cin + ex, uf, None, y3, Cout, nimg, Hh, Ww, Cin, Cout)
    assert torch.equal(y2, y3)
    close(from_nhwc(y2, Cout), F.conv2d(x.double(), w.double(), padding=1), F.conv2d(x, w, padding=1), slack=8.0, floor=4e-6, name="wino plain")
    # input gradient: the same kernel on ud with the channel roles swapped
    if Cout % 16 == 0 and Cin % 4 == 0:
        dy = rnd(nimg, Cout, Hh, Ww, seed=5)
        x64 = x.double().requires_grad_(True)
        F.conv2d(x64, w.double(), padding=1).backward(dy.double())
        x32 = x.clone().requires_grad_(True)
        F.conv2d(x32, w, padding=1).backward(dy)
        dx = torch.empty(nimg, Hh, Ww, Cin, device=DEV)
        H.conv3x3_wino(nhwc(dy), Cout, ud, None, dx, Cin, nimg, Hh, Ww, Cout, Cin)
        torch.cuda.synchronize()
        close(from_nhwc(dx, Cin), x64.grad, x32.grad, slack=8.0, floor=4e-6, name="wino dgrad")


def test_conv3x3_wino_rejects_unsupported(H):
    f = H.lib().vd_conv3x3_wino_supported
    assert f(2, 4, 4, 32, 32, 32, 32, 0) == 0            # 2 tiles per row
    assert f(2, 16, 16, 24, 32, 24, 32, 0) == 0          # Cin % 16
    assert f(2, 12, 16, 32, 32, 32, 32, 0) == 0          # H/2 not a power of two
    assert f(2, 16, 16, 32, 32, 32, 32, 0) == 1
    x = torch.zeros(2, 4, 4, 32, device=DEV)
    with pytest.raises(H.HipError, match="unsupported geometry"):
        H.conv3x3_wino(x, 32, torch.zeros(16, 32, 32, device=DEV), None, torch.empty_like(x), 32, 2, 4, 4, 32, 32)


WINO_WGRAD_CASES = [  # nimg, H, W, Cin, Cout, Cin_w, Cout_w, ldx_extra, lddy_extra
    (2, 32, 32, 64, 64, 64, 64, 0, 0),        # 16 tiles per row: one stage = one tile-row segment, 2 stages per row? (TW = 16: one)
    (3, 16, 16, 32, 96, 32, 96, 32, 64),      # 8 tiles per row (stage = 2 tile rows), ld > C, partial 64-blocks
    (5, 8, 8, 48, 36, 48, 36, 0, 4),          # 4 tiles per row (stage = one image)
    (1, 64, 64, 16, 32, 16, 32, 0, 0),        # 32 tiles per row: two stages per tile row (interior left / right borders)
    (2, 16, 32, 32, 32, 32, 32, 0, 0),        # non-square
    (4, 32, 32, 4, 32, 3, 32, 0, 0), (4, 32, 32, 32, 4, 32, 3, 0, 0),     # padded thin sides
    (8, 32, 32, 256, 256, 256, 256, 0, 0), (8, 16, 16, 512, 256, 512, 256, 0, 0), (32, 8, 8, 256, 256, 256, 256, 0, 0),
    (2, 64, 64, 192, 192, 192, 192, 0, 0)]


@pytest.mark.parametrize("case", WINO_WGRAD_CASES)
def test_conv3x3_wgrad_wino(H, case):
    """vd_conv3x3_wgrad_wino == autograd of F.conv2d with respect to the kernel and the bias (fp64 truth, torch fp32 yardstick;
    the Winograd form carries ~2x the rounding error of a direct fp32 sum: slack 8), bitwise reproducible, accumulate mode."""
    nimg, Hh, Ww, Cin, Cout, Cin_w, Cout_w, ex, ey = case
    assert H.lib().vd_conv3x3_wgrad_wino_supported(nimg, Hh, Ww, Cin, Cout, Cin + ex, Cout + ey) == 1
    x = torch.zeros(nimg, Cin, Hh, Ww)
    x[:, :Cin_w] = rnd(nimg, Cin_w, Hh, Ww, seed=1)
    dy = torch.zeros(nimg, Cout, Hh, Ww)
    dy[:, :Cout_w] = rnd(nimg, Cout_w, Hh, Ww, seed=2)
    w64 = torch.zeros(Cout_w, Cin_w, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x[:, :Cin_w].double(), w64, padding=1).backward(dy[:, :Cout_w].double())
    w32 = torch.zeros(Cout_w, Cin_w, 3, 3, requires_grad=True)
    F.conv2d(x[:, :Cin_w], w32, padding=1).backward(dy[:, :Cout_w])
    dw = torch.ones(Cout_w, Cin_w, 3, 3, device=DEV)
    db = torch.ones(Cout_w, device=DEV)
    xd, dyd = nhwc(x, Cin + ex), nhwc(dy, Cout + ey)
    H.conv3x3_wgrad_wino(xd, Cin + ex, dyd, Cout + ey, nimg, Hh, Ww, Cin, Cout, dw, Cin_w, Cout_w, accumulate=True, dbias=db)
    torch.cuda.synchronize()
    close(dw, w64.grad + 1.0, w32.grad + 1.0, slack=8.0, floor=4e-6, name="wino wgrad")
    close(db, dy[:, :Cout_w].double().sum((0, 2, 3)) + 1.0, dy[:, :Cout_w].sum((0, 2, 3)) + 1.0, slack=8.0, floor=4e-6, name="wino dbias")
    a, b = torch.empty_like(dw), torch.empty_like(dw)
    H.conv3x3_wgrad_wino(xd, Cin + ex, dyd, Cout + ey, nimg, Hh, Ww, Cin, Cout, a, Cin_w, Cout_w)
    H.conv3x3_wgrad_wino(xd, Cin + ex, dyd, Cout + ey, nimg, Hh, Ww, Cin, Cout, b, Cin_w, Cout_w)
    assert torch.equal(a, b), "not bitwise reproducible"
    close(a, w64.grad, w32.grad, slack=8.0, floor=4e-6, name="wino wgrad (no accumulate, no bias)")


# ------------------------------------------------------------------------------------------------ Winograd F(4x4,3x3) input gradient
WINO43_CASES = [  # nimg, H, W, Cin, Cout, lddy_extra, lddx_extra
    (2, 32, 32, 32, 8, 0, 0),          # one K tile, one channel block
    (3, 32, 32, 96, 24, 8, 32),        # 3 images x 3 channel blocks = 9 items (plain ids), ld > C on both sides
    (8, 32, 32, 128, 64, 0, 0),        # 8 tile groups x 4 channel blocks: the XCD-aware 4 x 8 item order
    (1, 16, 64, 32, 16, 0, 0),         # 64-wide image, one 16-row part
    (2, 64, 64, 64, 40, 0, 0),         # 64x64: 4 parts per image (interior parts have halo rows on both sides)
    (40, 32, 32, 256, 256, 0, 0),      # 40 x 8 = 320 items: one full persistent round (permuted ids) + a ragged one
    (4, 16, 16, 32, 8, 0, 0),          # 16x16 images: four per item (shared zero rows between them), one item
    (12, 16, 16, 96, 40, 8, 32),       # 3 image quads x 3 channel blocks, ld > C on both sides
    (128, 16, 16, 256, 256, 0, 0),     # the CIFAR 16x16 level at batch 128: 32 quads x 8 channel blocks = one full round (4 x 8 item order)
]


@pytest.mark.parametrize("case", WINO43_CASES)
def test_conv3x3_dgrad_wino43(H, case):
    """vd_conv3x3_dgrad_wino43 == autograd of F.conv2d with respect to its input (fp64 truth).  F(4x4,3x3) in fp32 carries about
    7x the rounding error of a direct fp32 sum (measured: relative L2 3-4e-6, max 1.5e-5 of the largest element at K = 256), which
    is why it serves gradients only (stated bound on gradients: relative L2 <= 1e-4): bounds here 1.5e-5 relative L2 and 6e-5 of
    the largest element.  Bitwise reproducible; padding channels of dx untouched."""
    nimg, Hh, Ww, Cin, Cout, ey, ex = case
    assert H.lib().vd_conv3x3_dgrad_wino43_supported(nimg, Hh, Ww, Cin, Cout, Cout + ey, Cin + ex) == 1
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    dy = rnd(nimg, Cout, Hh, Ww, seed=5)
    x64 = torch.zeros(nimg, Cin, Hh, Ww, dtype=torch.float64, requires_grad=True)
    F.conv2d(x64, w.double(), padding=1).backward(dy.double())
    u43 = torch.empty(H.lib().vd_wino43_u_floats(Cout, Cin), device=DEV)
    H.wino43_pack(w.to(DEV), Cout, Cin, u43)
    dx = torch.full((nimg, Hh, Ww, Cin + ex), 5.0, device=DEV)
    H.conv3x3_dgrad_wino43(nhwc(dy, Cout + ey), Cout + ey, u43, dx, Cin + ex, nimg, Hh, Ww, Cin, Cout)
    assert H.lib().vd_wino43_last_kernel() == Ww // 4
    dx2 = torch.full_like(dx, 5.0)
    H.conv3x3_dgrad_wino43(nhwc(dy, Cout + ey), Cout + ey, u43, dx2, Cin + ex, nimg, Hh, Ww, Cin, Cout)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2), "not bitwise reproducible"
    if ex:
        assert (dx[..., Cin:] == 5.0).all(), "padding channels of dx were written"
    got, ref = from_nhwc(dx, Cin).double().cpu(), x64.grad
    rel = ((got - ref).norm() / ref.norm()).item()
    err = (got - ref).abs().max().item()
    assert rel <= 1.5e-5 and err <= 6e-5 * ref.abs().max().item(), f"wino43 dgrad rel-L2 {rel:.3e}, max err {err:.3e} of {ref.abs().max().item():.2f}"


# forward pass through the same kernel (FWD = true: dyadic interpolation points, bias / residual / GroupNorm partials in the epilogue).
# Cases as above with the roles of the channel counts swapped: GEMM K = Cin (multiple of 8), columns = Cout (multiple of 32)
@pytest.mark.parametrize("case", WINO43_CASES)
def test_conv3x3_wino43_fwd(H, case):
    """vd_conv3x3_wino43_fwd == F.conv2d + bias + residual (fp64 truth), GroupNorm partials == statistics of what was written,
    bitwise reproducible, padding channels of y untouched, ld > C on x / y / res.  Per-layer bound: the F(2x2,3x3) forward tests'
    slack (8 x torch's own fp32-vs-fp64 error of the same layer, floor 4e-6)."""
    nimg, Hh, Ww, Cout, Cin, ex, ey = case                     # (the table's Cin % 32 column is this kernel's Cout)
    HW = Hh * Ww
    assert H.lib().vd_conv3x3_wino43_fwd_supported(nimg, Hh, Ww, Cin, Cout, Cin + ex, Cout + ey, Cout + ey) == 1
    x = rnd(nimg, Cin, Hh, Ww, seed=1)
    w, b = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=3)
    res = rnd(nimg, Cout, Hh, Ww, seed=4)
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    ref32 = F.conv2d(x, w, b, padding=1) + res
    u43f = torch.empty(H.lib().vd_wino43_u_floats(Cout, Cin), device=DEV)
    H.wino43_pack_fwd(w.to(DEV), Cout, Cin, u43f)
    y = torch.full((nimg, Hh, Ww, Cout + ey), 5.0, device=DEV)
    part = torch.full((H.stats_part_numel(nimg, HW, Cout),), 7.0, device=DEV)
    H.conv3x3_wino43_fwd(nhwc(x, Cin + ex), Cin + ex, u43f, b.to(DEV), y, Cout + ey, nimg, Hh, Ww, Cin, Cout, res=nhwc(res, Cout + ey),
                         ldres=Cout + ey, stats_part=part)
    assert H.lib().vd_wino43_last_kernel() == -(Ww // 4)
    y2 = torch.full_like(y, 5.0)
    H.conv3x3_wino43_fwd(nhwc(x, Cin + ex), Cin + ex, u43f, b.to(DEV), y2, Cout + ey, nimg, Hh, Ww, Cin, Cout, res=nhwc(res, Cout + ey),
                         ldres=Cout + ey)
    torch.cuda.synchronize()
    assert torch.equal(y, y2), "not bitwise reproducible / statistics-free launch differs"
    if ey:
        assert (y[..., Cout:] == 5.0).all(), "padding channels of y were written"
    close(from_nhwc(y, Cout), ref64, ref32, slack=8.0, floor=4e-6, name="wino43 forward")
    if Cout % 32 == 0 and (Cout // 32) >= 1 and Cout % 32 == 0 and Cout >= 32:
        rows = H.wino43_fwd_chunk_rows(Hh, Ww)
        stats = torch.empty(nimg, 32, 2, device=DEV)
        H.gn_stats_from_partials([(part, Cout, HW // rows)], nimg, HW, stats)
        g = from_nhwc(y, Cout).double().cpu().reshape(nimg, 32, -1)          # statistics of what was WRITTEN
        mean, var = g.mean(-1), g.var(-1, unbiased=False)
        close(stats[..., 0], mean, None, floor=2e-6, name="mean")
        close(stats[..., 1], 1 / torch.sqrt(var + 1e-6), None, floor=5e-6, name="rstd")
    # no bias, no residual
    y3 = torch.empty(nimg, Hh, Ww, Cout, device=DEV)
    H.conv3x3_wino43_fwd(nhwc(x, Cin + ex), Cin + ex, u43f, None, y3, Cout, nimg, Hh, Ww, Cin, Cout)
    close(from_nhwc(y3, Cout), ref64 - res.double() - b.double().view(1, -1, 1, 1), ref32 - res - b.view(1, -1, 1, 1), slack=8.0, floor=4e-6,
          name="wino43 forward (plain)")


def test_wino43_tensors_beyond_2gib(H):
    """The 32x32 / 64-wide geometries address one image with 32-bit offsets and carry the image's 64-bit offset in the buffer descriptor,
    so x / y / residual / dy / dx may exceed 2 GiB (CelebA sampling at 512 rows: [512, 64, 64, 576] is 4.8 GB).  136 images of 64x64x1024
    = 2.28 GB on the wide side; first, middle and last image against fp64."""
    nimg, Hh, Ww, Cw, Cn = 136, 64, 64, 1024, 32
    assert nimg * Hh * Ww * Cw * 4 > 2 ** 31
    g = torch.Generator(DEV).manual_seed(5)
    big = torch.randn((nimg, Hh, Ww, Cw), device=DEV, generator=g)
    w = torch.randn((Cn, Cw, 3, 3), device=DEV, generator=g) * (9 * Cw) ** -0.5
    pick = (0, nimg // 2, nimg - 1)

    def conv64(xs, wt):
        xp = F.pad(xs.double(), (0, 0, 1, 1, 1, 1))
        out = torch.zeros(xs.shape[0], Hh, Ww, wt.shape[0], dtype=torch.float64, device=DEV)
        for ky in range(3):
            for kx in range(3):
                out += xp[:, ky:ky + Hh, kx:kx + Ww, :] @ wt[:, :, ky, kx].double().T
        return out
    # forward: x beyond 2 GiB (Cin = 1024 -> Cout = 32), residual + statistics on
    assert H.lib().vd_conv3x3_wino43_fwd_supported(nimg, Hh, Ww, Cw, Cn, Cw, Cn, Cn) == 1
    u43f = torch.empty(H.lib().vd_wino43_u_floats(Cn, Cw), device=DEV)
    H.wino43_pack_fwd(w, Cn, Cw, u43f)
    res = torch.randn((nimg, Hh, Ww, Cn), device=DEV, generator=g)
    y = torch.empty(nimg, Hh, Ww, Cn, device=DEV)
    H.conv3x3_wino43_fwd(big, Cw, u43f, None, y, Cn, nimg, Hh, Ww, Cw, Cn, res=res, ldres=Cn)
    for i in pick:
        ref = conv64(big[i:i + 1], w) + res[i:i + 1].double()
        err = (y[i:i + 1].double() - ref).abs().max().item()
        assert err <= 1.5e-5 * max(ref.abs().max().item(), 1.0), f"forward, image {i}: {err:.3e}"
    # input gradient: dx beyond 2 GiB (conv Cin = 1024, Cout = 32: dy has 32 channels)
    assert H.lib().vd_conv3x3_dgrad_wino43_supported(nimg, Hh, Ww, Cw, Cn, Cn, Cw) == 1
    u43 = torch.empty(H.lib().vd_wino43_u_floats(Cn, Cw), device=DEV)
    H.wino43_pack(w, Cn, Cw, u43)
    dy = torch.randn((nimg, Hh, Ww, Cn), device=DEV, generator=g)
    H.conv3x3_dgrad_wino43(dy, Cn, u43, big, Cw, nimg, Hh, Ww, Cw, Cn)            # (dx overwrites the big buffer)
    wrot = w.flip(2, 3).transpose(0, 1).contiguous()
    for i in pick:
        ref = conv64(dy[i:i + 1], wrot)
        err = (big[i:i + 1].double() - ref).abs().max().item()
        assert err <= 6e-5 * max(ref.abs().max().item(), 1.0), f"input gradient, image {i}: {err:.3e}"


def test_conv3x3_dgrad_wino43_rejects_unsupported(H):
    f = H.lib().vd_conv3x3_dgrad_wino43_supported
    assert f(2, 16, 16, 32, 32, 32, 32) == 0             # 16x16 images are served four at a time
    assert f(4, 16, 16, 32, 32, 32, 32) == 1
    assert f(4, 8, 8, 32, 32, 32, 32) == 0               # 8x8 images (served by the F(2x2,3x3) kernels)
    assert f(2, 32, 32, 48, 32, 32, 48) == 0             # Cin % 32
    assert f(2, 32, 32, 32, 12, 12, 32) == 0             # Cout % 8
    assert f(2, 24, 64, 32, 32, 32, 32) == 0             # 64-wide, H % 16
    assert f(2, 32, 32, 32, 32, 32, 32) == 1
    z = torch.zeros(2, 16, 16, 32, device=DEV)
    with pytest.raises(H.HipError, match="unsupported geometry"):
        H.conv3x3_dgrad_wino43(z, 32, torch.zeros(36 * 32 * 32, device=DEV), torch.empty_like(z), 32, 2, 16, 16, 32, 32)


# ------------------------------------------------------------------------------------------------ grouped weight-gradient GEMMs
@pytest.mark.parametrize("count,M,N,K,splitk,pad", [(3, 64, 96, 1000, 4, 8), (9, 256, 256, 8192, 7, 0), (27, 512, 1024, 128, 1, 0),
                                                     (2, 768, 256, 4096, 16, 0), (32, 36, 260, 512, 2, 4)])
def test_gemm_grouped_wgrad(H, count, M, N, K, splitk, pad):
    """vd_gemm_grouped_wgrad: `count` same-shape dW = dY^T X products (operands in unrelated buffers, row pitches > M / N) and their
    bias gradients in one launch == the per-entry fp64 products; bitwise reproducible; equal to the single-launch vd_gemm path to
    rounding.  Reference ops: the weight / bias gradients of modules.py:79-80 (Linear) and :141-144 (1x1 Conv2d)."""
    g = torch.Generator(DEV).manual_seed(count * 1000 + M)
    ents, refs = [], []
    for e in range(count):
        dy = torch.randn((K, M + pad), device=DEV, generator=g)
        x = torch.randn((K, N + pad), device=DEV, generator=g)
        dw = torch.full((M, N + pad), 3.0, device=DEV)
        db = torch.full((M,), 3.0, device=DEV)
        ents.append((dy, x, dw, db))
        refs.append((dy[:, :M].double().T @ x[:, :N].double(), dy[:, :M].double().sum(0)))
    H.gemm_grouped_wgrad(ents, M, N, K, M + pad, N + pad, N + pad, splitk)
    torch.cuda.synchronize()
    for (dy, x, dw, db), (rw, rb) in zip(ents, refs):
        scale = rw.abs().max().item()
        assert (dw[:, :N].double() - rw).abs().max().item() <= 2e-6 * scale * max(1.0, (K / 1024) ** 0.5), "grouped dW"
        assert (db.double() - rb).abs().max().item() <= 2e-6 * max(rb.abs().max().item(), 1.0) * max(1.0, (K / 1024) ** 0.5), "grouped dbias"
        if pad:
            assert (dw[:, N:] == 3.0).all(), "padding columns of dW were written"
    again = [(dy, x, torch.empty_like(dw), torch.empty_like(db)) for dy, x, dw, db in ents]
    H.gemm_grouped_wgrad(again, M, N, K, M + pad, N + pad, N + pad, splitk)
    torch.cuda.synchronize()
    for a, b in zip(ents, again):
        assert torch.equal(a[2][:, :N], b[2][:, :N]) and torch.equal(a[3], b[3]), "not bitwise reproducible"


# ------------------------------------------------------------------------------------------------ F(4x4,3x3) weight gradient (unfused)
WGRAD43_CASES = [  # nimg, H, W, Cin, Cout, Cin_w, Cout_w, ldx_extra, lddy_extra
    (16, 32, 32, 32, 64, 32, 64, 0, 0),        # 1024 tiles
    (128, 8, 8, 64, 96, 64, 96, 0, 0),         # 512 tiles: the smallest served problem (8x8 images at batch 128)
    (9, 64, 32, 48, 36, 48, 36, 16, 12),       # non-square, channel counts off the tile sizes, ld > C on both sides
    (64, 16, 16, 96, 32, 96, 32, 0, 0),        # 16x16 images
    (20, 32, 32, 4, 32, 3, 32, 0, 0), (20, 32, 32, 32, 4, 32, 3, 0, 0),      # padded thin sides
    (4, 64, 64, 192, 192, 192, 192, 0, 0),
    # >= 384 x 384 weights per plane: the fold kernel sums the split-K slabs itself (wino43_wgrad_reduce_finish_kernel) -- accumulate mode,
    # real dims below the padded ones on both sides, row pitches above the channel counts
    (32, 16, 16, 384, 388, 381, 386, 8, 4), (8, 32, 32, 512, 384, 512, 384, 0, 0),
]


@pytest.mark.parametrize("case", WGRAD43_CASES)
def test_conv3x3_wgrad_wino43(H, case):
    """vd_conv3x3_wgrad_wino43 == autograd of F.conv2d with respect to the kernel and the bias (fp64 truth), bitwise reproducible,
    accumulate mode, padded dims.  Bounds: relative L2 8e-6, 4e-5 of the largest element (the stated bound on gradients is 1e-4)."""
    nimg, Hh, Ww, Cin, Cout, Cin_w, Cout_w, ex, ey = case
    assert H.lib().vd_conv3x3_wgrad_wino43_supported(nimg, Hh, Ww, Cin, Cout, Cin + ex, Cout + ey) == 1
    x = torch.zeros(nimg, Cin, Hh, Ww)
    x[:, :Cin_w] = rnd(nimg, Cin_w, Hh, Ww, seed=1)
    dy = torch.zeros(nimg, Cout, Hh, Ww)
    dy[:, :Cout_w] = rnd(nimg, Cout_w, Hh, Ww, seed=2)
    w64 = torch.zeros(Cout_w, Cin_w, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x[:, :Cin_w].double(), w64, padding=1).backward(dy[:, :Cout_w].double())
    dw = torch.ones(Cout_w, Cin_w, 3, 3, device=DEV)
    db = torch.ones(Cout_w, device=DEV)
    xd, dyd = nhwc(x, Cin + ex), nhwc(dy, Cout + ey)
    H.conv3x3_wgrad_wino43(xd, Cin + ex, dyd, Cout + ey, nimg, Hh, Ww, Cin, Cout, dw, Cin_w, Cout_w, accumulate=True, dbias=db)
    torch.cuda.synchronize()
    ref = w64.grad
    got = dw.double().cpu() - 1.0
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel <= 8e-6 and (got - ref).abs().max().item() <= 4e-5 * ref.abs().max().item(), f"wgrad43 rel-L2 {rel:.3e}"
    dbr = dy[:, :Cout_w].double().sum((0, 2, 3))
    assert (db.double().cpu() - 1.0 - dbr).abs().max().item() <= 2e-5 * max(dbr.abs().max().item(), 1.0)
    a, b = torch.empty_like(dw), torch.empty_like(dw)
    H.conv3x3_wgrad_wino43(xd, Cin + ex, dyd, Cout + ey, nimg, Hh, Ww, Cin, Cout, a, Cin_w, Cout_w)
    H.conv3x3_wgrad_wino43(xd, Cin + ex, dyd, Cout + ey, nimg, Hh, Ww, Cin, Cout, b, Cin_w, Cout_w)
    assert torch.equal(a, b), "not bitwise reproducible"
    assert ((a.double().cpu() - ref).norm() / ref.norm()).item() <= 8e-6


def test_conv3x3_wgrad_wino43_rejects_unsupported(H):
    f = H.lib().vd_conv3x3_wgrad_wino43_supported
    assert f(4, 32, 32, 32, 32, 32, 32) == 0             # 256 tiles: too few (the fused F(2x2,3x3) kernel is faster)
    assert f(16, 30, 32, 32, 32, 32, 32) == 0            # H % 4
    assert f(16, 32, 32, 30, 32, 32, 32) == 0            # Cin % 4
    assert f(16, 32, 32, 32, 32, 32, 32) == 1
