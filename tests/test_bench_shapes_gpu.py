"""Parity at the BENCHMARK'S OWN SHAPES (BASELINE configs[1]: CIFAR-10 cond UNet, per-GPU batch 128), at natural kernel
selection -- no env knobs.  The small-shape tests of test_kernels_gpu.py reach the KT = 16 / 4-workgroups-per-CU
instantiations only through VD_GEMM_KT; here the launches are big enough to select them by themselves, and each test
asserts (vd_gemm_last_tile) that the instantiation named in DESIGN section 1 really ran:

  test_conv3x3_stats_at_bench_shapes   gemm_dma_kernel<128,128,IM2COL,ROW,false,16,false>   statistics-emitting forward (32x32 layers)
                                       gemm_dma_kernel<128,128,IM2COL,ROW,false,32,false>   the same for the 16x16 layers (512 workgroups)
  test_conv3x3_dgrad_at_bench_shape    gemm_dma_kernel<128,128,IM2COL,ROW,false,16,true>    forward / input gradient (TR epilogue)
  test_conv3x3_wgrad_at_bench_shapes   gemm_dma_kernel<128,128,COL,IM2COL,true,16,true>     weight gradient, 131072 / 32768 pixels
  test_cifar_train_step_b64_vs_oracle  one full CIFAR-cond train step at B = 64 vs the CPU oracle: wino_conv_kernel for the 108
                                       residual-block convolutions + the weight-gradient form above; ..._direct_convolutions re-runs
                                       it with VD_WINO=0 (the direct forms above inside the full step)

Truth = fp64: F.conv2d on the CPU for a subset of images (exact, slow) and fp64 matmuls on the GPU for whole tensors (the
GPU-side checker is itself validated against the CPU one inside the test).  Reference ops: modules.py:141-144 (conv),
unet.py:28-30 (GroupNorm(32, C, eps=1e-6))."""
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


def _tile(H):
    t = H.lib().vd_gemm_last_tile()
    return dict(tr=t // 100000000, kt=(t // 1000000) % 100, bm=(t // 1000) % 1000, bn=t % 1000)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(DEV).manual_seed(seed)
    return torch.randn(shape, device=DEV, generator=g) * scale


def _conv_fp64_gpu(x_nhwc, w_oihw, bias):
    """3x3 cross-correlation, pad 1, as nine fp64 matmuls on the device: [B,H,W,Cin] -> [B,H,W,Cout]"""
    B, Hh, Ww, Cin = x_nhwc.shape
    xp = F.pad(x_nhwc.double(), (0, 0, 1, 1, 1, 1))
    out = bias.double().reshape(1, 1, 1, -1).repeat(B, Hh, Ww, 1)
    for ky in range(3):
        for kx in range(3):
            out += xp[:, ky:ky + Hh, kx:kx + Ww, :] @ w_oihw[:, :, ky, kx].double().T
    return out


# (the 16x16 layers of the batch-128 step are 512-workgroup launches: they keep the KT = 32 form, 2 workgroups per CU)
@pytest.mark.parametrize("case", [(64, 32, 32, 256, 256, 16), (128, 32, 32, 256, 256, 16), (128, 16, 16, 512, 256, 32)])
def test_conv3x3_stats_at_bench_shapes(H, case):
    nimg, Hh, Ww, Cin, Cout, kt_expected = case
    HW = Hh * Ww
    x = F.silu(_rand((nimg, Hh, Ww, Cin), 1))                     # what the conv really sees: SiLU(GroupNorm(.))
    w = _rand((Cout, Cin, 3, 3), 2, (9 * Cin) ** -0.5)
    b = _rand((Cout,), 3)
    res = _rand((nimg, Hh, Ww, Cout), 4)
    wf = torch.empty(Cout, 9, Cin, device=DEV)
    H.pack_conv3x3(w, Cout, Cin, wf=wf, Cin_p=Cin)
    y = torch.empty(nimg, Hh, Ww, Cout, device=DEV)
    part = torch.full((H.stats_part_numel(nimg, HW, Cout),), 7.0, device=DEV)
    H.conv3x3(x, Cin, wf, b, y, Cout, nimg, Hh, Ww, Cin, Cout, res=res, ldres=Cout, stats_part=part)
    tl = _tile(H)
    assert tl == dict(tr=0, kt=kt_expected, bm=128, bn=128), f"expected the statistics-emitting KT={kt_expected} 128x128 form, got {tl}"
    stats = torch.empty(nimg, 32, 2, device=DEV)
    H.gn_stats_from_partials([(part, Cout, HW // (tl["bm"] // 2))], nimg, HW, stats)
    torch.cuda.synchronize()
    # ---- truth on the device (all images) ...
    ref = _conv_fp64_gpu(x, w, b) + res.double()
    err = (y.double() - ref).abs().max().item()
    assert err <= 1.5e-5 * max(ref.abs().max().item(), 1.0), f"conv output max err {err:.3e}"
    grp = ref.reshape(nimg, HW, 32, Cout // 32).permute(0, 2, 1, 3).reshape(nimg, 32, -1)
    mean, var = grp.mean(-1), grp.var(-1, unbiased=False)
    assert (stats[..., 0].double() - mean).abs().max().item() <= 2e-6 * max(mean.abs().max().item(), 1.0)
    rstd = 1 / torch.sqrt(var + 1e-6)
    assert ((stats[..., 1].double() - rstd) / rstd).abs().max().item() <= 5e-6
    # ---- ... and the device-side checker against fp64 F.conv2d on the host for three whole images
    for i in (0, nimg // 2, nimg - 1):
        xi = x[i:i + 1].permute(0, 3, 1, 2).double().cpu()
        ri = F.conv2d(xi, w.double().cpu(), b.double().cpu(), padding=1).permute(0, 2, 3, 1) + res[i:i + 1].double().cpu()
        assert (ri - ref[i:i + 1].cpu()).abs().max().item() <= 1e-11 * max(ri.abs().max().item(), 1.0)
        assert (y[i:i + 1].double().cpu() - ri).abs().max().item() <= 1.5e-5 * max(ri.abs().max().item(), 1.0)


def test_conv3x3_dgrad_at_bench_shape(H):
    """the statistics-free launch (input gradient = the same implicit GEMM on the rotated kernel): transposed epilogue"""
    nimg, Hh, Ww, Cin, Cout = 64, 32, 32, 256, 256
    dy = _rand((nimg, Hh, Ww, Cout), 5)
    w = _rand((Cout, Cin, 3, 3), 6, (9 * Cin) ** -0.5)
    wd = torch.empty(Cin, 9, Cout, device=DEV)
    H.pack_conv3x3(w, Cout, Cin, wd=wd, Cout_p=Cout)
    dx = torch.empty(nimg, Hh, Ww, Cin, device=DEV)
    H.conv3x3(dy, Cout, wd, None, dx, Cin, nimg, Hh, Ww, Cout, Cin)
    tl = _tile(H)
    assert tl == dict(tr=1, kt=16, bm=128, bn=128), tl
    torch.cuda.synchronize()
    wrot = w.flip(2, 3).transpose(0, 1).contiguous()              # conv_transpose == conv with the rotated, transposed kernel
    ref = _conv_fp64_gpu(dy, wrot, torch.zeros(Cin, device=DEV))
    err = (dx.double() - ref).abs().max().item()
    assert err <= 1.5e-5 * max(ref.abs().max().item(), 1.0), f"dgrad max err {err:.3e}"


@pytest.mark.parametrize("case", [(128, 32, 32, 256, 256), (128, 16, 16, 256, 256), (128, 32, 32, 512, 256)])
def test_conv3x3_wgrad_at_bench_shapes(H, case):
    nimg, Hh, Ww, Cin, Cout = case
    x = F.silu(_rand((nimg, Hh, Ww, Cin), 7))
    dy = _rand((nimg, Hh, Ww, Cout), 8, 0.05)
    dw = torch.full((Cout, Cin, 3, 3), 3.0, device=DEV)
    db = torch.full((Cout,), 3.0, device=DEV)
    H.conv3x3_wgrad(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw, Cin, Cout, dbias=db)
    tl = _tile(H)
    assert tl == dict(tr=1, kt=16, bm=128, bn=128), f"expected the KT=16 split-K weight-gradient form, got {tl}"
    dw2 = torch.empty_like(dw)
    H.conv3x3_wgrad(x, Cin, dy, Cout, nimg, Hh, Ww, Cin, Cout, dw2, Cin, Cout, dbias=None)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2), "weight gradient is not bitwise reproducible"
    xp = F.pad(x.double(), (0, 0, 1, 1, 1, 1))
    d2 = dy.double().reshape(-1, Cout)
    ref = torch.empty(Cout, Cin, 3, 3, dtype=torch.float64, device=DEV)
    for ky in range(3):
        for kx in range(3):
            ref[:, :, ky, kx] = d2.T @ xp[:, ky:ky + Hh, kx:kx + Ww, :].reshape(-1, Cin)
    rel = ((dw.double() - ref).norm() / ref.norm()).item()
    err = (dw.double() - ref).abs().max().item()
    assert rel <= 2e-6 and err <= 2e-5 * ref.abs().max().item(), f"wgrad rel-L2 {rel:.3e}, max err {err:.3e}"
    dbr = d2.sum(0)
    assert (db.double() - dbr).abs().max().item() <= 2e-5 * max(dbr.abs().max().item(), 1.0)
    # the device-side checker against autograd of fp64 F.conv2d on the host, first 2 images
    xs, ds = x[:2].permute(0, 3, 1, 2).double().cpu(), dy[:2].permute(0, 3, 1, 2).double().cpu()
    wz = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xs, wz, padding=1).backward(ds)
    part = torch.empty_like(ref)
    for ky in range(3):
        for kx in range(3):
            part[:, :, ky, kx] = d2[:2 * Hh * Ww].T @ xp[:2, ky:ky + Hh, kx:kx + Ww, :].reshape(-1, Cin)
    assert (part.cpu() - wz.grad).abs().max().item() <= 1e-10 * wz.grad.abs().max().item()


def test_cifar_train_step_b64_vs_oracle():
    """One full CIFAR-cond train-step forward/backward at B = 64 (every 32x32 conv launch has 512+ row tiles x 2 column tiles
    = the KT = 16 forms; the 32x32 weight gradients see 65536 pixels = the wide split-K form) against the CPU oracle on the
    same inputs: per-sample loss and per-tensor gradient norms + leading elements (reference train_utils.py:137-154)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import v_diffusion
    from v_diffusion import _hip
    from oracle import unet_ref, diffusion_ref as dref, detrand
    from oracle.cases import CIFAR_COND, make_inputs, make_weights
    cfg = dict(CIFAR_COND, drop_rate=0.0)
    B = 64
    sd = make_weights(cfg)
    model = v_diffusion.UNet(**cfg)
    model.load_state_dict(sd)
    model.to(DEV).train()
    x0, t, y = make_inputs(cfg, B, 32, "single", seed=11)
    x0 = x0.clamp(-1, 1)
    noise = detrand.normal("noise", tuple(x0.shape), 11)
    gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), 50, "v", "fixed_medium", "snr_trunc",
                                       "mse", intp_frac=0.3, w_guide=1.0, p_uncond=0.0)
    seen = set()
    orig = _hip.lib().vd_gemm_last_tile

    def spy_conv(*a, **k):
        r = real_conv(*a, **k)
        seen.add(("conv", orig()))
        return r

    def spy_wgrad(*a, **k):
        r = real_wgrad(*a, **k)
        seen.add(("wgrad", orig()))
        return r
    wino_calls = [0]

    def spy_wino(*a, **k):
        wino_calls[0] += 1
        return real_wino(*a, **k)
    real_conv, real_wgrad, real_wino = _hip.conv3x3, _hip.conv3x3_wgrad, _hip.conv3x3_wino
    _hip.conv3x3, _hip.conv3x3_wgrad, _hip.conv3x3_wino = spy_conv, spy_wgrad, spy_wino
    try:
        loss = gd.train_loss(model, x0.to(DEV), t.to(DEV), y.to(DEV), noise.to(DEV))
        loss.mean().backward()
        torch.cuda.synchronize()
    finally:
        _hip.conv3x3, _hip.conv3x3_wgrad, _hip.conv3x3_wino = real_conv, real_wgrad, real_wino
    code = lambda tr, kt: ((tr * 100 + kt) * 1000 + 128) * 1000 + 128
    assert ("wgrad", code(1, 16)) in seen, sorted(seen)
    if _hip.WINO:           # every residual-block convolution of this network is served by the Winograd kernel
        assert wino_calls[0] == 108 and not any(k == "conv" for k, _ in seen), (wino_calls, sorted(seen))
    else:                   # VD_WINO=0 (test_cifar_train_step_b64_direct_convolutions): the direct implicit-GEMM forms
        assert ("conv", code(0, 16)) in seen and ("conv", code(1, 16)) in seen and wino_calls[0] == 0, sorted(seen)
    # ---- CPU oracle, same weights / inputs
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    den = lambda a, b, c: unet_ref.unet_forward(sdo, cfg, a, b, c)
    lo = dref.train_loss(den, dref.make_schedule("cosine"), x0, t, y, noise, "v", "snr_trunc")
    lo.mean().backward()
    assert torch.allclose(loss.detach().cpu(), lo.detach(), rtol=2e-4, atol=1e-6), (loss[:4], lo[:4])
    gmax = max(v.grad.norm().item() for v in sdo.values())
    worst = 0.0
    for k, p in model.named_parameters():
        ref = sdo[k].grad
        err = (p.grad.cpu() - ref).norm().item()
        worst = max(worst, err / max(ref.norm().item(), 1e-2 * gmax))
        assert err <= 1e-4 * ref.norm().item() + 1e-6 * gmax, f"{k}: rel-L2 {err / max(ref.norm().item(), 1e-30):.3e}"
    print(f"B=64 train step vs oracle: worst per-tensor gradient rel-L2 {worst:.2e}")


def test_cifar_train_step_b64_direct_convolutions():
    """the same step with the Winograd path switched off (VD_WINO=0, read once per process): the KT = 16 direct
    implicit-GEMM instantiations inside the full step, against the same oracle"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "--no-header", "-p", "no:cacheprovider",
                        "-k", "test_cifar_train_step_b64_vs_oracle"], env=dict(os.environ, VD_WINO="0"), capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
