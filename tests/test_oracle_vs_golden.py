"""The oracle (oracle/*.py) must reproduce the fixtures captured from the real reference
(tests/golden/*.npz, written by oracle/make_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_ref, diffusion_ref as dref, detrand
from oracle.cases import TINY, CIFAR_COND, CELEBA, make_inputs, make_weights, kl_case, full_size_inputs


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _digest_check(g, grads, rtol=2e-5, head_atol=1e-6, floor=0.0):
    """``floor`` (relative to the largest gradient norm) is for tensors whose exact gradient is zero -- a conv bias in front
    of a GroupNorm with one channel per group -- which hold pure rounding noise in the reference too"""
    names = [str(n) for n in g["grad_names"]]
    assert names == list(grads.keys())
    fl = floor * float(np.max(g["grad_norms"]))
    for i, n in enumerate(names):
        gr = grads[n].double().flatten()
        ref_norm = float(g["grad_norms"][i])
        assert abs(float(gr.norm()) - ref_norm) <= rtol * max(ref_norm, 1e-12) + fl, n
        k = min(16, gr.numel())
        np.testing.assert_allclose(gr[:k].numpy(), g["grad_heads"][i][:k], rtol=1e-4, atol=head_atol * max(ref_norm, 1e-6) + fl, err_msg=n)
        if "grad_projs" in g.files:
            # four +-1 projections over the whole tensor (oracle/detrand.py::projections): position and sign of every element.  A random
            # projection of an error vector e is ~ |e|_2, so 4x the norm tolerance keeps the same rel-L2 statement
            pr = detrand.projections(n, gr)
            assert np.abs(pr - g["grad_projs"][i]).max() <= 4 * (rtol * max(ref_norm, 1e-12) + fl), (n, pr, g["grad_projs"][i])


@pytest.mark.parametrize("name", list(TINY))
def test_tiny_unet_fwd_bwd(golden_dir, name):
    case = TINY[name]
    cfg, B, R, label = case["cfg"], case["B"], case["R"], case["label"]
    g = _load(golden_dir, f"unet_{name}.npz")
    sd = {k: v.requires_grad_(True) for k, v in make_weights(cfg).items()}
    x, t, y = make_inputs(cfg, B, R, label)
    x.requires_grad_(True)
    out = unet_ref.unet_forward(sd, cfg, x, t, y)
    gout = detrand.normal("gout", tuple(out.shape), 1)
    (out * gout).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], atol=2e-6, rtol=1e-5)
    _digest_check(g, {k: v.grad for k, v in sd.items()})


def test_param_order_matches_reference(golden_dir):
    g = _load(golden_dir, "unet_cifar10_cond.npz")
    assert [str(n) for n in g["grad_names"]] == list(unet_ref.param_shapes(CIFAR_COND).keys())
    assert int(g["nparam"]) == sum(int(np.prod(s)) for s in unet_ref.param_shapes(CIFAR_COND).values()) == 60806403
    g = _load(golden_dir, "unet_celeba.npz")
    assert [str(n) for n in g["grad_names"]] == list(unet_ref.param_shapes(CELEBA).keys())
    assert int(g["nparam"]) == sum(int(np.prod(s)) for s in unet_ref.param_shapes(CELEBA).values()) == 266825859


def test_cifar_unet_forward(golden_dir):
    g = _load(golden_dir, "unet_cifar10_cond.npz")
    sd = make_weights(CIFAR_COND)
    x, t, y = full_size_inputs(CIFAR_COND, 2, 32, "single")          # (one labelled row, one unlabelled)
    with torch.no_grad():
        out = unet_ref.unet_forward(sd, CIFAR_COND, x, t, y)
    np.testing.assert_allclose(out.numpy(), g["out"], atol=1e-5, rtol=0)


def test_tables(golden_dir):
    g = _load(golden_dir, "tables.npz")
    tt = torch.from_numpy(g["t_probe"])
    for dim in (256, 192, 33):
        np.testing.assert_array_equal(unet_ref.timestep_embedding(tt, dim).numpy(), g[f"temb_{dim}"])
    for sched in ("cosine", "linear", "sigmoid", "legacy"):
        for T in (8, 50, 250):
            grid = torch.arange(T + 1, dtype=torch.float64) / T
            if sched == "linear":
                grid = grid.clamp(1e-6, 1 - 1e-6)
            l = dref.make_schedule(sched)(grid)
            np.testing.assert_allclose(l.numpy(), g[f"logsnr_{sched}_{T}"], rtol=1e-12, atol=1e-9)
    for T in (8, 50, 250):
        l = torch.from_numpy(g[f"logsnr_cosine_{T}"])
        ls, lt = l[:-1].float(), l[1:].float()
        c1, c2, _ = dref.ddim_coefs(ls, lt)
        np.testing.assert_allclose(np.stack([c1.numpy(), c2.numpy()]), g[f"ddim_{T}"], rtol=1e-6)
        for vt, frac in (("fixed_large", None), ("fixed_small", None), ("fixed_medium", 0.3)):
            c1, c2, lv = dref.ddpm_coefs(ls, lt, vt, frac)
            np.testing.assert_allclose(np.stack([c1.numpy(), c2.numpy(), lv.numpy()]), g[f"ddpm_{vt}_{T}"], rtol=1e-6)
    # reference probe recorded in SURVEY.md 8a-12: DDIM c1 at T=50
    np.testing.assert_allclose(g["ddim_50"][0][[0, 1, 25, 49]], [0.00144, 0.5006, 0.9700, 0.99951], rtol=5e-3)


def test_train_loss_variants(golden_dir):
    g = _load(golden_dir, "train_loss.npz")
    case = TINY["tinyA"]
    for mot in ("v", "x0", "eps", "both"):
        cfg = dict(case["cfg"], out_channels=6 if mot == "both" else 3)
        sd = make_weights(cfg)
        x0, t, y = make_inputs(cfg, 4, case["R"], case["label"], seed=3)
        x0 = x0.clamp(-1, 1)
        noise = detrand.normal("noise", tuple(x0.shape), 3)
        den = lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c)
        for rw in ("constant", "snr", "snr_trunc", "snr_1plus"):
            key = f"loss_{mot}_{rw}"
            if key not in g.files:
                continue
            with torch.no_grad():
                lo = dref.train_loss(den, dref.make_schedule("cosine"), x0, t, y, noise, mot, rw)
            np.testing.assert_allclose(lo.numpy(), g[key], rtol=1e-5, atol=1e-6)
    # v / snr_trunc gradient
    cfg = case["cfg"]
    sd = {k: v.requires_grad_(True) for k, v in make_weights(cfg).items()}
    x0, t, y = make_inputs(cfg, 4, case["R"], case["label"], seed=3)
    noise = detrand.normal("noise", tuple(x0.shape), 3)
    den = lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c)
    dref.train_loss(den, dref.make_schedule("cosine"), x0.clamp(-1, 1), t, y, noise, "v", "snr_trunc").mean().backward()
    _digest_check(g, {k: v.grad for k, v in sd.items()})


def test_sampling_trajectories(golden_dir):
    g = _load(golden_dir, "p_sample.npz")
    case = TINY["tinyA"]
    cfg = case["cfg"]
    sd = make_weights(cfg)
    B, R, T = 3, case["R"], 8
    shape = (B, 3, R, R)
    x_T = detrand.normal("x_T", shape, 5)
    y = torch.tensor([1.0, 7.0, 10.0])
    noises = [detrand.normal(f"step{k}", shape, 5) for k in range(T)]
    den = lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c)
    for tag, kw in (("ddim_cfg", dict(use_ddim=True, w_guide=1.0, var_type="fixed_large")),
                    ("ddpm_medium_cfg", dict(use_ddim=False, w_guide=0.5, var_type="fixed_medium", intp_frac=0.3)),
                    ("ddpm_large_nocfg", dict(use_ddim=False, w_guide=0.0, var_type="fixed_large"))):
        with torch.no_grad():
            xo = dref.p_sample(den, dref.make_schedule("cosine"), x_T, T, y, noises, model_out_type="v", **kw)
        np.testing.assert_allclose(xo.numpy(), g[tag], atol=2e-5, rtol=0)


# ---------------------------------------------------------------- extension fixtures (oracle/make_goldens_ext.py)
def test_ext_posterior_tables(golden_dir):
    """eps/x0 form of the posteriors (x0eps_coef=True) and 0 < eta < 1 DDIM coefficients"""
    g = _load(golden_dir, "ext_tables.npz")
    f = dref.make_schedule("cosine")
    for T in (8, 50):
        l = f(torch.arange(T + 1, dtype=torch.float64) / T)
        ls, lt = l[:-1].float(), l[1:].float()
        for vt, frac in (("fixed_large", None), ("fixed_small", None), ("fixed_medium", 0.3)):
            c = dref.ddpm_coefs(ls, lt, vt, frac, x0eps_coef=True)
            np.testing.assert_allclose(np.stack([v.numpy() for v in c]), g[f"ddpm_x0eps_{vt}_{T}"], rtol=1e-6)
        c = dref.ddim_coefs(ls, lt, x0eps_coef=True)
        np.testing.assert_allclose(np.stack([c[0].numpy(), c[1].numpy()]), g[f"ddim_x0eps_{T}"], rtol=1e-6)
        for eta in (0.5, 0.2):
            for xe in (False, True):
                c = dref.ddim_coefs_eta(ls, lt, eta, x0eps_coef=xe)
                np.testing.assert_allclose(np.stack([v.numpy() for v in c]), g[f"ddim_eta{eta}_{'x0eps' if xe else 'xt'}_{T}"], rtol=1e-6)


def test_ext_sampling_trajectories_x0eps(golden_dir):
    g = _load(golden_dir, "ext_p_sample.npz")
    case = TINY["tinyA"]
    cfg = case["cfg"]
    sd = make_weights(cfg)
    B, R, T = 3, case["R"], 8
    shape = (B, 3, R, R)
    x_T = detrand.normal("x_T", shape, 5)
    y = torch.tensor([1.0, 7.0, 10.0])
    noises = [detrand.normal(f"step{k}", shape, 5) for k in range(T)]
    den = lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c)
    for tag, kw in (("ddpm_medium_cfg_x0eps", dict(use_ddim=False, w_guide=0.5, var_type="fixed_medium", intp_frac=0.3)),
                    ("ddpm_large_nocfg_x0eps", dict(use_ddim=False, w_guide=0.0, var_type="fixed_large")),
                    ("ddim_cfg_x0eps", dict(use_ddim=True, w_guide=1.0, var_type="fixed_large"))):
        with torch.no_grad():
            xo = dref.p_sample(den, dref.make_schedule("cosine"), x_T, T, y, noises, model_out_type="v", x0eps_coef=True, **kw)
        np.testing.assert_allclose(xo.numpy(), g[tag], atol=2e-5, rtol=1e-5)


def test_ext_kl_terms(golden_dir):
    g = _load(golden_dir, "ext_kl.npz")
    cfg, x0, t, y, noise, out = kl_case()
    sd = make_weights(cfg)
    den = lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c)
    sched = dref.make_schedule("cosine")
    for mot, vt, frac in (("v", "fixed_large", None), ("v", "fixed_medium", 0.3), ("eps", "fixed_small", None), ("x0", "fixed_large", None)):
        with torch.no_grad():
            lo = dref.train_loss_kl(den, sched, x0, t, y, noise, 8, mot, vt, frac)
        np.testing.assert_allclose(lo.numpy(), g[f"loss_{mot}_{vt}"], rtol=2e-5, atol=1e-6)
    for step in (0, 3, 7):
        ls = sched(torch.full((6,), step / 8, dtype=torch.float64)).float().reshape(-1, 1, 1, 1)
        lt = sched(torch.full((6,), (step + 1) / 8, dtype=torch.float64)).float().reshape(-1, 1, 1, 1)
        xt = dref.q_sample(x0, lt, noise)
        for clip in (False, True):
            a, b, _ = dref.loss_term_bpd(out, x0, xt, ls, lt, "v", "fixed_medium", 0.3, clip)
            np.testing.assert_allclose(np.stack([a.numpy(), b.numpy()]), g[f"terms_{step}_{int(clip)}"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(dref.prior_bpd(sched, x0).numpy(), g["prior"], rtol=1e-6, atol=1e-12)
    # gradient of the (v, fixed_medium) bound through the network
    sdg = {k: v.requires_grad_(True) for k, v in make_weights(cfg).items()}
    deng = lambda a, b, c: unet_ref.unet_forward(sdg, cfg, a, b, c)
    dref.train_loss_kl(deng, sched, x0, t, y, noise, 8, "v", "fixed_medium", 0.3).mean().backward()
    # the decoder-NLL row differentiates log(cdf(a) - cdf(b)) of nearly equal fp32 numbers: individual small elements of
    # the fp32 gradient move by ~1e-5 of the tensor norm between two evaluation orders of the same network
    _digest_check(g, {k: v.grad for k, v in sdg.items()}, rtol=1e-4, head_atol=2e-5, floor=1e-6)
