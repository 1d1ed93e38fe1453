"""The oracle (oracle/*.py) must reproduce the fixtures captured from the real reference
(tests/golden/*.npz, written by oracle/make_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_ref, diffusion_ref as dref, detrand
from oracle.cases import TINY, CIFAR_COND, CELEBA, make_inputs, make_weights


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _digest_check(g, grads, rtol=2e-5):
    names = [str(n) for n in g["grad_names"]]
    assert names == list(grads.keys())
    for i, n in enumerate(names):
        gr = grads[n].double().flatten()
        ref_norm = float(g["grad_norms"][i])
        assert abs(float(gr.norm()) - ref_norm) <= rtol * max(ref_norm, 1e-12), n
        k = min(16, gr.numel())
        np.testing.assert_allclose(gr[:k].numpy(), g["grad_heads"][i][:k], rtol=1e-4, atol=1e-6 * max(ref_norm, 1e-6), err_msg=n)


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
    x, t, y = make_inputs(CIFAR_COND, 2, 32, "single")
    with torch.no_grad():
        out = unet_ref.unet_forward(sd, CIFAR_COND, x, t, y.clamp(min=1))
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
