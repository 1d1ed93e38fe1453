"""Generate tests/golden/*.npz from the REAL reference (build container only; test infrastructure).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_goldens [--skip-celeba]

Imports tqch/v-diffusion-torch from /root/reference through an empty package stub (its
``__init__`` needs torchvision, which is absent; the hot-path modules need only torch+numpy), feeds
it build-owned deterministic weights/inputs (oracle/detrand.py), asserts that the oracle restatement
(oracle/unet_ref.py, oracle/diffusion_ref.py) reproduces it, and stores the REFERENCE's outputs.
Nothing from /root/reference (source, bytecode, pickles) is written into the repo: fixtures hold
numbers only.
"""
import argparse
import math
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True


def import_reference():
    pkg = types.ModuleType("v_diffusion")
    pkg.__path__ = ["/root/reference/v_diffusion"]
    sys.modules["v_diffusion"] = pkg
    from v_diffusion.models.unet import UNet                      # noqa
    from v_diffusion.diffusion import GaussianDiffusion, get_logsnr_schedule   # noqa
    from v_diffusion import diffusion as refdiff, functions as reffn            # noqa
    return UNet, GaussianDiffusion, get_logsnr_schedule, refdiff, reffn


def grad_digest(named_grads):
    """Per-tensor L2 norm + first 16 elements + four +-1 projections over the WHOLE tensor (keeps fixtures small; the projections
    catch what norm + leading elements cannot: a permutation, transposition or sign error deep inside a tensor)."""
    from oracle import detrand
    norms, heads, names, projs = [], [], [], []
    for n, g in named_grads:
        g = g.detach().double().flatten()
        names.append(n)
        norms.append(float(g.norm()))
        h = np.zeros(16)
        k = min(16, g.numel())
        h[:k] = g[:k].numpy()
        heads.append(h)
        projs.append(detrand.projections(n, g))
    return np.array(names), np.array(norms), np.stack(heads), np.stack(projs)


def check(name, a, b, atol, rtol=0.0):
    err = (a.double() - b.double()).abs().max().item()
    ref = b.double().abs().max().item()
    ok = err <= atol + rtol * ref
    print(f"  [{'ok' if ok else 'FAIL'}] {name}: max|oracle-ref| = {err:.3e} (ref max {ref:.3e})")
    assert ok, name
    return err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-celeba", action="store_true")
    ap.add_argument("--only-unet", action="store_true", help="rewrite tests/golden/unet_*.npz only (the other fixtures are untouched)")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RefUNet, RefGD, ref_get_schedule, refdiff, reffn = import_reference()
    sys.path.insert(0, ROOT)
    from oracle import unet_ref, diffusion_ref as dref, detrand
    from oracle.cases import TINY, CIFAR_COND, CELEBA, make_inputs, make_weights, full_size_inputs

    def build_ref(cfg):
        m = RefUNet(**cfg)
        shapes = unet_ref.param_shapes(cfg)
        ref_shapes = [(k, tuple(v.shape)) for k, v in m.named_parameters()]
        assert ref_shapes == [(k, tuple(v)) for k, v in shapes.items()], "parameter names/shapes/order differ"
        assert list(m.state_dict().keys()) == list(shapes.keys())
        sd = make_weights(cfg)
        m.load_state_dict(sd)
        m.eval()
        return m, sd

    # ------------------------------------------------------------------ (i) tiny UNets: output + grads
    for name, case in TINY.items():
        print(f"== {name}")
        cfg, B, R, label = case["cfg"], case["B"], case["R"], case["label"]
        m, sd = build_ref(cfg)
        x, t, y = make_inputs(cfg, B, R, label)
        gout = detrand.normal("gout", (B, cfg["out_channels"], R, R), 1)
        xr = x.clone().requires_grad_(True)
        out = m(xr, t, None if y is None else y.clone())
        (out * gout).sum().backward()
        ref_grads = [(k, p.grad) for k, p in m.named_parameters()]
        # oracle
        sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xo = x.clone().requires_grad_(True)
        taps = {}
        oo = unet_ref.unet_forward(sdo, cfg, xo, t, y, taps=taps)
        (oo * gout).sum().backward()
        check("output", oo, out, 2e-6)
        check("dx", xo.grad, xr.grad, 1e-6, 1e-5)
        for (k, g) in ref_grads:
            check("d" + k, sdo[k].grad, g, 1e-7, 2e-5)
        names, norms, heads, projs = grad_digest(ref_grads)
        np.savez_compressed(os.path.join(GOLD, f"unet_{name}.npz"), out=out.detach().numpy(), dx=xr.grad.numpy(),
                            grad_names=names, grad_norms=norms, grad_heads=heads, grad_projs=projs)

    # ------------------------------------------------------------------ (ii) full-size UNets
    big = [("cifar10_cond", CIFAR_COND, 2, 32, "single")]
    if not args.skip_celeba:
        big.append(("celeba", CELEBA, 1, 64, "multi"))
    for name, cfg, B, R, label in big:
        print(f"== {name} (B={B})")
        m, sd = build_ref(cfg)
        nparam = sum(p.numel() for p in m.parameters())
        print("  params:", nparam)
        x, t, y = full_size_inputs(cfg, B, R, label)     # (single-label: one labelled row and one UNLABELLED row)
        gout = detrand.normal("gout", (B, cfg["out_channels"], R, R), 1)
        out = m(x, t, None if y is None else y.clone())
        (out * gout).sum().backward()
        ref_grads = [(k, p.grad) for k, p in m.named_parameters()]
        sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        oo = unet_ref.unet_forward(sdo, cfg, x, t, y)
        (oo * gout).sum().backward()
        check("output", oo, out, 1e-5)
        worst = 0.0
        for (k, g) in ref_grads:
            rel = (sdo[k].grad.double() - g.double()).norm().item() / max(g.double().norm().item(), 1e-30)
            worst = max(worst, rel)
        print(f"  worst grad rel-L2 oracle vs ref: {worst:.3e}")
        assert worst < 1e-4
        names, norms, heads, projs = grad_digest(ref_grads)
        np.savez_compressed(os.path.join(GOLD, f"unet_{name}.npz"), out=out.detach().numpy(), nparam=nparam,
                            grad_names=names, grad_norms=norms, grad_heads=heads, grad_projs=projs)
        del m, sd, sdo

    if args.only_unet:
        print("unet goldens written to", GOLD)
        return

    # ------------------------------------------------------------------ (iv) embedding / schedule / posterior tables
    print("== tables")
    tt = torch.tensor([0.0, 1e-3, 0.25, 0.5, 0.999, 1.0], dtype=torch.float64)
    tab = {"t_probe": tt.numpy()}
    for dim in (256, 192, 33):
        e_ref = reffn.get_timestep_embedding(tt, dim)
        check(f"temb{dim}", unet_ref.timestep_embedding(tt, dim), e_ref, 0.0)
        tab[f"temb_{dim}"] = e_ref.numpy()
    for sched in ("cosine", "linear", "sigmoid", "legacy"):
        for T in (8, 50, 250):
            grid = torch.arange(T + 1, dtype=torch.float64) / T
            if sched == "linear":
                grid = grid.clamp(1e-6, 1 - 1e-6)
            fr = ref_get_schedule(sched, -20.0, 20.0)
            fo = dref.make_schedule(sched, -20.0, 20.0)
            lr = fr(grid.clone())
            lo = fo(grid.clone())
            check(f"logsnr_{sched}_{T}", lo, lr, 1e-9, 1e-12)
            tab[f"logsnr_{sched}_{T}"] = lr.numpy()
            if sched != "cosine":
                continue
            ls, lt = lr[:-1].float(), lr[1:].float()
            c1, c2, lv = refdiff.logsnr_to_posterior_ddim(ls, lt, 0.0)
            o1, o2, _ = dref.ddim_coefs(ls, lt)
            check(f"ddim_c1_{T}", o1, c1, 0.0); check(f"ddim_c2_{T}", o2, c2, 0.0)
            tab[f"ddim_{T}"] = np.stack([c1.numpy(), c2.numpy()])
            for vt, frac in (("fixed_large", None), ("fixed_small", None), ("fixed_medium", 0.3)):
                c1, c2, lv = refdiff.logsnr_to_posterior(ls, lt, vt, frac)
                o1, o2, ov = dref.ddpm_coefs(ls, lt, vt, frac)
                check(f"ddpm_{vt}_{T}", torch.stack([o1, o2, ov]), torch.stack([c1, c2, lv]), 0.0, 1e-6)
                tab[f"ddpm_{vt}_{T}"] = np.stack([c1.numpy(), c2.numpy(), lv.numpy()])
    np.savez_compressed(os.path.join(GOLD, "tables.npz"), **tab)

    # ------------------------------------------------------------------ (v) train_loss per sample, all variants
    print("== train_loss")
    case = TINY["tinyA"]
    losses = {}
    for mot in ("v", "x0", "eps", "both"):
        cfg = dict(case["cfg"], out_channels=6 if mot == "both" else 3)
        m, sd = build_ref(cfg)
        x0, t, y = make_inputs(cfg, 4, case["R"], case["label"], seed=3)
        x0 = x0.clamp(-1, 1)
        noise = detrand.normal("noise", tuple(x0.shape), 3)
        for rw in ("constant", "snr", "snr_trunc", "snr_1plus"):
            if rw != "snr_trunc" and mot == "both":
                continue        # reference compares the raw 6-channel output with a 3-channel target: shape error
            gd = RefGD(ref_get_schedule("cosine", -20.0, 20.0), 8, mot, "fixed_large", rw, "mse", p_uncond=0.0)
            with torch.no_grad():
                lr = gd.train_loss(m, x0, t.clone(), y.clone(), noise)
                lo = dref.train_loss(lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c),
                                     dref.make_schedule("cosine"), x0, t, y, noise, mot, rw)
            check(f"loss_{mot}_{rw}", lo, lr, 1e-6, 1e-5)
            losses[f"{mot}_{rw}"] = lr.numpy()
    # value + gradient golden for the flagship (v, snr_trunc)
    cfg = case["cfg"]
    m, sd = build_ref(cfg)
    x0, t, y = make_inputs(cfg, 4, case["R"], case["label"], seed=3)
    x0 = x0.clamp(-1, 1)
    noise = detrand.normal("noise", tuple(x0.shape), 3)
    gd = RefGD(ref_get_schedule("cosine", -20.0, 20.0), 8, "v", "fixed_large", "snr_trunc", "mse", p_uncond=0.0)
    loss = gd.train_loss(m, x0, t.clone(), y.clone(), noise)
    loss.mean().backward()
    names, norms, heads, projs = grad_digest([(k, p.grad) for k, p in m.named_parameters()])
    np.savez_compressed(os.path.join(GOLD, "train_loss.npz"), grad_names=names, grad_norms=norms, grad_heads=heads, grad_projs=projs,
                        **{"loss_" + k: v for k, v in losses.items()})

    # ------------------------------------------------------------------ (vi) sampling trajectories (explicit noises)
    print("== p_sample")
    traj = {}
    cfg = case["cfg"]
    m, sd = build_ref(cfg)
    B, R, T = 3, case["R"], 8
    shape = (B, 3, R, R)
    x_T = detrand.normal("x_T", shape, 5)
    y = torch.tensor([1.0, 7.0, 10.0])
    noises = [detrand.normal(f"step{k}", shape, 5) for k in range(T)]
    den = lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c)
    for tag, kw in (("ddim_cfg", dict(use_ddim=True, w_guide=1.0, var_type="fixed_large")),
                    ("ddpm_medium_cfg", dict(use_ddim=False, w_guide=0.5, var_type="fixed_medium", intp_frac=0.3)),
                    ("ddpm_large_nocfg", dict(use_ddim=False, w_guide=0.0, var_type="fixed_large"))):
        gd = RefGD(ref_get_schedule("cosine", -20.0, 20.0), T, "v", kw["var_type"], "snr_trunc", "mse",
                   intp_frac=kw.get("intp_frac"), w_guide=kw["w_guide"], p_uncond=0.0)
        order = iter(reversed(range(T)))

        def fake_normal_(self, *a, **k):            # the reference draws its step noise at diffusion.py:389
            return self.copy_(noises[next(order)])
        with mock.patch.object(torch.Tensor, "normal_", fake_normal_):
            xr = gd.p_sample(m, shape, noise=x_T.clone(), label=y.clone(), device="cpu", seed=None, use_ddim=kw["use_ddim"])
        with torch.no_grad():
            xo = dref.p_sample(den, dref.make_schedule("cosine"), x_T, T, y, noises, model_out_type="v",
                               var_type=kw["var_type"], intp_frac=kw.get("intp_frac"), w_guide=kw["w_guide"],
                               use_ddim=kw["use_ddim"])
        check(f"traj_{tag}", xo, xr, 2e-5)
        traj[tag] = xr.numpy()
    np.savez_compressed(os.path.join(GOLD, "p_sample.npz"), **traj)
    print("goldens written to", GOLD)


if __name__ == "__main__":
    main()
