"""Second fixture set from the REAL reference (build container only; test infrastructure): the eps/x0 form of the
posteriors (``x0eps_coef=True``), 0 < eta < 1 DDIM coefficients, and the variational-bound terms (``loss_type="kl"``).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_goldens_ext

Same rules as oracle/make_goldens.py: the reference is imported from /root/reference, fed deterministic inputs, the
oracle restatement is asserted equal to it, and only the reference's NUMBERS are stored (tests/golden/ext_*.npz).
The reference functions that cannot run (``calc_all_bpd`` unpacks a shape tuple into the batch size and a 3-tuple into
two names, diffusion.py:556,568; ``_prior_bpd`` unpacks a (B,1,1,1) tensor into one name, :550) are not called: the
prior term is pinned on the functions they are built from (``q_mean_var``, ``normal_kl``).
"""
import math
import os
import sys
from unittest import mock

import numpy as np
import torch

from .make_goldens import GOLD, ROOT, check, grad_digest, import_reference


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RefUNet, RefGD, ref_get_schedule, refdiff, reffn = import_reference()
    sys.path.insert(0, ROOT)
    from oracle import unet_ref, diffusion_ref as dref, detrand
    from oracle.cases import TINY, make_inputs, make_weights

    def build_ref(cfg):
        m = RefUNet(**cfg)
        sd = make_weights(cfg)
        m.load_state_dict(sd)
        m.eval()
        return m, sd

    # ------------------------------------------------------------------ posterior tables, eps/x0 form and 0 < eta < 1
    print("== posterior tables (x0eps_coef, eta)")
    tab = {}
    fr = ref_get_schedule("cosine", -20.0, 20.0)
    for T in (8, 50):
        l = fr(torch.arange(T + 1, dtype=torch.float64) / T)
        ls, lt = l[:-1].float(), l[1:].float()
        for vt, frac in (("fixed_large", None), ("fixed_small", None), ("fixed_medium", 0.3)):
            c1, c2, lv = refdiff.logsnr_to_posterior(ls, lt, vt, frac, x0eps_coef=True)
            o1, o2, ov = dref.ddpm_coefs(ls, lt, vt, frac, x0eps_coef=True)
            check(f"ddpm_x0eps_{vt}_{T}", torch.stack([o1, o2, ov]), torch.stack([c1, c2, lv]), 0.0, 1e-6)
            tab[f"ddpm_x0eps_{vt}_{T}"] = np.stack([c1.numpy(), c2.numpy(), lv.numpy()])
        c1, c2, _ = refdiff.logsnr_to_posterior_ddim(ls, lt, 0.0, x0eps_coef=True)
        o1, o2, _ = dref.ddim_coefs(ls, lt, x0eps_coef=True)
        check(f"ddim_x0eps_{T}", torch.stack([o1, o2]), torch.stack([c1, c2]), 0.0, 1e-6)
        tab[f"ddim_x0eps_{T}"] = np.stack([c1.numpy(), c2.numpy()])
        for eta in (0.5, 0.2):
            for xe in (False, True):
                c1, c2, lv = refdiff.logsnr_to_posterior_ddim(ls, lt, eta, x0eps_coef=xe)
                o1, o2, ov = dref.ddim_coefs_eta(ls, lt, eta, x0eps_coef=xe)
                check(f"ddim_eta{eta}_{xe}_{T}", torch.stack([o1, o2, ov]), torch.stack([c1, c2, lv]), 0.0, 1e-6)
                tab[f"ddim_eta{eta}_{'x0eps' if xe else 'xt'}_{T}"] = np.stack([c1.numpy(), c2.numpy(), lv.numpy()])
    np.savez_compressed(os.path.join(GOLD, "ext_tables.npz"), **tab)

    # ------------------------------------------------------------------ sampling trajectories with x0eps_coef=True
    print("== p_sample (x0eps_coef)")
    case = TINY["tinyA"]
    cfg = case["cfg"]
    m, sd = build_ref(cfg)
    den = lambda a, b, c: unet_ref.unet_forward(sd, cfg, a, b, c)
    B, R, T = 3, case["R"], 8
    shape = (B, 3, R, R)
    x_T = detrand.normal("x_T", shape, 5)
    y = torch.tensor([1.0, 7.0, 10.0])
    noises = [detrand.normal(f"step{k}", shape, 5) for k in range(T)]
    traj = {}
    for tag, kw in (("ddpm_medium_cfg_x0eps", dict(use_ddim=False, w_guide=0.5, var_type="fixed_medium", intp_frac=0.3)),
                    ("ddpm_large_nocfg_x0eps", dict(use_ddim=False, w_guide=0.0, var_type="fixed_large")),
                    ("ddim_cfg_x0eps", dict(use_ddim=True, w_guide=1.0, var_type="fixed_large"))):
        gd = RefGD(ref_get_schedule("cosine", -20.0, 20.0), T, "v", kw["var_type"], "snr_trunc", "mse", intp_frac=kw.get("intp_frac"),
                   w_guide=kw["w_guide"], p_uncond=0.0, x0eps_coef=True)
        order = iter(reversed(range(T)))

        def fake_normal_(self, *a, **k):
            return self.copy_(noises[next(order)])
        with mock.patch.object(torch.Tensor, "normal_", fake_normal_):
            xr = gd.p_sample(m, shape, noise=x_T.clone(), label=y.clone(), device="cpu", seed=None, use_ddim=kw["use_ddim"])
        with torch.no_grad():
            xo = dref.p_sample(den, dref.make_schedule("cosine"), x_T, T, y, noises, model_out_type="v", var_type=kw["var_type"],
                               intp_frac=kw.get("intp_frac"), w_guide=kw["w_guide"], use_ddim=kw["use_ddim"], x0eps_coef=True)
        check(f"traj_{tag}", xo, xr, 2e-5, 1e-5)
        traj[tag] = xr.numpy()
    np.savez_compressed(os.path.join(GOLD, "ext_p_sample.npz"), **traj)

    # ------------------------------------------------------------------ variational bound terms
    print("== kl / bpd")
    kl = {}
    x0, t, y = make_inputs(cfg, 6, case["R"], case["label"], seed=3)
    x0 = (x0.clamp(-1, 1) * 127.5).round() / 127.5          # 8-bit data, as the discretised likelihood assumes
    x0[0, :, :2] = 1.0                                       # both saturation branches of the decoder likelihood
    x0[1, :, :2] = -1.0
    t[0] = 0.05                                              # snaps to t = 1/T, s = 0: decoder-NLL row
    noise = detrand.normal("noise", tuple(x0.shape), 3)
    Tk = 8
    for mot, vt, frac in (("v", "fixed_large", None), ("v", "fixed_medium", 0.3), ("eps", "fixed_small", None), ("x0", "fixed_large", None)):
        gd = RefGD(ref_get_schedule("cosine", -20.0, 20.0), Tk, mot, vt, "snr_trunc", "kl", intp_frac=frac, p_uncond=0.0)
        m.zero_grad()
        lr = gd.train_loss(m, x0, t.clone(), y.clone(), noise)
        lr.mean().backward()
        with torch.no_grad():
            lo = dref.train_loss_kl(den, dref.make_schedule("cosine"), x0, t, y, noise, Tk, mot, vt, frac)
        check(f"kl_loss_{mot}_{vt}", lo, lr.detach(), 1e-6, 2e-5)
        tag = f"{mot}_{vt}"
        kl["loss_" + tag] = lr.detach().numpy()
        if tag == "v_fixed_medium":
            names, norms, heads, projs = grad_digest([(k, p.grad) for k, p in m.named_parameters()])
            kl["grad_names"], kl["grad_norms"], kl["grad_heads"], kl["grad_projs"] = names, norms, heads, projs
    # the two terms + prediction on explicit tensors, clipped and not (what calc_all_bpd's loop body evaluates)
    gd = RefGD(ref_get_schedule("cosine", -20.0, 20.0), Tk, "v", "fixed_medium", "snr_trunc", "kl", intp_frac=0.3, p_uncond=0.0)
    out = detrand.normal("out", tuple(x0.shape), 4) * 0.5
    for step in (0, 3, 7):
        s = torch.full((6,), step / Tk, dtype=torch.float64)
        tt = torch.full((6,), (step + 1) / Tk, dtype=torch.float64)
        ls, lt = gd.t2logsnr(s, tt, x=x0)
        xt = refdiff.q_sample(x0, lt, eps=noise)
        for clip in (False, True):
            a, b, c = gd._loss_term_bpd(out, x0, xt, ls, lt, clip_denoised=clip, return_pred=True)
            oa, ob, oc = dref.loss_term_bpd(out, x0, xt, ls, lt, "v", "fixed_medium", 0.3, clip)
            check(f"bpd_terms_{step}_{clip}", torch.cat([oa, ob]), torch.cat([a, b]), 1e-6, 2e-5)
            check(f"bpd_pred_{step}_{clip}", oc, c, 2e-6)
            kl[f"terms_{step}_{int(clip)}"] = np.stack([a.numpy(), b.numpy()])
    # prior term from the pieces _prior_bpd is written with
    lt1 = gd.t2logsnr(torch.ones((6,), dtype=torch.float32), x=x0)[0]
    mean, logvar = refdiff.q_mean_var(x0, lt1)
    pr = reffn.flat_mean(reffn.normal_kl(mean, logvar, torch.zeros(()), torch.zeros(()))) / math.log(2.0)
    check("prior_bpd", dref.prior_bpd(dref.make_schedule("cosine"), x0), pr, 0.0, 1e-6)
    kl["prior"] = pr.numpy()
    np.savez_compressed(os.path.join(GOLD, "ext_kl.npz"), **kl)
    print("extension goldens written to", GOLD)


if __name__ == "__main__":
    main()
