"""Third fixture set from the REAL reference (build container only; test infrastructure): the public diffusion helpers
(``q_sample``, ``q_mean_var``, the ``pred_*`` conversions, ``from_model_out_to_pred``, ``repeat_along_dim`` /
``slice_along_batch`` / ``broadcast_to``), ``p_sample_step`` with a NON-UNIFORM (B,) step tensor, and sampling with a
rescaling schedule (``get_logsnr_schedule(..., rescale=True)``: the network sees the rewritten t, diffusion.py:105-109,363-374).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_goldens_r2

Same rules as oracle/make_goldens.py: only the reference's NUMBERS are stored (tests/golden/r2_*.npz)."""
import os
import sys
from unittest import mock

import numpy as np
import torch

from .make_goldens import GOLD, ROOT, import_reference


def helper_inputs():
    from oracle import detrand
    B, C, R = 5, 3, 8
    x0 = detrand.normal("h_x0", (B, C, R, R), 21).clamp(-1, 1)
    eps = detrand.normal("h_eps", (B, C, R, R), 22)
    out = detrand.normal("h_out", (B, C, R, R), 23) * 0.7
    out2 = detrand.normal("h_out2", (B, 2 * C, R, R), 24) * 0.7
    logsnr = torch.tensor([-12.0, -3.5, 0.25, 4.0, 15.0]).reshape(B, 1, 1, 1)
    return x0, eps, out, out2, logsnr


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RefUNet, RefGD, ref_get_schedule, refdiff, reffn = import_reference()
    sys.path.insert(0, ROOT)
    from oracle import unet_ref, diffusion_ref as dref, detrand
    from oracle.cases import TINY, make_weights

    # ------------------------------------------------------------------ helpers
    print("== helpers")
    g = {}
    x0, eps, out, out2, l = helper_inputs()
    xt = refdiff.q_sample(x0, l, eps=eps)
    g["q_sample"] = xt.numpy()
    m, lv = refdiff.q_mean_var(x0, l)
    g["q_mean"], g["q_logvar"] = m.numpy(), lv.numpy()
    g["pred_x0_from_eps"] = refdiff.pred_x0_from_eps(xt, out, l).numpy()
    g["pred_x0_from_x0eps"] = refdiff.pred_x0_from_x0eps(xt, out2, l).numpy()
    g["pred_eps_from_x0"] = refdiff.pred_eps_from_x0(xt, out, l).numpy()
    g["pred_v_from_x0eps"] = refdiff.pred_v_from_x0eps(x0, eps, l).numpy()
    g["pred_v_from_x0"] = refdiff.pred_v_from_x0(xt, out, l).numpy()
    g["pred_x0_from_v"] = refdiff.pred_x0_from_v(xt, out, l).numpy()
    g["pred_eps_from_v"] = refdiff.pred_eps_from_v(xt, out, l).numpy()
    for mot in ("v", "x0", "eps", "both"):
        gd = RefGD(ref_get_schedule("cosine", -20.0, 20.0), 8, mot, "fixed_large", "snr_trunc", "mse")
        d = gd.from_model_out_to_pred(xt, out2 if mot == "both" else out, l)
        g[f"fmo_{mot}_constant"], g[f"fmo_{mot}_snr"], g[f"fmo_{mot}_snr_1plus"] = d["constant"].numpy(), d["snr"].numpy(), d["snr_1plus"].numpy()
        assert torch.equal(d["snr_trunc"][0], d["constant"]) and torch.equal(d["snr_trunc"][1], d["snr"])
    r = refdiff.repeat_along_dim(x0[:, :, 0, 0], 3, dim=0)
    g["repeat_dim0"] = r.numpy()
    g["repeat_dim1"] = refdiff.repeat_along_dim(x0[:, :, 0, 0], 2, dim=1).numpy()
    sl = refdiff.slice_along_batch(r, 3)
    g["slice_0"], g["slice_2"] = sl[0].numpy(), sl[2].numpy()
    g["broadcast"] = refdiff.broadcast_to([1.0, 2.0, 3.0], x0).numpy()
    np.savez_compressed(os.path.join(GOLD, "r2_helpers.npz"), **g)

    # ------------------------------------------------------------------ p_sample_step with per-sample steps; rescale sampling
    print("== per-sample steps / rescale")
    case = TINY["tinyA"]
    cfg = case["cfg"]
    net = RefUNet(**cfg)
    sd = make_weights(cfg)
    net.load_state_dict(sd)
    net.eval()
    B, R, T = 4, case["R"], 8
    shape = (B, 3, R, R)
    xs = detrand.normal("ps_x", shape, 31)
    y = torch.tensor([1.0, 7.0, 10.0, 3.0])
    nz = detrand.normal("ps_noise", shape, 32)
    st = {}

    def fake_normal_(self, *a, **k):
        return self.copy_(nz)
    for tag, kw in (("ddim_cfg", dict(use_ddim=True, w_guide=1.0, vt="fixed_large", frac=None)),
                    ("ddpm_medium_cfg", dict(use_ddim=False, w_guide=0.5, vt="fixed_medium", frac=0.3)),
                    ("ddpm_large_nocfg", dict(use_ddim=False, w_guide=0.0, vt="fixed_large", frac=None))):
        gd = RefGD(ref_get_schedule("cosine", -20.0, 20.0), T, "v", kw["vt"], "snr_trunc", "mse", intp_frac=kw["frac"],
                   w_guide=kw["w_guide"], p_uncond=0.0)
        step = torch.tensor([0.0, 3.0, 7.0, 5.0], dtype=torch.float64)
        with torch.no_grad(), mock.patch.object(torch.Tensor, "normal_", fake_normal_):
            sample, pred = gd.p_sample_step(net, xs.clone(), step.clone(), y.clone(), return_pred=True, use_ddim=kw["use_ddim"])
        st[f"step_{tag}_sample"], st[f"step_{tag}_pred"] = sample.numpy(), pred.numpy()
    # sampling with rescale=True: t is rewritten to logsnr2t(logsnr(t)) before the network call (a no-op in exact arithmetic for
    # the cosine schedule -- so the float rescale, t *= 0.5, is the case that tells a missed rescale apart)
    noises = [detrand.normal(f"rs_step{k}", shape, 33) for k in range(T)]
    x_T = detrand.normal("rs_xT", shape, 34)
    for tag, rescale in (("bool", True), ("half", 0.5)):
        gd = RefGD(ref_get_schedule("cosine", -20.0, 20.0, rescale=rescale), T, "v", "fixed_large", "snr_trunc", "mse",
                   w_guide=1.0, p_uncond=0.0)
        order = iter(reversed(range(T)))

        def fake_seq_(self, *a, **k):
            return self.copy_(noises[next(order)])
        with mock.patch.object(torch.Tensor, "normal_", fake_seq_):
            xr = gd.p_sample(net, shape, noise=x_T.clone(), label=y.clone(), device="cpu", seed=None, use_ddim=True)
        st[f"rescale_{tag}"] = xr.numpy()
        # train_loss sees the rescaled t as well
        gdt = RefGD(ref_get_schedule("cosine", -20.0, 20.0, rescale=rescale), T, "v", "fixed_large", "snr_trunc", "mse", p_uncond=0.0)
        t = detrand.uniform("rs_t", (B,), 35, dtype=torch.float64)
        with torch.no_grad():
            st[f"rescale_{tag}_loss"] = gdt.train_loss(net, xs.clamp(-1, 1), t.clone(), y.clone(), nz).numpy()
    np.savez_compressed(os.path.join(GOLD, "r2_steps.npz"), **st)
    print("round-2 goldens written to", GOLD)


if __name__ == "__main__":
    main()
