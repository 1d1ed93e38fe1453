"""BASELINE.json configs[0]: CIFAR-10 32x32 unconditional v-prediction, batch 16, one train step + an 8-step DDIM chain
(configs/cifar10_uncond.json with --model-out-type v; the reference's own CPU-runnable case).

  * CPU tier: the oracle runs the whole case (plumbing check of the checker itself).
  * GPU tier: the HIP path against the oracle on the same seeded inputs, full-size 60.8 M-parameter model.
"""
import numpy as np
import pytest
import torch

from oracle import unet_ref, diffusion_ref as dref, detrand
from oracle.cases import CIFAR_UNCOND, make_weights

B, R, T = 16, 32, 8


def _inputs():
    x0 = detrand.uniform("c0_x", (B, 3, R, R), 11, lo=-1.0, hi=1.0)
    t = detrand.uniform("c0_t", (B,), 11, dtype=torch.float64)
    noise = detrand.normal("c0_n", (B, 3, R, R), 11)
    x_T = detrand.normal("c0_xT", (B, 3, R, R), 12)
    return x0, t, noise, x_T


def _oracle_case(sd, grads=True):
    x0, t, noise, x_T = _inputs()
    cfg = dict(CIFAR_UNCOND, drop_rate=0.0)
    sdo = {k: v.clone().requires_grad_(grads) for k, v in sd.items()}
    den = lambda a, b, c: unet_ref.unet_forward(sdo, cfg, a, b, c)
    sched = dref.make_schedule("cosine")
    loss = dref.train_loss(den, sched, x0, t, None, noise, "v", "snr_trunc")
    gnorm = None
    if grads:
        loss.mean().backward()
        gnorm = {k: float(v.grad.double().norm()) for k, v in sdo.items()}
    with torch.no_grad():
        xs = dref.p_sample(den, sched, x_T, T, None, [torch.zeros_like(x_T)] * T, model_out_type="v", var_type="fixed_large",
                           w_guide=0.0, use_ddim=True)
    return loss.detach(), gnorm, xs


def test_config0_oracle_plumbing_cpu():
    torch.set_num_threads(max(1, torch.get_num_threads()))
    sd = make_weights(CIFAR_UNCOND, seed=2)
    loss, gnorm, xs = _oracle_case(sd, grads=True)
    assert loss.shape == (B,) and torch.isfinite(loss).all() and float(loss.min()) > 0
    assert all(np.isfinite(v) and v >= 0 for v in gnorm.values()) and max(gnorm.values()) > 0
    assert xs.shape == (B, 3, R, R) and torch.isfinite(xs).all() and float(xs.abs().max()) <= 1.0 + 1e-6   # last step = clipped x0


def _ddim_chain(sd, dtype, rows):
    """the oracle's 8-step DDIM chain on the first ``rows`` rows of the case, evaluated entirely in ``dtype``"""
    _, _, _, x_T = _inputs()
    cfg = dict(CIFAR_UNCOND, drop_rate=0.0)
    sdo = {k: v.to(dtype) for k, v in sd.items()}
    den = lambda a, b, c: unet_ref.unet_forward(sdo, cfg, a, b, c)
    xT = x_T[:rows].to(dtype)
    with torch.no_grad():
        return dref.p_sample(den, dref.make_schedule("cosine"), xT, T, None, [torch.zeros_like(xT)] * T, model_out_type="v",
                             var_type="fixed_large", w_guide=0.0, use_ddim=True)


# How far the reference's own arithmetic moves the end of the 8-step chain: the oracle in fp32 against the oracle in fp64 on the same
# inputs (rows are independent, 4 of the 16 keep the CPU tier short).  Measured here: 7.6e-7 max-abs (8 threads vs 1 thread of the fp32
# evaluation: 1.5e-6).  The chain does not amplify: every step forms x0-hat = alpha x_t - sigma v-hat, clips it to [-1, 1] and
# interpolates, so a perturbation of the network output enters scaled by sigma <= 1.  The GPU test below therefore holds the HIP
# path's chain end to the stated trajectory bound of 1e-4 -- two orders above this spread, five times the stated
# per-evaluation output bound of 2e-5 -- and prints what it measures.
CHAIN_SPREAD_BOUND = 5e-6
CHAIN_END_TOL = 1e-4


def test_config0_oracle_fp32_vs_fp64_chain_spread():
    sd = make_weights(CIFAR_UNCOND, seed=2)
    a, b = _ddim_chain(sd, torch.float32, 4), _ddim_chain(sd, torch.float64, 4)
    spread = float((a.double() - b).abs().max())
    print(f"oracle 8-step DDIM chain end, fp32 vs fp64 evaluation: max-abs {spread:.3e}")
    assert b.dtype == torch.float64 and 0 < spread <= CHAIN_SPREAD_BOUND, spread
    assert CHAIN_END_TOL >= 10 * CHAIN_SPREAD_BOUND


@pytest.mark.gpu
def test_config0_hip_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import v_diffusion
    sd = make_weights(CIFAR_UNCOND, seed=2)
    loss_o, gnorm_o, xs_o = _oracle_case(sd, grads=True)
    model = v_diffusion.UNet(**dict(CIFAR_UNCOND, drop_rate=0.0))
    model.load_state_dict(sd)
    model.to("cuda").train()
    x0, t, noise, x_T = _inputs()
    gd = v_diffusion.GaussianDiffusion(v_diffusion.get_logsnr_schedule("cosine", -20.0, 20.0), T, "v", "fixed_large", "snr_trunc",
                                       "mse", w_guide=0.0, p_uncond=0.0)
    loss = gd.train_loss(model, x0.cuda(), t.cuda(), None, noise.cuda())
    loss.mean().backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy(), loss_o.numpy(), rtol=1e-4, atol=1e-6)
    gmax = max(gnorm_o.values())
    for k, p in model.named_parameters():
        n = float(p.grad.double().norm())
        assert abs(n - gnorm_o[k]) <= 1e-4 * gnorm_o[k] + 1e-6 * gmax, (k, n, gnorm_o[k])
    xs = gd.p_sample(model.eval(), (B, 3, R, R), noise=x_T, label=None, device="cuda", seed=None, use_ddim=True)
    err = float((xs - xs_o).abs().max())
    print(f"config 0: 8-step DDIM chain end, HIP path vs CPU oracle: max-abs {err:.3e} (bound {CHAIN_END_TOL:.0e}; the oracle's own "
          f"fp32-vs-fp64 spread on this chain is <= {CHAIN_SPREAD_BOUND:.0e}: test_config0_oracle_fp32_vs_fp64_chain_spread)")
    assert err <= CHAIN_END_TOL, f"8-step DDIM chain differs from the oracle by {err:.3e}"
