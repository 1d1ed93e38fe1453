"""CPU oracle for the diffusion process -- TEST INFRASTRUCTURE, not a product path.

Restates, in plain torch on CPU:

  * log-SNR schedules .............. reference v_diffusion/diffusion.py:42-112
  * stable log(1-exp(x)) ........... diffusion.py:115-123
  * DDPM posterior coefficients .... diffusion.py:126-163
  * DDIM posterior coefficients .... diffusion.py:169-203
  * prediction conversions ......... diffusion.py:206-245
  * train_loss (mse branch) ........ diffusion.py:492-545 (+ from_model_out_to_pred :466-490)
  * p_mean_var / p_sample_step / p_sample ... diffusion.py:317-414

Schedule and posterior arithmetic is fp64 and cast to fp32 at the end, as in the reference.
"""
import math

import torch
import torch.nn.functional as F

F64 = torch.float64


# ----------------------------------------------------------------------------- schedules
def make_schedule(name, logsnr_min=-20.0, logsnr_max=20.0):
    """Returns f(t) -> logsnr with the dtype of t (fp64 arithmetic inside).  No in-place rescale of t
    (``allow_rescale`` is False in every shipped config, defaults.json:45)."""
    lo, hi = float(logsnr_min), float(logsnr_max)
    if name == "legacy":                                                      # diffusion.py:78-92
        x_max, x_min, slope = 0.9999, 0.98, -0.0199
        c0 = x_max * math.log(x_max) - x_max

        def f(t):
            xt = x_max + (x_min - x_max) * t
            log_alpha = 1000 / slope * (xt * torch.log(xt) - xt - c0)
            return log_alpha - log1mexp(log_alpha - 1e-9)
        return f
    if name == "linear":
        t_of = lambda l: 1.0 / (1.0 + math.exp(-l))
        inv = lambda u: torch.log(u) - torch.log1p(-u)
    elif name == "sigmoid":
        t_of = lambda l: (hi - l) / (hi - lo)
        inv = lambda u: hi - u * (hi - lo)
    elif name == "cosine":
        t_of = lambda l: math.atan(math.exp(-0.5 * l)) / (0.5 * math.pi)
        inv = lambda u: -2.0 * torch.log(torch.tan(u * (0.5 * math.pi)))
    else:
        raise NotImplementedError(name)
    t0, t1 = t_of(hi), t_of(lo)                                               # diffusion.py:98-99

    def f(t):
        w = t.to(F64)
        u = torch.lerp(torch.full_like(w, t0), torch.full_like(w, t1), w)     # diffusion.py:103
        return inv(u).to(t.dtype)
    return f


def log1mexp(x):
    """log(1 - exp(x)) for x < 0 (diffusion.py:115-123)."""
    return torch.where(x < -9, torch.log1p(-torch.exp(x)), torch.log(-torch.expm1(x)))


# ----------------------------------------------------------------------------- posteriors
def ddpm_coefs(logsnr_s, logsnr_t, var_type, intp_frac=None):
    """E[x_s | x_t, x_0] = c1*x_t + c2*x_0 and the log-variance (diffusion.py:126-163, x0eps_coef=False)."""
    ls, lt = logsnr_s.to(F64), logsnr_t.to(F64)
    logr = lt - ls
    l1mr = log1mexp(logr)
    c1 = torch.exp(logr + 0.5 * (F.logsigmoid(ls) - F.logsigmoid(lt)))
    c2 = torch.exp(l1mr + 0.5 * F.logsigmoid(ls))
    v_small = l1mr + F.logsigmoid(-ls)
    v_large = l1mr + F.logsigmoid(-lt)
    if var_type == "fixed_large":
        lv = v_large
    elif var_type == "fixed_small":
        lv = v_small
    elif var_type == "fixed_medium":
        lv = v_small + intp_frac * (v_large - v_small)
    else:
        raise NotImplementedError(var_type)
    return c1.float(), c2.float(), lv.float()


def ddim_coefs(logsnr_s, logsnr_t):
    """eta = 0 DDIM (diffusion.py:169-187): c1 = sigma_s/sigma_t, c2 = alpha_s (1 - sqrt(SNR_t/SNR_s)), logvar = -inf."""
    ls, lt = logsnr_s.to(F64), logsnr_t.to(F64)
    c1 = torch.exp(0.5 * (F.logsigmoid(-ls) - F.logsigmoid(-lt)))
    c2 = torch.exp(log1mexp(0.5 * (lt - ls)) + 0.5 * F.logsigmoid(ls))
    return c1.float(), c2.float(), torch.tensor(-math.inf)


# ----------------------------------------------------------------------------- conversions
def alpha_sigma(logsnr):
    return torch.sigmoid(logsnr).sqrt(), torch.sigmoid(-logsnr).sqrt()


def q_sample(x0, logsnr, eps):                                                # diffusion.py:242-245
    a, s = alpha_sigma(logsnr)
    return x0 * a + eps * s


def x0_from_v(xt, v, logsnr):                                                 # diffusion.py:232-234
    a, s = alpha_sigma(logsnr)
    return xt * a - v * s


def eps_from_v(xt, v, logsnr):                                                # diffusion.py:237-239
    a, s = alpha_sigma(logsnr)
    return xt * s + v * a


def x0_from_eps(xt, eps, logsnr):                                             # diffusion.py:206-208
    return xt * torch.sigmoid(logsnr).rsqrt() - eps * torch.exp(-0.5 * logsnr)


def eps_from_x0(xt, x0, logsnr):                                              # diffusion.py:217-219
    return xt * torch.sigmoid(-logsnr).rsqrt() - x0 * torch.exp(0.5 * logsnr)


def x0_from_both(xt, out, logsnr):                                            # diffusion.py:211-214
    x0, eps = out.chunk(2, dim=1)
    return x0 * torch.sigmoid(-logsnr) + x0_from_eps(xt, eps, logsnr) * torch.sigmoid(logsnr)


def v_from_x0eps(x0, eps, logsnr):                                            # diffusion.py:222-224
    a, s = alpha_sigma(logsnr)
    return -x0 * s + eps * a


def predictions(model_out_type, xt, out, logsnr):
    """x0-, eps- and v-predictions implied by the network output (diffusion.py:466-490)."""
    if model_out_type == "v":
        return x0_from_v(xt, out, logsnr), eps_from_v(xt, out, logsnr), out
    if model_out_type == "x0":
        x0 = out
        eps = eps_from_x0(xt, x0, logsnr)
    elif model_out_type == "eps":
        eps = out
        x0 = x0_from_eps(xt, eps, logsnr)
    elif model_out_type == "both":
        x0 = x0_from_both(xt, out, logsnr)
        eps = eps_from_x0(xt, x0, logsnr)
    else:
        raise NotImplementedError(model_out_type)
    return x0, eps, v_from_x0eps(x0, eps, logsnr)


def _bcast(v, x):
    return v.to(x.dtype).reshape((-1,) + (1,) * (x.ndim - 1))


def _fmean(x):
    return x.flatten(1).mean(dim=1)


# ----------------------------------------------------------------------------- training loss
def train_loss(denoise_fn, schedule, x0, t, y, noise, model_out_type="v", reweight_type="snr_trunc"):
    """Per-sample loss (B,), mse branch.  The p_uncond label drop (diffusion.py:527-529) happens after the
    forward and does not influence this call's value, so it is left to the caller."""
    logsnr = _bcast(schedule(t), x0)
    xt = q_sample(x0, logsnr, noise)
    out = denoise_fn(xt, t, y)
    if reweight_type == "snr_1plus":                                          # compared against model_out (:541)
        return _fmean((v_from_x0eps(x0, noise, logsnr) - out) ** 2)
    if reweight_type == "constant":
        return _fmean((x0 - out) ** 2)
    if reweight_type == "snr":
        return _fmean((noise - out) ** 2)
    if reweight_type == "snr_trunc":
        px0, peps, _ = predictions(model_out_type, xt, out, logsnr)
        return torch.maximum(_fmean((x0 - px0) ** 2), _fmean((noise - peps) ** 2))
    raise NotImplementedError(reweight_type)


# ----------------------------------------------------------------------------- sampling
def p_sample_step(denoise_fn, schedule, xt, step, T, y, noise, *, model_out_type="v", var_type="fixed_large",
                  intp_frac=None, w_guide=0.0, use_ddim=False, clip=True):
    """One reverse step (diffusion.py:360-392).  ``step`` is a python int (identical for the batch);
    ``noise`` is the N(0,1) tensor the reference would draw at :389."""
    B = xt.shape[0]
    s = torch.full((B,), step / T, dtype=F64)
    t = torch.full((B,), (step + 1) / T, dtype=F64)
    ls, lt = _bcast(schedule(s), xt), _bcast(schedule(t), xt)
    cfg = w_guide > 0 and y is not None
    if cfg:                                         # rows interleaved cond, uncond (:369-372)
        x_in = xt.repeat_interleave(2, dim=0)
        t_in = t.repeat_interleave(2, dim=0)
        y_in = y.repeat_interleave(2, dim=0).clone()
        y_in[1::2] = 0
        ls2, lt2 = ls.repeat_interleave(2, dim=0), lt.repeat_interleave(2, dim=0)
    else:
        x_in, t_in, y_in, ls2, lt2 = xt, t, y, ls, lt
    out = denoise_fn(x_in, t_in, y_in)
    px0 = predictions(model_out_type, x_in, out, lt2)[0]
    if clip:
        px0 = px0.clamp(-1.0, 1.0)
    if use_ddim:
        c1, c2, lv = ddim_coefs(ls2, lt2)
    else:
        c1, c2, lv = ddpm_coefs(ls2, lt2, var_type, intp_frac)
    mean = c1 * x_in + c2 * px0
    if step == 0:
        mean = px0                                  # :378
    if cfg:
        mc, mu = mean[0::2], mean[1::2]
        mean = mc + w_guide * (mc - mu)             # :383-385
        if lv.ndim > 0:
            lv = lv[0::2]
    gate = 1.0 if step > 0 else 0.0
    return mean + gate * torch.exp(0.5 * lv) * noise


def p_sample(denoise_fn, schedule, x_T, T, y, noises, **kw):
    """Full reverse chain from ``x_T`` with explicit per-step noises (list indexed by step)."""
    x = x_T
    for step in reversed(range(T)):
        x = p_sample_step(denoise_fn, schedule, x, step, T, y, noises[step], **kw)
    return x
