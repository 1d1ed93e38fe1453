"""CPU oracle for the diffusion process -- TEST INFRASTRUCTURE, not a product path.

Restates, in plain torch on CPU:

  * log-SNR schedules .............. reference v_diffusion/diffusion.py:42-112
  * stable log(1-exp(x)) ........... diffusion.py:115-123
  * DDPM posterior coefficients .... diffusion.py:126-163
  * DDIM posterior coefficients .... diffusion.py:169-203
  * prediction conversions ......... diffusion.py:206-245
  * train_loss (mse branch) ........ diffusion.py:492-545 (+ from_model_out_to_pred :466-490)
  * p_mean_var / p_sample_step / p_sample ... diffusion.py:317-414
  * eps/x0 form of the posteriors (x0eps_coef=True) ... diffusion.py:137-140,180-182,195-197,338-347
  * variational-bound terms (loss_type="kl") ... diffusion.py:446-464,497-515 + functions.py:31-63

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
def ddpm_coefs(logsnr_s, logsnr_t, var_type, intp_frac=None, x0eps_coef=False):
    """E[x_s | x_t, x_0] = c1*x_t + c2*x_0 and the log-variance (diffusion.py:126-163); with ``x0eps_coef`` the mean
    is written c1*eps + c2*x_0 instead (:137-140)."""
    ls, lt = logsnr_s.to(F64), logsnr_t.to(F64)
    logr = lt - ls
    l1mr = log1mexp(logr)
    if x0eps_coef:
        c1 = torch.exp(0.5 * (F.logsigmoid(ls) - lt) + logr)
        c2 = torch.sigmoid(ls).sqrt()
    else:
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


def ddim_coefs(logsnr_s, logsnr_t, x0eps_coef=False):
    """eta = 0 DDIM (diffusion.py:169-187): c1 = sigma_s/sigma_t, c2 = alpha_s (1 - sqrt(SNR_t/SNR_s)), logvar = -inf.
    With ``x0eps_coef`` the reference returns the LOGARITHMS of (sigma_s, alpha_s) -- it forgets the exp at :180-182 --
    and that is what its sampler then multiplies with; restated as is."""
    ls, lt = logsnr_s.to(F64), logsnr_t.to(F64)
    if x0eps_coef:
        return (0.5 * F.logsigmoid(-ls)).float(), (0.5 * F.logsigmoid(ls)).float(), torch.tensor(-math.inf)
    c1 = torch.exp(0.5 * (F.logsigmoid(-ls) - F.logsigmoid(-lt)))
    c2 = torch.exp(log1mexp(0.5 * (lt - ls)) + 0.5 * F.logsigmoid(ls))
    return c1.float(), c2.float(), torch.tensor(-math.inf)


def ddim_coefs_eta(logsnr_s, logsnr_t, eta, x0eps_coef=False):
    """0 < eta < 1 (diffusion.py:188-203; eta = 1 is the fixed_small DDPM posterior, :173-174)."""
    ls, lt = logsnr_s.to(F64), logsnr_t.to(F64)
    logr = lt - ls
    l1mr = log1mexp(logr)
    lv = l1mr + F.logsigmoid(-ls) + 2 * math.log(eta)
    w = log1mexp(2 * math.log(eta) + l1mr)
    if x0eps_coef:
        c1 = 0.5 * (w + F.logsigmoid(-ls))
        c2 = 0.5 * F.logsigmoid(ls)
    else:
        c1 = 0.5 * (w + F.logsigmoid(-ls) - F.logsigmoid(-lt))
        c2 = log1mexp(0.5 * (logr + w)) + 0.5 * F.logsigmoid(ls)
    return c1.exp().float(), c2.exp().float(), lv.float()


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
                  intp_frac=None, w_guide=0.0, use_ddim=False, clip=True, x0eps_coef=False):
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
    first = x_in
    if x0eps_coef:                                  # the posterior is taken over (eps, x0) instead of (x_t, x0), :338-347
        first = eps_from_x0(x_in, px0, lt2) if (clip or model_out_type != "eps") else out
    if use_ddim:
        c1, c2, lv = ddim_coefs(ls2, lt2, x0eps_coef)
    else:
        c1, c2, lv = ddpm_coefs(ls2, lt2, var_type, intp_frac, x0eps_coef)
    mean = c1 * first + c2 * px0
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


# ----------------------------------------------------------------------------- variational bound (loss_type = "kl")
def normal_kl(mean1, logvar1, mean2, logvar2):                                # functions.py:31-37
    d = logvar1 - logvar2
    return 0.5 * (-1.0 - d + (mean1 - mean2) ** 2 * torch.exp(-logvar2) + torch.exp(d))


def approx_std_normal_cdf(x):                                                 # functions.py:40-47 (Page 1977)
    return 0.5 * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def discretized_gaussian_loglik(x, means, log_scale, precision=1.0 / 255, cutoff=0.999, tol=1e-12):   # functions.py:50-67
    xc = x - means
    inv_std = torch.exp(-log_scale)
    cdf_u = torch.where(x > cutoff, torch.ones((), dtype=torch.float32), approx_std_normal_cdf(inv_std * (xc + precision)))
    cdf_l = torch.where(x < -cutoff, torch.zeros((), dtype=torch.float32), approx_std_normal_cdf(inv_std * (xc - precision)))
    return torch.log(torch.clamp(cdf_u - cdf_l - tol, min=0) + tol)


def loss_term_bpd(out, x0, xt, logsnr_s, logsnr_t, model_out_type="v", var_type="fixed_large", intp_frac=None, clip=False):
    """(kl, decoder_nll, pred_x0) per sample, in bits per dimension (diffusion.py:446-464)."""
    tc1, tc2, tlv = ddpm_coefs(logsnr_s, logsnr_t, "fixed_small")
    true_mean = tc1 * xt + tc2 * x0
    px0 = predictions(model_out_type, xt, out, logsnr_t)[0]
    if clip:
        px0 = px0.clamp(-1.0, 1.0)
    c1, c2, lv = ddpm_coefs(logsnr_s, logsnr_t, var_type, intp_frac)
    model_mean = c1 * xt + c2 * px0
    kl = _fmean(normal_kl(true_mean, tlv, model_mean, lv)) / math.log(2.0)
    nll = _fmean(-discretized_gaussian_loglik(x0, px0, 0.5 * lv)) / math.log(2.0)
    return kl, nll, px0


def train_loss_kl(denoise_fn, schedule, x0, t, y, noise, T, model_out_type="v", var_type="fixed_large", intp_frac=None):
    """Per-sample un-weighted bound term (diffusion.py:497-515): t snapped up to the sampling grid, KL for s > 0,
    decoder NLL for the last step."""
    t = torch.ceil(t * T) / T
    s = (t - 1.0 / T).clamp(min=0.0)
    lt = _bcast(schedule(t), x0)
    xt = q_sample(x0, lt, noise)
    out = denoise_fn(xt, t, y)
    ls = _bcast(schedule(s), x0)
    kl, nll, _ = loss_term_bpd(out, x0, xt, ls, lt, model_out_type, var_type, intp_frac, clip=False)
    return torch.where(s != 0, kl, nll)


def prior_bpd(schedule, x0):
    """KL(q(x_1 | x_0) || N(0, I)) in bits/dim -- what diffusion.py:547-553 is written to compute (the reference line
    ``logsnr_t, = ...`` only unpacks for a batch of one; q_mean_var :249-250 and normal_kl are used as there)."""
    lt = _bcast(schedule(torch.ones((x0.shape[0],), dtype=torch.float32)), x0)
    mean, logvar = torch.sigmoid(lt).sqrt() * x0, F.logsigmoid(-lt)
    return _fmean(normal_kl(mean, logvar, torch.zeros(()), torch.zeros(()))) / math.log(2.0)
