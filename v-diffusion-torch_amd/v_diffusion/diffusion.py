"""Continuous-time log-SNR Gaussian diffusion on the MI355X HIP kernels, behind the call surface of the reference's
``v_diffusion/diffusion.py`` (``get_logsnr_schedule`` :42-112, ``GaussianDiffusion`` :260-576).

What runs where
  * per-element work on images (q_sample, the v/x0/eps conversions + per-sample MSE and its gradient, the whole
    reverse-step update with x0-clipping, DDIM/DDPM mean, classifier-free guidance and noise) = one fused HIP kernel
    each (csrc/diffusion.hip), instead of the reference's ~15-50 elementwise launches;
  * schedule / posterior coefficients = fp64 scalars.  In sampling they are identical for the whole batch, so the host
    computes them once per step (same formulas, same fp32 rounding points as reference :126-203) and hands 8 floats
    to the kernel; in training ``logsnr(t)`` is a (B,) fp64 torch expression (glue, not the hot path);
  * ``denoise_fn`` stays an opaque callable ``(x_t, t, y) -> model_out`` exactly as in the reference (:374,:508).

MI355X only: CPU tensors raise (the CPU restatement is oracle/diffusion_ref.py, test infrastructure).
``loss_type="kl"`` (variational-bound terms, reference :446-464) is one fused kernel pair as well (vd_bpd_terms / vd_bpd_bwd).
"""
import math

import torch
import torch.nn.functional as F

from . import _hip
from .functions import flat_mean

F64 = torch.float64



# ------------------------------------------------------------------------------------------------ small public helpers
def broadcast_to(arr, x, dtype=None, device=None, ndim=None):
    """``arr`` as a tensor shaped (-1, 1, 1, ...) with x's dtype / device / rank (reference :19-27)"""
    if x is not None:
        dtype, device, ndim = dtype or x.dtype, device or x.device, ndim or x.ndim
    return torch.as_tensor(arr, dtype=dtype, device=device).reshape((-1,) + (1,) * (ndim - 1))


def repeat_along_dim(x: torch.Tensor, repeats: int, dim: int = 0):
    """every slice along ``dim`` repeated ``repeats`` times in place: [a, b] -> [a, a, b, b] (reference :30-35)"""
    return x.repeat_interleave(repeats, dim=dim) if repeats != 1 else x.contiguous()


def slice_along_batch(x: torch.Tensor, span: int):
    """the ``span`` interleaved sub-batches x[i::span] (reference :38-39)"""
    return [x[i::span, ...] for i in range(span)]


def _per_sample(logsnr, x):
    """(B,) float32 log-SNR vector of a per-sample argument -- shape (B,) or (B,1,1,1), nothing else: a (1,C,1,1) per-channel
    tensor with C == B has B elements too, and is NOT one value per sample -- or None"""
    if not torch.is_tensor(logsnr) or not x.is_cuda or x.dtype != torch.float32 or x.ndim != 4:
        return None
    B = x.shape[0]
    if logsnr.device != x.device or tuple(logsnr.shape) not in ((B,), (B, 1, 1, 1)):
        return None
    return logsnr.reshape(-1).to(torch.float32).contiguous()


def q_sample(x_0, logsnr_t, eps=None):
    """x_t = alpha * x_0 + sigma * eps with alpha^2 = sigmoid(logsnr), sigma^2 = sigmoid(-logsnr) (reference :242-245).
    One per-sample log-SNR per image on the device, nothing asking for a gradient = the fused ``vd_q_sample`` kernel (it has no
    autograd node: the reference's q_sample is differentiable in x_0, eps and logsnr, so any input that requires grad under
    grad mode takes the tensor expression, as do all other broadcast shapes)."""
    if eps is None:
        eps = torch.randn_like(x_0)
    wants_grad = torch.is_grad_enabled() and any(torch.is_tensor(a) and a.requires_grad for a in (x_0, logsnr_t, eps))
    l = None if wants_grad else _per_sample(logsnr_t, x_0)
    if l is not None and eps.shape == x_0.shape and eps.dtype == torch.float32:
        x_0, eps = x_0.contiguous(), eps.contiguous()
        out = torch.empty_like(x_0)
        with torch.cuda.device(x_0.device):
            _hip.q_sample(x_0, eps, l, out, x_0.shape[0], x_0.shape[1], x_0[0, 0].numel())
        return out
    return x_0 * torch.sigmoid(logsnr_t).sqrt() + eps * torch.sigmoid(-logsnr_t).sqrt()


def q_mean_var(x_0, logsnr_t):
    """mean and log-variance of q(x_t | x_0) (reference :248-250)"""
    return x_0 * torch.sigmoid(logsnr_t).sqrt(), F.logsigmoid(-logsnr_t)


# conversions between the parameterisations (reference :206-239); alpha = sqrt(sigmoid(l)), sigma = sqrt(sigmoid(-l)),
# so 1/alpha = rsqrt(sigmoid(l)), sigma/alpha = exp(-l/2), alpha/sigma = exp(l/2), 1/sigma = rsqrt(sigmoid(-l))
def pred_x0_from_eps(x_t, eps, logsnr_t):
    return x_t * torch.sigmoid(logsnr_t).rsqrt() - eps * torch.exp(-0.5 * logsnr_t)


def pred_x0_from_x0eps(x_t, x0eps, logsnr_t):
    x_0, eps = x0eps.chunk(2, dim=1)
    return x_0 * torch.sigmoid(-logsnr_t) + pred_x0_from_eps(x_t, eps, logsnr_t) * torch.sigmoid(logsnr_t)


def pred_eps_from_x0(x_t, x_0, logsnr_t):
    return x_t * torch.sigmoid(-logsnr_t).rsqrt() - x_0 * torch.exp(0.5 * logsnr_t)


def pred_v_from_x0eps(x_0, eps, logsnr_t):
    return eps * torch.sigmoid(logsnr_t).sqrt() - x_0 * torch.sigmoid(-logsnr_t).sqrt()


def pred_v_from_x0(x_t, x_0, logsnr_t):
    return x_t * torch.exp(0.5 * logsnr_t) - x_0 * torch.sigmoid(-logsnr_t).rsqrt()


def pred_x0_from_v(x_t, v, logsnr_t):
    return x_t * torch.sigmoid(logsnr_t).sqrt() - v * torch.sigmoid(-logsnr_t).sqrt()


def pred_eps_from_v(x_t, v, logsnr_t):
    return x_t * torch.sigmoid(-logsnr_t).sqrt() + v * torch.sigmoid(logsnr_t).sqrt()


# ------------------------------------------------------------------------------------------------ schedules
def stable_log1mexp(x):
    """log(1 - e^x), x < 0 (reference :115-123)"""
    return torch.where(x < -9, torch.log1p(-torch.exp(x)), torch.log(-torch.expm1(x)))


def get_logsnr_schedule(schedule, logsnr_min: float = -20., logsnr_max: float = 20., rescale: bool = False):
    """Returns ``f(t) -> logsnr(t)`` (dtype of t; fp64 inside).  ``rescale`` keeps the reference's in-place rewrite of
    ``t`` (:105-109): bool -> t <- logsnr2t(logsnr), float -> t *= rescale."""
    lo, hi = float(logsnr_min), float(logsnr_max)
    if schedule == "legacy":                          # continuous version of the DDPM linear-beta schedule (:78-92)
        x_max, x_min, slope = 0.9999, 0.98, -0.0199
        c0 = x_max * math.log(x_max) - x_max

        def legacy_fn(t):
            xt = torch.lerp(torch.full_like(t, x_max), torch.full_like(t, x_min), t)
            log_alpha = 1000 / slope * (xt * torch.log(xt) - xt - c0)
            return log_alpha - stable_log1mexp(log_alpha - 1e-9)
        return legacy_fn
    if schedule == "linear":
        to_t = lambda l: torch.sigmoid(l)
        from_t = lambda u: torch.logit(u)
    elif schedule == "sigmoid":
        to_t = lambda l: (hi - l) / (hi - lo)
        from_t = lambda u: hi - u * (hi - lo)
    elif schedule == "cosine":
        to_t = lambda l: torch.atan(torch.exp(-0.5 * l)) / (0.5 * math.pi)
        from_t = lambda u: -2.0 * torch.log(torch.tan(u * (0.5 * math.pi)))
    else:
        raise NotImplementedError(schedule)
    t0 = float(to_t(torch.tensor(hi, dtype=F64)))
    t1 = float(to_t(torch.tensor(lo, dtype=F64)))

    def schedule_fn(t):
        w = t.to(F64)
        logsnr = from_t(torch.lerp(torch.full_like(w, t0), torch.full_like(w, t1), w))
        if rescale:
            if isinstance(rescale, bool):
                t.copy_(to_t(logsnr).to(t.dtype))
            elif isinstance(rescale, float):
                t.mul_(rescale)
        return logsnr.to(t.dtype)
    return schedule_fn


# ------------------------------------------------------------------------------------------------ posterior coefficients
def logsnr_to_posterior(logsnr_s, logsnr_t, var_type: str, intp_frac: float = None, x0eps_coef: bool = False):
    """E[x_s | x_t, x_0] = c1 * x_t + c2 * x_0 and log-variance, fp64 inside, fp32 out (reference :126-163)."""
    ls, lt = logsnr_s.to(F64), logsnr_t.to(F64)
    logr = lt - ls
    l1mr = stable_log1mexp(logr)
    if x0eps_coef:                                    # E[x_s | x_t] = c1 * eps + c2 * x_0 (reference :137-140)
        c1 = torch.exp(0.5 * (F.logsigmoid(ls) - lt) + logr)
        c2 = torch.sigmoid(ls).sqrt()
    else:
        c1 = torch.exp(logr + 0.5 * (F.logsigmoid(ls) - F.logsigmoid(lt)))
        c2 = torch.exp(l1mr + 0.5 * F.logsigmoid(ls))
    lv_small, lv_large = l1mr + F.logsigmoid(-ls), l1mr + F.logsigmoid(-lt)
    if var_type == "fixed_large":
        lv = lv_large
    elif var_type == "fixed_small":
        lv = lv_small
    elif var_type == "fixed_medium":
        assert isinstance(intp_frac, (float, torch.Tensor))
        lv = torch.lerp(lv_small, lv_large, intp_frac)
    else:
        raise NotImplementedError(var_type)
    return c1.float(), c2.float(), lv.float()


def logsnr_to_posterior_ddim(logsnr_s, logsnr_t, eta: float = 0., x0eps_coef: bool = False):
    """DDIM posterior (reference :169-203): eta = 0 deterministic, eta = 1 the fixed_small DDPM posterior, 0 < eta < 1
    in between.  Quirk kept from the reference (:180-182): with ``eta = 0`` and ``x0eps_coef`` it returns the LOGARITHMS
    of (sigma_s, alpha_s) -- the exp is missing there -- and its sampler multiplies with those values."""
    ls, lt = logsnr_s.to(F64), logsnr_t.to(F64)
    if eta == 1.:
        return logsnr_to_posterior(ls, lt, "fixed_small")
    logr = lt - ls
    if eta == 0:
        if x0eps_coef:
            c1, c2 = 0.5 * F.logsigmoid(-ls), 0.5 * F.logsigmoid(ls)
        else:
            c1 = torch.exp(0.5 * (F.logsigmoid(-ls) - F.logsigmoid(-lt)))
            c2 = torch.exp(stable_log1mexp(0.5 * logr) + 0.5 * F.logsigmoid(ls))
        return c1.float(), c2.float(), torch.as_tensor(-math.inf)
    l1mr = stable_log1mexp(logr)
    logvar = l1mr + F.logsigmoid(-ls) + 2 * math.log(eta)
    w = stable_log1mexp(2 * math.log(eta) + l1mr)
    if x0eps_coef:
        c1, c2 = 0.5 * (w + F.logsigmoid(-ls)), 0.5 * F.logsigmoid(ls)
    else:
        c1 = 0.5 * (w + F.logsigmoid(-ls) - F.logsigmoid(-lt))
        c2 = stable_log1mexp(0.5 * (logr + w)) + 0.5 * F.logsigmoid(ls)
    return c1.exp().float(), c2.exp().float(), logvar.float()


def _pred_coefs(model_out_type, lt32):
    """(a0, b0x, b0e): x0_hat = a0 * x_t + b0x * out (+ b0e * out_eps), fp32 as the reference evaluates :206-239."""
    l = torch.as_tensor(lt32, dtype=torch.float32)
    s1, s0 = torch.sigmoid(l), torch.sigmoid(-l)
    if model_out_type == "v":
        return float(s1.sqrt()), float(-s0.sqrt()), 0.0
    if model_out_type == "x0":
        return 0.0, 1.0, 0.0
    if model_out_type == "eps":
        return float(s1.rsqrt()), float(-torch.exp(-0.5 * l)), 0.0
    if model_out_type == "both":
        return float(s1.rsqrt() * s1), float(s0), float(-torch.exp(-0.5 * l) * s1)
    raise NotImplementedError(model_out_type)


def _device_ctx(device):
    """make ``device`` the current HIP device for the duration of a sampler call (kernels use its current stream)"""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("GaussianDiffusion sampling: device must be an MI355X ('cuda'); there is no CPU path")
    return torch.cuda.device(device)


def _need_cuda(x, what):
    if not x.is_cuda:
        raise RuntimeError(f"GaussianDiffusion.{what}: tensors must live on an MI355X (no CPU path; see oracle/ for the CPU restatement)")


class _MSELoss(torch.autograd.Function):
    """Per-sample weighted MSE of reference :520-541 with its analytic gradient wrt the network output."""

    @staticmethod
    def forward(ctx, out, x0, eps, xt, logsnr32, mot, rw):
        B, C = x0.shape[:2]
        HW = x0[0, 0].numel()
        out = out.contiguous()
        loss = torch.empty((B,), dtype=torch.float32, device=x0.device)
        aux = torch.empty((B, 2), dtype=torch.float32, device=x0.device)
        with torch.cuda.device(x0.device):
            _hip.loss_fwd(x0, eps, xt, out, logsnr32, mot, rw, loss, aux, B, C, HW)
        ctx.save_for_backward(out, x0, eps, xt, logsnr32, aux)
        ctx.cfg = (mot, rw, B, C, HW)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        out, x0, eps, xt, logsnr32, aux = ctx.saved_tensors
        mot, rw, B, C, HW = ctx.cfg
        dout = torch.empty_like(out)
        with torch.cuda.device(x0.device):
            _hip.loss_bwd(x0, eps, xt, out, logsnr32, aux, gloss.to(torch.float32).contiguous(), mot, rw, dout, B, C, HW)
        return dout, None, None, None, None, None, None


class _KLLoss(torch.autograd.Function):
    """where(s > 0, KL, decoder NLL) per sample in bits/dim (reference :510-515) with its analytic gradient wrt the output."""

    @staticmethod
    def forward(ctx, out, x0, xt, coef, use_kl, mot):
        B, C = x0.shape[:2]
        HW = x0[0, 0].numel()
        out = out.contiguous()
        kl = torch.empty((B,), dtype=torch.float32, device=x0.device)
        nll = torch.empty_like(kl)
        with torch.cuda.device(x0.device):
            _hip.bpd_terms(x0, xt, out, coef, mot, False, kl, nll, None, None, B, C, HW)
        ctx.save_for_backward(out, x0, xt, coef, use_kl)
        ctx.cfg = (mot, B, C, HW)
        return torch.where(use_kl != 0, kl, nll)

    @staticmethod
    def backward(ctx, gloss):
        out, x0, xt, coef, use_kl = ctx.saved_tensors
        mot, B, C, HW = ctx.cfg
        dout = torch.empty_like(out)
        with torch.cuda.device(x0.device):
            _hip.bpd_bwd(x0, xt, out, coef, use_kl, gloss.to(torch.float32).contiguous(), mot, False, dout, B, C, HW)
        return dout, None, None, None, None, None


class PendingImages:
    """uint8 HWC images on their way to pinned host memory (see ``GaussianDiffusion.p_sample_uint8_async``)."""
    _side = {}

    def __init__(self, u8_dev):
        dev = u8_dev.device
        side = PendingImages._side.get(dev)
        if side is None:
            side = PendingImages._side[dev] = torch.cuda.Stream(dev)
        self.host = torch.empty(u8_dev.shape, dtype=torch.uint8, pin_memory=True)
        self.done = torch.cuda.Event()
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.host.copy_(u8_dev, non_blocking=True)
            self.done.record(side)
        u8_dev.record_stream(side)

    def ready(self):
        return self.done.query()

    def tensor(self):
        self.done.synchronize()
        return self.host

    def numpy(self):
        return self.tensor().numpy()


class GaussianDiffusion:
    def __init__(self, logsnr_fn, sample_timesteps, model_out_type, model_var_type, reweight_type, loss_type,
                 intp_frac=None, w_guide=0.1, p_uncond=0.1, x0eps_coef=False):
        self.logsnr_fn = logsnr_fn
        self.sample_timesteps = sample_timesteps
        self.model_out_type = model_out_type
        self.model_var_type = model_var_type
        self.reweight_type = reweight_type
        self.loss_type = loss_type
        self.intp_frac = intp_frac
        self.w_guide = w_guide
        self.p_uncond = p_uncond
        self.x0eps_coef = x0eps_coef

    def t2logsnr(self, *ts, x=None):
        """logsnr(t) reshaped to (B,1,1,..) in x's dtype (reference :293-295)"""
        def one(t):
            l = self.logsnr_fn(t)
            if x is not None:
                return l.to(dtype=x.dtype, device=x.device).reshape((-1,) + (1,) * (x.ndim - 1))
            return l.reshape(-1, 1, 1, 1)
        return tuple(map(one, ts))

    # ------------------------------------------------------------------------------------------ per-sample posterior API
    # (reference :297-356).  These keep the reference's tensor-valued log-SNR arguments (one value per sample) and are
    # NOT what the samplers run -- p_sample / p_sample_step use the fused vd_sample_step kernel with host-computed
    # coefficients.  They are small compositions of device tensor ops kept for callers of the reference API.
    def q_posterior_mean_var(self, x_0, x_t, logsnr_s, logsnr_t, model_var_type=None, intp_frac=None):
        c1, c2, logvar = logsnr_to_posterior(logsnr_s, logsnr_t, var_type=model_var_type or self.model_var_type,
                                             intp_frac=self.intp_frac if intp_frac is None else intp_frac,
                                             x0eps_coef=self.x0eps_coef)
        return c1 * x_t + c2 * x_0, logvar

    def q_posterior_mean_var_ddim(self, x_0, x_t, logsnr_s, logsnr_t):
        c1, c2, logvar = logsnr_to_posterior_ddim(logsnr_s, logsnr_t, eta=0., x0eps_coef=self.x0eps_coef)
        return c1 * x_t + c2 * x_0, logvar

    def p_mean_var(self, model_out, x_t, logsnr_s, logsnr_t, clip_denoised, return_pred, use_ddim=False):
        if self.model_var_type == "learned":
            raise NotImplementedError("model_var_type='learned' is not supported (the reference asserts the same in train_loss)")
        l = logsnr_t
        s1, s0 = torch.sigmoid(l), torch.sigmoid(-l)
        if self.model_out_type == "x0":
            pred = model_out
        elif self.model_out_type == "eps":
            pred = x_t * s1.rsqrt() - model_out * torch.exp(-0.5 * l)
        elif self.model_out_type == "v":
            pred = x_t * s1.sqrt() - model_out * s0.sqrt()
        elif self.model_out_type == "both":
            x0p, epsp = model_out.chunk(2, dim=1)
            pred = x0p * s0 + (x_t * s1.rsqrt() - epsp * torch.exp(-0.5 * l)) * s1
        else:
            raise NotImplementedError(self.model_out_type)
        if clip_denoised:
            pred = pred.clamp(-1., 1.)
        if self.x0eps_coef:                    # posterior over (eps, x0): eps re-derived from the clipped x0 (reference :338-347)
            if clip_denoised or self.model_out_type != "eps":
                x_t = (x_t - s1.sqrt() * pred) * s0.rsqrt()
            else:
                x_t = model_out
        if use_ddim:
            mean, logvar = self.q_posterior_mean_var_ddim(pred, x_t, logsnr_s, logsnr_t)
        else:
            mean, logvar = self.q_posterior_mean_var(pred, x_t, logsnr_s, logsnr_t)
        return (mean, logvar, pred) if return_pred else (mean, logvar)

    # ------------------------------------------------------------------------------------------ training
    def from_model_out_to_pred(self, x_t, model_out, logsnr_t):
        """{reweight_type: regression target counterpart} of a network output (reference :466-490): x0 for "constant",
        eps for "snr", (x0, eps) for "snr_trunc", v for "snr_1plus".  ``train_loss`` does not call this (its fused kernel
        evaluates the same conversions per element); kept for callers of the reference API."""
        mot = self.model_out_type
        if mot == "v":
            v = model_out
            x_0, eps = pred_x0_from_v(x_t, v, logsnr_t), pred_eps_from_v(x_t, v, logsnr_t)
        else:
            if mot == "x0":
                x_0 = model_out
                eps = pred_eps_from_x0(x_t, x_0, logsnr_t)
            elif mot == "eps":
                eps = model_out
                x_0 = pred_x0_from_eps(x_t, eps, logsnr_t)
            elif mot == "both":
                x_0 = pred_x0_from_x0eps(x_t, model_out, logsnr_t)
                eps = pred_eps_from_x0(x_t, x_0, logsnr_t)
            else:
                raise NotImplementedError(mot)
            v = pred_v_from_x0eps(x_0, eps, logsnr_t)
        return {"constant": x_0, "snr": eps, "snr_trunc": (x_0, eps), "snr_1plus": v}

    def train_loss(self, denoise_fn, x_0, t, y, noise=None):
        """Per-sample loss (B,) -- reference :492-545, mse branch.  ``y`` is mutated in place by the label drop
        (after the forward, reference quirk :527-529), which consumes one ``torch.rand(B)`` of the global CPU RNG."""
        _need_cuda(x_0, "train_loss")
        if self.loss_type == "kl":
            return self._train_loss_kl(denoise_fn, x_0, t, y, noise)
        if self.loss_type != "mse":
            raise NotImplementedError(self.loss_type)
        assert self.model_var_type != "learned"
        assert self.reweight_type in _hip.REWEIGHTS and self.model_out_type in _hip.OUT_TYPES
        if noise is None:
            noise = torch.randn_like(x_0)
        x_0 = x_0.to(torch.float32).contiguous()
        noise = noise.to(torch.float32).contiguous()
        B, C = x_0.shape[:2]
        HW = x_0[0, 0].numel()
        logsnr32 = self.logsnr_fn(t).to(torch.float32).reshape(-1).contiguous()
        x_t = torch.empty_like(x_0)
        with torch.cuda.device(x_0.device):
            _hip.q_sample(x_0, noise, logsnr32, x_t, B, C, HW)
        model_out = denoise_fn(x_t, t, y)
        if self.p_uncond and y is not None:
            keep = (torch.rand((y.shape[0],)) > self.p_uncond).to(device=y.device, dtype=y.dtype)
            y *= keep.reshape((-1,) + (1,) * (y.ndim - 1))
        mot, rw = _hip.OUT_TYPES[self.model_out_type], _hip.REWEIGHTS[self.reweight_type]
        if self.reweight_type != "snr_trunc" and model_out.shape != x_0.shape:
            raise RuntimeError(f"the size of target {tuple(x_0.shape)} must match model_out {tuple(model_out.shape)}")
        return _MSELoss.apply(model_out.to(torch.float32), x_0, noise, x_t, logsnr32, mot, rw)


    # ------------------------------------------------------------------------------------------ variational bound (loss_type "kl")
    def _bpd_coefs(self, logsnr_s, logsnr_t):
        """coef[B][8] of vd_bpd_terms from per-sample (B,) log-SNRs: fp64 posterior arithmetic on the (tiny) vectors,
        fp32 prediction weights as reference :206-239 evaluates them."""
        if self.model_var_type == "learned":
            raise NotImplementedError("model_var_type='learned'")
        if self.x0eps_coef:
            raise NotImplementedError("x0eps_coef=True with the bits-per-dim terms")
        ls, lt = logsnr_s.reshape(-1).to(torch.float32), logsnr_t.reshape(-1).to(torch.float32)
        c1, c2, tlv = logsnr_to_posterior(ls, lt, "fixed_small")
        _, _, mlv = logsnr_to_posterior(ls, lt, self.model_var_type, self.intp_frac)
        s1, s0 = torch.sigmoid(lt), torch.sigmoid(-lt)
        z = torch.zeros_like(lt)
        if self.model_out_type == "v":
            a0, b0x, b0e = s1.sqrt(), -s0.sqrt(), z
        elif self.model_out_type == "x0":
            a0, b0x, b0e = z, torch.ones_like(lt), z
        elif self.model_out_type == "eps":
            a0, b0x, b0e = s1.rsqrt(), -torch.exp(-0.5 * lt), z
        elif self.model_out_type == "both":
            a0, b0x, b0e = s1.rsqrt() * s1, s0, -torch.exp(-0.5 * lt) * s1
        else:
            raise NotImplementedError(self.model_out_type)
        return torch.stack([a0, b0x, b0e, c1, c2, tlv, mlv, z], dim=1).contiguous()

    def _loss_term_bpd(self, model_out, x_0, x_t, logsnr_s, logsnr_t, clip_denoised, return_pred=False):
        """(kl, decoder_nll[, pred_x_0]) per sample in bits per dimension -- reference :446-464, one fused kernel."""
        _need_cuda(x_0, "_loss_term_bpd")
        x_0, x_t = x_0.to(torch.float32).contiguous(), x_t.to(torch.float32).contiguous()
        out = model_out.to(torch.float32).contiguous()
        B, C = x_0.shape[:2]
        HW = x_0[0, 0].numel()
        coef = self._bpd_coefs(logsnr_s, logsnr_t)
        kl = torch.empty((B,), dtype=torch.float32, device=x_0.device)
        nll = torch.empty_like(kl)
        pred = torch.empty_like(x_0) if return_pred else None
        with torch.cuda.device(x_0.device):
            _hip.bpd_terms(x_0, x_t, out, coef, _hip.OUT_TYPES[self.model_out_type], clip_denoised, kl, nll, pred, None, B, C, HW)
        return (kl, nll, pred) if return_pred else (kl, nll)

    def _train_loss_kl(self, denoise_fn, x_0, t, y, noise):
        """un-weighted bound term per sample (reference :497-515): t snapped up to the sampling grid, s = t - 1/T,
        KL(q || p) where s > 0 and the discretised decoder NLL for the last step."""
        if noise is None:
            noise = torch.randn_like(x_0)
        x_0 = x_0.to(torch.float32).contiguous()
        noise = noise.to(torch.float32).contiguous()
        B, C = x_0.shape[:2]
        HW = x_0[0, 0].numel()
        T = self.sample_timesteps
        t = torch.ceil(t * T).div(T)
        s = t.sub(1 / T).clamp(min=0.)
        use_kl = (s != 0).to(torch.float32)
        lt32 = self.logsnr_fn(t).to(torch.float32).reshape(-1).contiguous()
        ls32 = self.logsnr_fn(s).to(torch.float32).reshape(-1)
        x_t = torch.empty_like(x_0)
        with torch.cuda.device(x_0.device):
            _hip.q_sample(x_0, noise, lt32, x_t, B, C, HW)
        model_out = denoise_fn(x_t, t, y)
        coef = self._bpd_coefs(ls32, lt32)
        return _KLLoss.apply(model_out.to(torch.float32), x_0, x_t, coef, use_kl, _hip.OUT_TYPES[self.model_out_type])

    def _prior_bpd(self, x_0):
        """KL(q(x_1 | x_0) || N(0, I)) per sample in bits/dim.  Reference :547-553 (its ``logsnr_t, = ...`` only unpacks for
        a batch of one; this is the same expression for any batch).  A (B,)-vector reduction of tensor ops, not a kernel."""
        B = x_0.shape[0]
        lt = self.logsnr_fn(torch.ones((B,), dtype=torch.float32, device=x_0.device)).reshape((-1,) + (1,) * (x_0.ndim - 1))
        mean, logvar = x_0 * torch.sigmoid(lt).sqrt(), F.logsigmoid(-lt)
        kl = 0.5 * (-1.0 - logvar + mean.pow(2) + torch.exp(logvar))
        return flat_mean(kl) / math.log(2.)

    def calc_all_bpd(self, denoise_fn, x_0, y, clip_denoised=True, generator=None):
        """Bits-per-dimension decomposition over all sampling steps: (total_bpd (B,), terms (B,T), prior (B,), mse (B,T)),
        all on x_0's device.  Reference :555-576 states this loop but cannot run as written (it unpacks ``x_0.shape`` into
        the batch size and the 3-tuple of ``_loss_term_bpd`` into two names); the evident intent -- L_0 = decoder NLL,
        L_i = KL for i > 0, as the commented ``torch.where(s > 0, kl, decoder_nll)`` at :462 says -- is implemented."""
        _need_cuda(x_0, "calc_all_bpd")
        x_0 = x_0.to(torch.float32).contiguous()
        B, C = x_0.shape[:2]
        HW = x_0[0, 0].numel()
        T = self.sample_timesteps
        dev = x_0.device
        loss = torch.zeros((B, T), dtype=torch.float32, device=dev)
        mse = torch.zeros((B, T), dtype=torch.float32, device=dev)
        kl, nll, ms = (torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(3))
        x_t = torch.empty_like(x_0)
        mot = _hip.OUT_TYPES[self.model_out_type]
        with torch.cuda.device(dev):
            for i in range(T - 1, -1, -1):
                s = torch.full((B,), i / T, dtype=F64, device=dev)
                t = torch.full((B,), (i + 1) / T, dtype=F64, device=dev)
                ls32, lt32 = self.logsnr_fn(s).to(torch.float32), self.logsnr_fn(t).to(torch.float32)
                noise = torch.empty_like(x_0).normal_(generator=generator)
                _hip.q_sample(x_0, noise, lt32.contiguous(), x_t, B, C, HW)
                out = denoise_fn(x_t, t, y).to(torch.float32).contiguous()
                _hip.bpd_terms(x_0, x_t, out, self._bpd_coefs(ls32, lt32), mot, clip_denoised, kl, nll, None, ms, B, C, HW)
                loss[:, i] = kl if i > 0 else nll
                mse[:, i] = ms
        prior = self._prior_bpd(x_0)
        return loss.sum(dim=1) + prior, loss, prior, mse

    # ------------------------------------------------------------------------------------------ sampling
    def _step_coefs(self, step, use_ddim):
        """(the 8 floats of vd_sample_step for reverse step ``step`` (python int), the time the network is called with)."""
        T = self.sample_timesteps
        st = torch.tensor([step / T, (step + 1) / T], dtype=F64)
        l = self.logsnr_fn(st)                                 # a rescaling schedule rewrites st in place (:105-109)
        t_net = float(st[1])                                   # ... and the network is called with THAT t (:363-374)
        ls32, lt32 = l[0:1].float(), l[1:2].float()            # cast to the image dtype before the posterior (:365)
        if use_ddim:
            c1, c2, lv = logsnr_to_posterior_ddim(ls32, lt32, eta=0., x0eps_coef=self.x0eps_coef)
        else:
            c1, c2, lv = logsnr_to_posterior(ls32, lt32, self.model_var_type, self.intp_frac, x0eps_coef=self.x0eps_coef)
        a0, b0x, b0e = _pred_coefs(self.model_out_type, lt32[0])
        nscale = float(torch.exp(0.5 * lv.float().reshape(-1)[0])) if step > 0 else 0.0
        c1, c2 = float(c1.reshape(-1)[0]), float(c2.reshape(-1)[0])
        if self.x0eps_coef:
            # mean = c1*eps + c2*x0_hat with eps = (x_t - alpha*x0_hat)/sigma (reference :338-347; for an eps-network without
            # clipping eps is the raw output, the same number): folded on the host into weights of (x_t, x0_hat)
            l = lt32[0].double()
            alpha, rsig = float(torch.sigmoid(l).sqrt()), float(torch.sigmoid(-l).rsqrt())
            c1, c2 = c1 * rsig, c2 - c1 * alpha * rsig
        return [a0, b0x, b0e, c1, c2, nscale, float(self.w_guide), 0.0], t_net

    def _use_cfg(self, y):
        return (self.w_guide > 0) and (y is not None)

    def _reverse_step(self, denoise_fn, x_t, x_in, y_in, step, cfg, noise, use_ddim, clip, want_pred, x_next, x_dup):
        B, C = x_t.shape[:2]
        HW = x_t[0, 0].numel()
        k8, t_net = self._step_coefs(step, use_ddim)
        t_in = torch.full((x_in.shape[0],), t_net, dtype=F64, device=x_t.device)
        out = denoise_fn(x_in, t_in, y_in).to(torch.float32).contiguous()
        mot = _hip.OUT_TYPES[self.model_out_type]
        pred = None
        if want_pred:                                          # guided x0 prediction = the step-0 rule without noise
            pred = torch.empty_like(x_t)
            kp = list(k8)
            kp[5] = 0.0
            _hip.sample_step(x_t, out, None, kp, mot, cfg, True, clip, pred, None, B, C, HW)
        _hip.sample_step(x_t, out, noise, k8, mot, cfg, step == 0, clip, x_next, x_dup, B, C, HW)
        return pred

    def p_sample_step(self, denoise_fn, x_t, step, y, generator=None, clip_denoised=True, return_pred=False, use_ddim=False):
        """One reverse step (reference :360-392).  ``step`` is the (B,) tensor the reference passes; all entries must be
        equal (they are in ``p_sample``), which lets the host pre-compute the fp64 coefficients."""
        _need_cuda(x_t, "p_sample_step")
        step = torch.as_tensor(step, device=x_t.device).reshape(-1)
        ti = int(step[0].item())
        if not bool((step == ti).all()):
            return self._p_sample_step_per_sample(denoise_fn, x_t, step, y, generator, clip_denoised, return_pred, use_ddim)
        x_t = x_t.to(torch.float32).contiguous()
        cfg = self._use_cfg(y)
        if cfg:
            x_in = x_t.repeat_interleave(2, dim=0)
            y_in = y.repeat_interleave(2, dim=0).clone()
            y_in[1::2] = 0
        else:
            x_in, y_in = x_t, y
        noise = torch.empty_like(x_t).normal_(generator=generator)
        x_next = torch.empty_like(x_t)
        pred = self._reverse_step(denoise_fn, x_t, x_in, y_in, ti, cfg, noise, use_ddim, clip_denoised, return_pred, x_next, None)
        return (x_next, pred) if return_pred else x_next

    def _p_sample_step_per_sample(self, denoise_fn, x_t, step, y, generator, clip_denoised, return_pred, use_ddim):
        """A (B,) step tensor with DIFFERENT entries (reference :360-392 accepts one): every sample has its own log-SNR pair,
        so the host cannot hand the fused kernel one coefficient set; the step is composed from the per-sample posterior
        API (``p_mean_var`` on (B,1,1,1) log-SNR tensors, device tensor ops) around the same HIP UNet forward.  The
        samplers never take this branch (their step index is uniform)."""
        T = self.sample_timesteps
        x_t = x_t.to(torch.float32)
        s, t = step.to(F64) / T, (step.to(F64) + 1) / T
        logsnr_s, logsnr_t = self.t2logsnr(s, t, x=x_t)          # may rescale s, t in place, as the reference does
        cond = (step > 0).reshape((-1,) + (1,) * (x_t.ndim - 1))
        cfg = self._use_cfg(y)
        rep = (lambda a: repeat_along_dim(a, 2)) if cfg else (lambda a: a)
        x_in, t_in, y_in = rep(x_t), rep(t), (rep(y) if y is not None else None)
        if cfg:
            y_in = y_in.clone()
            y_in[1::2] = 0
        out = denoise_fn(x_in, t_in, y_in)
        mean, logvar, pred = self.p_mean_var(out, x_in, rep(logsnr_s), rep(logsnr_t), clip_denoised=clip_denoised,
                                             return_pred=True, use_ddim=use_ddim)
        mean = torch.where(rep(cond), mean, pred)
        if cfg:
            mean, mean_u = slice_along_batch(mean, 2)
            pred, pred_u = slice_along_batch(pred, 2)
            mean = mean + self.w_guide * (mean - mean_u)
            pred = pred + self.w_guide * (pred - pred_u)
            if logvar.ndim > 0:
                logvar = logvar[0::2]
        noise = torch.empty_like(mean).normal_(generator=generator)
        sample = mean + cond.to(mean.dtype) * torch.exp(0.5 * logvar) * noise
        return (sample, pred) if return_pred else sample

    # ---- HIP-graph sampler: the whole reverse step -- UNet forward + fused update -- captured once
    # into a HIP graph and replayed; per-step scalars live in a device buffer (k8), the state is updated in place, and
    # only the noise draw / t fill / coefficient copy stay eager.  Measured on MI355X (tests/perf_sample.py): replay and
    # eager take the same time at every batch size (8.0 ms/step at batch 1, 11 ms at 8, 25.5 ms at 32) -- even the
    # batch-1 step is bound by the per-kernel critical path on the GPU (one 3x3 conv = 72 dependent K tiles), not by the
    # ~250 host launches -- so the graph is opt-in (``use_graph=True``) until small-batch kernels get split-K.
    GRAPH_MAX_ROWS = 0
    GRAPH_CACHE_MAX = 8

    def _graph_eligible(self, denoise_fn, rows, pred_freq):
        net = getattr(denoise_fn, "module", denoise_fn)
        return hasattr(net, "engine") and not net.training and rows <= self.GRAPH_MAX_ROWS and pred_freq is None

    def _sample_loop_graph(self, denoise_fn, shape, x_t, y_in, cfg, device, generator, use_ddim):
        B, C = shape[0], shape[1]
        HW = int(shape[2]) * int(shape[3])
        T = self.sample_timesteps
        net = getattr(denoise_fn, "module", denoise_fn)
        mot = _hip.OUT_TYPES[self.model_out_type]
        key = (id(denoise_fn), tuple(shape), bool(cfg), self.model_out_type, None if y_in is None else tuple(y_in.shape),
               hash(tuple(p.data_ptr() for p in net.parameters())))
        cache = self.__dict__.setdefault("_graphs", {})
        entry = cache.get(key)
        if entry is None:
            rows = B * (1 + cfg)
            st = dict(x=torch.zeros(shape, dtype=torch.float32, device=device),
                      t=torch.zeros((rows,), dtype=F64, device=device),
                      noise=torch.zeros(shape, dtype=torch.float32, device=device),
                      k=torch.zeros((8,), dtype=torch.float32, device=device),
                      y=None if y_in is None else torch.zeros_like(y_in))
            st["xin"] = torch.zeros((rows,) + tuple(shape[1:]), dtype=torch.float32, device=device) if cfg else st["x"]

            def body():
                out = denoise_fn(st["xin"], st["t"], st["y"]).to(torch.float32).contiguous()
                _hip.sample_step(st["x"], out, st["noise"], None, mot, cfg, False, True, st["x"], st["xin"] if cfg else None,
                                 B, C, HW, k_dev=st["k"])
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):              # warm-up outside capture (lazy initialisation, workspace growth)
                body()
            torch.cuda.current_stream(device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                body()
            # the graph's pack / convolution nodes hold raw pointers into the engine's pack state (device table, U images):
            # keep that state referenced by the cache entry, so an engine-side eviction can never free memory a cached graph
            # replays through; the graph cache itself is a small LRU (a graph pins its activations)
            eng = net.engine() if hasattr(net, "engine") else None
            while len(cache) >= self.GRAPH_CACHE_MAX:
                cache.pop(next(iter(cache)))
            entry = cache[key] = (graph, st, None if eng is None else eng._last_pack_state)
        else:
            cache[key] = cache.pop(key)                # most recently used last
        graph, st = entry[0], entry[1]
        st["x"].copy_(x_t)
        if cfg:
            st["xin"].copy_(x_t.repeat_interleave(2, dim=0))
        if y_in is not None:
            st["y"].copy_(y_in)
        rowsk, tnets = [], []
        for ti in range(T):
            k8, t_net = self._step_coefs(ti, use_ddim)
            if ti == 0:                                # last step returns the x0 prediction: mean = 0*x_t + 1*x0_hat, no noise
                k8[3], k8[4], k8[5] = 0.0, 1.0, 0.0
            rowsk.append(k8)
            tnets.append(t_net)
        ktab = torch.tensor(rowsk, dtype=torch.float32).to(device)
        for ti in reversed(range(T)):
            st["t"].fill_(tnets[ti])
            st["noise"].normal_(generator=generator)
            st["k"].copy_(ktab[ti])
            graph.replay()
        return st["x"].clone()

    def _sample_loop(self, denoise_fn, shape, noise, label, device, seed, use_ddim, pred_freq=None, use_graph=None):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("GaussianDiffusion.p_sample: device must be an MI355X ('cuda'); there is no CPU path")
        B, T = shape[0], self.sample_timesteps
        generator = None if seed is None else torch.Generator(device).manual_seed(seed)
        if noise is None:
            x_t = torch.randn(shape, device=device, generator=generator)
        else:
            x_t = noise.to(device=device, dtype=torch.float32).contiguous().clone()
        if label is not None:
            label = label.to(device)
        cfg = self._use_cfg(label)
        if cfg:
            y_in = label.repeat_interleave(2, dim=0).clone()
            y_in[1::2] = 0                                     # unconditional rows (reference :372)
            x_in = x_t.repeat_interleave(2, dim=0)
            x_in_next = torch.empty_like(x_in)
        else:
            y_in, x_in, x_in_next = label, x_t, None
        if use_graph is None:
            use_graph = self._graph_eligible(denoise_fn, B * (1 + cfg), pred_freq)
        if use_graph:
            return self._sample_loop_graph(denoise_fn, tuple(shape), x_t, y_in, cfg, device, generator, use_ddim), []
        x_next = torch.empty_like(x_t)
        preds = []
        net = getattr(denoise_fn, "module", denoise_fn)
        eng = net.engine() if hasattr(net, "engine") else None
        if eng is not None:
            eng.pack_cache = {}                                # the weights do not change inside one reverse chain
        try:
            return self._eager_chain(denoise_fn, x_t, x_in, x_in_next, x_next, y_in, cfg, B, T, device, generator, use_ddim,
                                     pred_freq, preds)
        finally:
            if eng is not None:
                eng.pack_cache = None

    def _eager_chain(self, denoise_fn, x_t, x_in, x_in_next, x_next, y_in, cfg, B, T, device, generator, use_ddim, pred_freq, preds):
        for ti in reversed(range(T)):
            step_noise = torch.empty_like(x_t).normal_(generator=generator)      # drawn every step, also for DDIM (:389)
            want = pred_freq is not None and (ti + 1) % pred_freq == 0
            pred = self._reverse_step(denoise_fn, x_t, x_in if cfg else x_t, y_in, ti, cfg, step_noise, use_ddim, True,
                                      want, x_next, x_in_next)
            if want:
                preds.append(pred.cpu())
            x_t, x_next = x_next, x_t
            if cfg:
                x_in, x_in_next = x_in_next, x_in
        return x_t, preds

    @staticmethod
    def _default_device(denoise_fn):
        """device of the network's parameters (the reference defaults to "cpu", :398, which cannot run here)"""
        net = getattr(denoise_fn, "module", denoise_fn)
        if isinstance(net, torch.nn.Module):
            for q in net.parameters():
                return q.device
        return torch.device("cuda")

    @torch.inference_mode()
    def p_sample(self, denoise_fn, shape, noise=None, label=None, device=None, seed=None, use_ddim=False, use_graph=None):
        """Full reverse chain (reference :394-414); returns a CPU tensor like the reference.  ``device=None`` (default)
        = the device of ``denoise_fn``'s parameters (documented deviation: the reference's default "cpu" has no path here).
        ``use_graph`` (extension): True replays one captured HIP graph per reverse step (see _sample_loop_graph)."""
        device = self._default_device(denoise_fn) if device is None else device
        with _device_ctx(device):
            x, _ = self._sample_loop(denoise_fn, tuple(shape), noise, label, device, seed, use_ddim, use_graph=use_graph)
        return x.cpu()

    @torch.inference_mode()
    def p_sample_uint8_async(self, denoise_fn, shape, noise=None, label=None, device=None, seed=None, use_ddim=False,
                             use_graph=None):
        """The sampler plumbing of reference generate.py:143-150 without its host round trips: the chain
        runs as in ``p_sample``; the result is quantised to uint8 and packed NCHW -> NHWC on the device
        (``(x*127.5+127.5).clamp(0,255).to(uint8).permute(0,2,3,1)`` as one kernel, 4x fewer bytes over PCIe) and copied
        into pinned host memory on a side stream.  Returns a ``PendingImages``: the call does not block, so the next
        batch's chain is enqueued while this copy drains; ``.numpy()`` waits for the copy only."""
        device = torch.device(self._default_device(denoise_fn) if device is None else device)
        with _device_ctx(device):
            x, _ = self._sample_loop(denoise_fn, tuple(shape), noise, label, device, seed, use_ddim, use_graph=use_graph)
            B, C, Hh, Ww = x.shape
            u8 = torch.empty((B, Hh, Ww, C), dtype=torch.uint8, device=device)
            _hip.images_to_uint8_hwc(x.contiguous(), u8, B, C, Hh * Ww)
            return PendingImages(u8)

    @torch.inference_mode()
    def p_sample_progressive(self, denoise_fn, shape, noise=None, label=None, device=None, seed=None, use_ddim=False,
                             pred_freq=50):
        """reference :416-441: also returns the x0 predictions every ``pred_freq`` steps (earliest step first)."""
        device = self._default_device(denoise_fn) if device is None else device
        with _device_ctx(device):
            x, preds = self._sample_loop(denoise_fn, tuple(shape), noise, label, device, seed, use_ddim, pred_freq)
        L = self.sample_timesteps // pred_freq
        out = torch.zeros((L, shape[0]) + tuple(shape[1:]), dtype=torch.float32)
        for i, p in enumerate(preds):                          # preds were collected from the last index down
            out[L - 1 - i] = p
        return x.cpu(), out
