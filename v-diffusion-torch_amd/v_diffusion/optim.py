"""Opt-in fused optimizer tail for the reference's OWN training loop (reference train.py:158, train_utils.py:159-168,
utils.py:123-190).  The reference's ``Trainer.step`` runs ``clip_grad_norm_`` -> ``torch.optim.AdamW.step`` -> ``LambdaLR.step`` ->
``EMA.update`` -- a dozen multi-tensor passes plus three launches per parameter for the EMA -- around this package's UNet.  Two edited lines of
``train.py`` put the hot path's one-pass kernels (csrc/optim.hip: vd_sumsq, vd_adamw_ema) under that loop without touching train_utils.py:

    optimizer = v_diffusion.optim.FusedAdamW(model.parameters(), lr=lr, betas=(beta1, beta2), weight_decay=weight_decay)   # was torch.optim.AdamW(...)
    v_diffusion.optim.use_fused_ema()            # the reference Trainer's ``EMA(model, decay)`` becomes v_diffusion.optim.EMA

``FusedAdamW`` moves the parameters into ONE flat fp32 buffer (``param.data`` become views: the module, ``state_dict()``, DDP and EMA see the
same tensors as before) and gives every parameter a slot in one flat gradient buffer; the UNet's autograd nodes hand autograd fresh VIEWS of those
slots (models/unet.py::_grad_targets), which ``AccumulateGrad`` keeps instead of cloning, so ``param.grad`` already lies in the flat buffer when
``step()`` runs: one vd_adamw_ema launch updates parameters and moments (``EMA.update()`` then is one ``lerp_`` over the flat buffer).  Same
arithmetic as torch.optim.AdamW (decoupled weight decay, bias corrections, eps outside the square root's bias correction as torch does);
parameters that received no gradient (a class-conditional network called with y = None) are skipped with their own step count, as torch does.
There is no CPU path: CPU parameters raise."""
import weakref

import torch

from . import _hip


def _flat_of(p):
    return getattr(p, "_vd_flat", None)


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        params = list(params)
        if not params or any(isinstance(p, dict) for p in params):
            raise ValueError("FusedAdamW takes ONE flat list of parameters (the reference builds a single group: train.py:158)")
        if any((not p.is_cuda) or p.dtype != torch.float32 for p in params):
            raise RuntimeError("FusedAdamW: fp32 parameters on an MI355X only (the v_diffusion hot path has no CPU fallback)")
        _hip.lib()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm))
        dev = params[0].device
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4                                   # every tensor 16-byte aligned
        self._params, self._offs, self._n = params, offs, n
        self.p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                pv = self.p[o:o + p.numel()].view_as(p)
                pv.copy_(p)
                p.data = pv                                                 # the module now lives in the flat buffer
                p._vd_flat = (weakref.ref(self), o)                         # models/unet.py::_grad_targets and EMA look for this
        self.steps = 0                  # updates of the parameters that always receive gradients
        self.lag_range = None           # (lo, hi): the ONE contiguous range that may see no gradient (class embedding), with its own count
        self.skipped = 0                # updates that range sat out (its per-parameter step of torch.optim.AdamW = steps - skipped)

    @property
    def lag_steps(self):
        return self.steps - self.skipped

    def _gather_grads(self):
        """param.grad -> the flat gradient buffer (no copy for gradients that already lie in their slot); returns the index range without gradient"""
        base, missing, stray_dst, stray_src = self.g.data_ptr(), [], [], []
        for i, (p, o) in enumerate(zip(self._params, self._offs)):
            gr = p.grad
            if gr is None:
                missing.append(i)
            elif gr.data_ptr() != base + 4 * o or not gr.is_contiguous():
                stray_dst.append(self.g[o:o + p.numel()].view_as(p)); stray_src.append(gr)
        if stray_dst:
            torch._foreach_copy_(stray_dst, stray_src)
        if not missing:
            return None
        if missing != list(range(missing[0], missing[-1] + 1)):
            raise NotImplementedError("FusedAdamW: parameters without a gradient must be adjacent in parameters() order (the class-embedding "
                                      "tensors of the UNet are); freeze other parameters with requires_grad_(False) before building the optimizer")
        lo = self._offs[missing[0]]
        last = missing[-1]
        hi = self._offs[last] + (self._params[last].numel() + 3) // 4 * 4
        return (lo, min(hi, self._n))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grp = self.param_groups[0]
        lr, (b1, b2), eps, wd = float(grp["lr"]), grp["betas"], float(grp["eps"]), float(grp["weight_decay"])
        max_norm = grp.get("max_grad_norm") or 0.0
        nograd = self._gather_grads()
        if nograd is not None or self.lag_range is not None:
            rng = nograd if nograd is not None else self.lag_range
            if self.lag_range is not None and rng != self.lag_range:
                raise NotImplementedError("FusedAdamW: a second range of parameters without gradients")
            self.lag_range = rng
        self.steps += 1
        k = self.steps
        if max_norm > 0:
            _hip.sumsq(self.g, self.gnorm_sq)
        r_lo = r_hi = r_mode = 0
        r_bc1 = r_bc2 = 1.0
        if self.lag_range is not None:
            r_lo, r_hi = self.lag_range
            if nograd is not None:
                self.skipped += 1
                r_mode = 1                                                  # torch.optim.AdamW skips parameters whose .grad is None
            else:
                kl = self.lag_steps                                         # the range's own step count: updates it took part in
                r_mode, r_bc1, r_bc2 = 2, 1 - b1 ** kl, 1 - b2 ** kl
        _hip.adamw_ema(self.p, self.g, self.m, self.v, None, self.gnorm_sq if max_norm > 0 else None, float(max_norm), lr, b1, b2, eps, wd,
                       1 - b1 ** k, 1 - b2 ** k, 1.0, r_lo, r_hi, r_mode, r_bc1, r_bc2)
        return loss

    # -- torch.optim.AdamW-format state (reference checkpoints: train_utils.py:317-331 saves optimizer.state_dict())
    def state_dict(self):
        st = {}
        for i, (p, o) in enumerate(zip(self._params, self._offs)):
            lag = self.lag_range is not None and self.lag_range[0] <= o < self.lag_range[1]
            st[i] = {"step": torch.tensor(float(self.lag_steps if lag else self.steps)),
                     "exp_avg": self.m[o:o + p.numel()].view_as(p).clone(), "exp_avg_sq": self.v[o:o + p.numel()].view_as(p).clone()}
        grp = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        grp["params"] = list(range(len(self._params)))
        return {"state": st, "param_groups": [grp]}

    def load_state_dict(self, sd):
        grp = dict(sd["param_groups"][0])
        grp.pop("params", None)
        self.param_groups[0].update(grp)
        steps = set()
        lag = []
        for i, (p, o) in enumerate(zip(self._params, self._offs)):
            s = sd["state"].get(i)
            if s is None:
                continue
            self.m[o:o + p.numel()].view_as(p).copy_(s["exp_avg"]); self.v[o:o + p.numel()].view_as(p).copy_(s["exp_avg_sq"])
            steps.add(int(float(s["step"])))
        self.steps = max(steps) if steps else 0
        if len(steps) > 1:
            low = min(steps)
            for i, (p, o) in enumerate(zip(self._params, self._offs)):
                s = sd["state"].get(i)
                if s is not None and int(float(s["step"])) == low:
                    lag.append(i)
            if lag != list(range(lag[0], lag[-1] + 1)) or len(steps) > 2:
                raise NotImplementedError("FusedAdamW.load_state_dict: more than one group of lagging step counts")
            last = lag[-1]
            self.lag_range = (self._offs[lag[0]], min(self._offs[last] + (self._params[last].numel() + 3) // 4 * 4, self._n))
            self.skipped = self.steps - low


class EMA:
    """The reference's ``EMA`` (utils.py:123-190: same constructor, ``update / apply / restore``, context manager, ``state_dict`` with per-name
    ``shadow``) over ONE flat shadow buffer when the parameters live in a ``FusedAdamW`` flat buffer: ``update()`` is one ``lerp_`` launch instead of
    three launches per parameter, ``apply()`` / ``restore()`` copy one buffer instead of cloning every parameter.  Parameters that are not flat
    (no FusedAdamW, or one built later: the shadow moves over at the next ``update()``) take a multi-tensor ``lerp_``."""

    def __init__(self, model, decay=0.9999):
        self._named = [(k, v) for k, v in model.named_parameters() if v.requires_grad]
        self._refs = {k: weakref.ref(v) for k, v in self._named}
        self.decay = decay
        self.num_updates = 0
        self.backup = None
        self._opt = None
        self.shadow = None
        self._build()

    def _build(self):
        """(re)build the shadow in the layout the parameters have NOW: flat if one FusedAdamW owns them all, in order; per tensor otherwise"""
        ps = [v for _, v in self._named]
        owners = {(_flat_of(p)[0]() if _flat_of(p) else None) for p in ps}
        opt = owners.pop() if len(owners) == 1 else None
        old = self.shadow
        if opt is not None and len(opt._params) == len(ps) and all(a is b for a, b in zip(opt._params, ps)):
            self._opt = weakref.ref(opt)
            self._shadow_flat = opt.p.clone()
            self.shadow = {k: self._shadow_flat[o:o + p.numel()].view_as(p) for (k, p), o in zip(self._named, opt._offs)}
        else:
            self._opt, self._shadow_flat = None, None
            self.shadow = {k: v.detach().clone() for k, v in self._named}
        if old is not None:
            with torch.no_grad():
                for k in self.shadow:
                    self.shadow[k].copy_(old[k])

    @torch.no_grad()
    def update(self):
        if self._opt is None and all(_flat_of(p) for _, p in self._named):
            self._build()                                                  # the optimizer was built after this object: move the shadow
        self.num_updates += 1
        decay = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        opt = self._opt() if self._opt is not None else None
        if opt is not None:
            self._shadow_flat.lerp_(opt.p, 1 - decay)                      # shadow += (1 - decay) (p - shadow), utils.py:144-149
        else:
            torch._foreach_lerp_([self.shadow[k] for k, _ in self._named], [r().data for r in self._refs.values()], 1 - decay)

    @torch.no_grad()
    def apply(self):
        opt = self._opt() if self._opt is not None else None
        if opt is not None:
            self.backup = opt.p.clone()
            opt.p.copy_(self._shadow_flat)
        else:
            self.backup = {k: r().detach().clone() for k, r in self._refs.items()}
            for k, r in self._refs.items():
                r().data.copy_(self.shadow[k])

    @torch.no_grad()
    def restore(self):
        opt = self._opt() if self._opt is not None else None
        if opt is not None and torch.is_tensor(self.backup):
            opt.p.copy_(self.backup)
        else:
            for k, r in self._refs.items():
                r().data.copy_(self.backup[k])
        self.backup = None

    def __enter__(self):
        self.apply()

    def __exit__(self, *exc):
        self.restore()

    def state_dict(self):
        return {"decay": self.decay, "shadow": self.shadow, "num_updates": self.num_updates}

    @property
    def extra_states(self):
        return {"decay", "num_updates"}

    def load_state_dict(self, state_dict, strict=True):
        mine, theirs = set(self.shadow).union(self.extra_states), set(state_dict["shadow"]).union(self.extra_states)
        bad = set.symmetric_difference(mine, theirs) if strict else set.difference(mine, theirs)
        if bad:
            raise RuntimeError("Key mismatch!\n" f"Missing key(s): {', '.join(set.difference(mine, theirs))}."
                               f"Unexpected key(s): {', '.join(set.difference(theirs, mine))}")
        with torch.no_grad():
            for k, v in state_dict["shadow"].items():
                if k in self.shadow:
                    self.shadow[k].copy_(v)
        self.decay = state_dict.get("decay", self.decay)
        self.num_updates = state_dict.get("num_updates", self.num_updates)


def use_fused_ema():
    """Make the reference's ``Trainer`` (train_utils.py:130-133, resolved through VDIFF_REFERENCE_ROOT) build this module's ``EMA`` instead of its own:
    the one name is replaced in the LOADED reference modules; no reference file is touched."""
    import sys
    from . import _reference
    ref = _reference()
    for name in ("v_diffusion_ref.train_utils", "v_diffusion_ref.utils"):
        mod = sys.modules.get(name)
        if mod is not None and hasattr(mod, "EMA"):
            mod.EMA = EMA
    return ref
