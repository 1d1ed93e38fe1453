"""Train-step driver for the hot path: the sequence of reference ``Trainer.loss`` / ``Trainer.step``
(v_diffusion/train_utils.py:137-169) on flat HBM buffers.

    t ~ U(0,1) fp64, noise ~ N(0,1) from a per-rank generator (seed 8191 + rank)     train_utils.py:124,140-146
    loss = diffusion.train_loss(model, x, t, y, noise).mean() / num_accum ; backward   :147-154
    gradient mean over ranks (what DDP does for the reference, train.py:148)           RCCL all-reduce, bucketed
    clip_grad_norm_(max_norm) -> AdamW -> LR warm-up -> EMA                             :159-168, utils.py:144-149

MI355X-first choices
  * parameters, gradients, Adam moments and the EMA shadow are each ONE contiguous fp32 buffer (a 60.8 M-parameter CIFAR
    model is 243 MB per buffer; 288 GB of HBM makes five copies a non-issue).  ``param.data`` / ``param.grad`` are views,
    so state_dict / checkpoints keep the reference's per-tensor layout.
  * the weight-gradient kernels write straight into the flat gradient buffer (no per-tensor .grad allocation, no bucket
    copy); it is laid out in the order the backward pass COMPLETES tensors, so ready gradients always form a prefix and
    fixed-size buckets can be all-reduced (async, RCCL's own stream) while the rest of backward still runs.  xGMI is
    point-to-point (7 links x ~153 GB/s): 32 MiB buckets keep each ring step long enough to run at link speed while
    leaving >= 7 buckets to overlap for the 243 MB CIFAR gradient.
  * clip + AdamW + EMA is two passes over the flat buffers (sum of squares, then one fused update kernel) instead of
    the reference's >= 5 foreach passes.
"""
import math
import os

import torch
import torch.distributed as dist

from . import _hip

BUCKET_BYTES = 32 << 20
FLAG_SLOTS = 4            # floats behind the flat gradient buffer that travel with its last bucket (16 bytes: keeps buckets float4-sized)
# CUs the persistent convolution kernels leave to RCCL in data-parallel runs unless VD_RESERVE_CUS says otherwise.  0: measured on one MI355X
# (1-rank RCCL, reducer on) reserving 8 CUs costs 6 % of the step -- the dominant kernel's 2048 work items need a ninth
# round on 248 CUs -- while an unreserved bucket waits at most one persistent launch (< 0.8 ms) and only the last bucket's wait is exposed.
RESERVE_CUS_DP = 0


def completion_order(model):
    """Parameter names in the order ``UNetEngine.backward`` finishes their gradients (engine.completion_order)."""
    return model.engine().completion_order()


class FlatState:
    """Flat fp32 buffers for parameters / gradients / Adam moments / EMA, with per-tensor views."""

    def __init__(self, model, use_ema=True):
        self.model = model
        order = completion_order(model)
        params = dict(model.named_parameters())
        dev = next(model.parameters()).device
        offs, n = {}, 0
        for k in order:
            offs[k] = n
            n += (params[k].numel() + 3) // 4 * 4                      # keep every tensor 16-byte aligned
        self.numel, self.offsets, self.order = n, offs, order
        self.p = torch.zeros(n, dtype=torch.float32, device=dev)
        # gradients + FLAG_SLOTS floats behind them: slot 0 says "this rank's micro-batch carried labels" (see cls_range below); it rides
        # in the last gradient bucket, so the rank SUM of the all-reduce makes it one decision for all replicas at no extra collective
        self.g_all = torch.zeros(n + FLAG_SLOTS, dtype=torch.float32, device=dev)
        self.g = self.g_all[:n]
        self.flag = self.g_all[n:n + 1]
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        views = {}
        with torch.no_grad():
            for k in order:
                q = params[k]
                pv = self.p[offs[k]: offs[k] + q.numel()].view_as(q)
                pv.copy_(q)
                q.data = pv                                            # the module now lives in the flat buffer
                views[k] = self.g[offs[k]: offs[k] + q.numel()].view_as(q)
        self.grad_views = views
        model._flat_grad_views = views
        self.ema = self.p.clone() if use_ema else None
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.ema_updates = 0
        # the class-embedding tensors (unet.py:207-215) are the one parameter group that may see no gradient in a step (a class-
        # conditional network called with y = None: the reference leaves .grad None and torch.optim.AdamW skips those parameters,
        # so their per-parameter step count lags): a contiguous range at the end of the flat buffers with its own step count
        cls = [k for k in order if k.startswith("class_embed.")]
        self.cls_names = set(cls)
        self.cls_range = None
        if cls:
            lo = min(offs[k] for k in cls)
            hi = max(offs[k] + (params[k].numel() + 3) // 4 * 4 for k in cls)
            assert all(lo <= offs[k] < hi for k in cls) and not any(lo <= offs[k] < hi for k in order if k not in self.cls_names), \
                "class-embedding tensors are not contiguous in the flat buffer"
            self.cls_range = (lo, min(hi, n))
        self.cls_steps_dev = torch.zeros(1, dtype=torch.int32, device=dev)      # advanced by vd_adamw_ema_flagged, never read per step

    @property
    def cls_steps(self):
        """updates in which the class embedding received a gradient (its torch.optim.AdamW per-parameter step): lives on the device;
        reading it synchronises (checkpoints only)"""
        return int(self.cls_steps_dev.item())

    @cls_steps.setter
    def cls_steps(self, k):
        self.cls_steps_dev.fill_(int(k))

    def ema_state_dict(self):
        """EMA shadow as a reference-format state_dict (utils.py:168-175 keeps ``shadow`` per parameter name)."""
        params = dict(self.model.named_parameters())
        return {k: self.ema[self.offsets[k]: self.offsets[k] + params[k].numel()].view_as(params[k]) for k in self.order}


class GradReducer:
    """Bucketed mean all-reduce of the flat gradient buffer, overlapped with the tail of backward."""

    def __init__(self, flat: FlatState, world_size, bucket_bytes=BUCKET_BYTES, group=None, force=False):
        self.flat, self.world, self.group = flat, world_size, group
        self.active = world_size > 1 or force        # force: run the collectives even on one rank (tests)
        n, per = flat.g_all.numel(), max(bucket_bytes // 4, 1)        # (the flag slots behind the gradients ride in the last bucket)
        self.total = n
        self.bounds = [(a, min(a + per, n)) for a in range(0, n, per)]
        # first flat offset AFTER each parameter, in completion order -> "ready prefix" length
        params = dict(flat.model.named_parameters())
        self.end_of = {k: flat.offsets[k] + params[k].numel() for k in flat.order}
        self.works, self.next_bucket = [], 0
        self.trace = None

    def start(self):
        self.works, self.next_bucket = [], 0
        if self.trace is not None:                 # (bench.py self-check: where inside backward does every bucket leave?)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.trace = [ev]

    def _launch(self, lo, hi):
        t = self.flat.g_all[lo:hi]
        if self.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                            # on the compute stream: the point of backward at which this bucket's gradients were final
            self.trace.append(ev)
        self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def launch_points_ms(self):
        """[ms after the start of backward at which bucket i was handed to the collective] + [ms at which backward ended] of the last
        traced step (``self.trace = []`` before the step switches tracing on); synchronises."""
        if not self.trace or len(self.trace) < 2:
            return None
        torch.cuda.synchronize()
        t0 = self.trace[0]
        return [round(t0.elapsed_time(e), 3) for e in self.trace[1:]]

    def ready(self, name):
        """Called by the backward pass when ``name`` (and everything before it in completion order) is final."""
        if not self.active:
            return
        upto = self.total if name is None else self.end_of[name]
        while self.next_bucket < len(self.bounds) and self.bounds[self.next_bucket][1] <= upto:
            self._launch(*self.bounds[self.next_bucket])
            self.next_bucket += 1

    def finish(self):
        if not self.active:
            return
        while self.next_bucket < len(self.bounds):
            self._launch(*self.bounds[self.next_bucket])
            self.next_bucket += 1
        if self.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                            # end of backward on the compute stream (before the waits below)
            self.trace.append(ev)
        for w in self.works:
            w.wait()
        self.works = []


class DeviceRunningStatistics:
    """Reference ``RunningStatistics`` (train_utils.py:30-59) with the sums kept on the device.

    The reference closes every step with ``self.stats.update(B, loss=loss.item() * B)`` (train_utils.py:169): one host
    synchronisation per step, which drains the launch queue (measured: 70.1 -> 72.0 ms per CIFAR step).  Here ``update``
    takes the device scalar the step already has and adds ``n * value`` into a float64 device accumulator (the reference's
    sums are Python floats: float64 too); the host reads the sums only in ``extract`` / ``repr`` -- once per progress-bar
    refresh or epoch instead of once per step.  Plain numbers are accepted as well (the reference call style)."""

    def __init__(self, device=None, **kwargs):
        self.device = device
        self.count = 0
        self.keys = list(kwargs)
        self.init = [float(v or 0) for v in kwargs.values()]
        self._sums = None

    def _buf(self):
        if self._sums is None:
            self._sums = torch.tensor(self.init, dtype=torch.float64, device=self.device)
        return self._sums

    def reset(self):
        self.count = 0
        self.init = [0.0] * len(self.keys)
        if self._sums is not None:
            self._sums.zero_()

    def update(self, n, **kwargs):
        """``n`` samples; every value is the batch SUM (``loss * B``) as a device scalar or a number (the reference's contract)"""
        self.count += n
        for k, v in kwargs.items():
            if k not in self.keys:
                self.keys.append(k)
                self.init.append(0.0)
                if self._sums is not None:
                    self._sums = torch.cat([self._sums, self._sums.new_zeros(1)])
            i = self.keys.index(k)
            if torch.is_tensor(v):
                self._buf()[i:i + 1].add_(v.detach().reshape(1).to(torch.float64))
            else:
                self._buf()[i] += float(v)

    @property
    def stats(self):
        return dict(zip(self.keys, self._buf().tolist()))          # the one host synchronisation

    def extract(self):
        return {k: v / self.count for k, v in self.stats.items()}

    def __repr__(self):
        return "Count(s): {}\nStatistics:\n".format(self.count) + "".join(f"\t{k} = {v}\n" for k, v in self.stats.items())


class HotPathTrainer:
    """Equivalent of reference ``Trainer.step`` for one rank (see module docstring)."""

    def __init__(self, model, diffusion, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, warmup=5000,
                 grad_norm=1.0, ema_decay=0.9999, use_ema=True, num_accum=1, timesteps=0, rank=0, world_size=1, group=None):
        self.model, self.diffusion = model, diffusion
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.warmup, self.grad_norm, self.ema_decay, self.num_accum, self.timesteps = warmup, grad_norm, ema_decay, num_accum, timesteps
        self.rank, self.world = rank, world_size
        self.device = next(model.parameters()).device
        self.flat = FlatState(model, use_ema=use_ema)
        self.reducer = GradReducer(self.flat, world_size, group=group)
        model._grads_ready_hook = None
        self.generator = torch.Generator(self.device).manual_seed(8191 + rank)       # train_utils.py:124
        self.stats = DeviceRunningStatistics(device=self.device, loss=None)           # train_utils.py:135
        # the leader = first member of the process group (global rank 0 need not belong to a sub-group)
        self.leader = dist.get_global_rank(group, 0) if (world_size > 1 and group is not None) else 0
        self.is_leader = world_size == 1 or (dist.get_rank() == self.leader if dist.is_initialized() else rank == 0)
        if world_size > 1 and os.environ.get("VD_RESERVE_CUS") is None and RESERVE_CUS_DP:
            # RCCL's kernels must find a CU while the persistent convolution kernels hold theirs for a whole launch (one workgroup per
            # CU, all of its LDS): leave RESERVE_CUS_DP of them out of those grids (vd_set_reserved_cus; 1-GPU cost in DESIGN section 4)
            _hip.lib().vd_set_reserved_cus(RESERVE_CUS_DP)
        if world_size > 1:                                                            # DDP ctor broadcast (train.py:148)
            dist.broadcast(self.flat.p, src=self.leader, group=group)
            if self.flat.ema is not None:
                self.flat.ema.copy_(self.flat.p)

    def draw(self, x):
        B = x.shape[0]
        if self.timesteps > 0:
            t = torch.randint(self.timesteps, size=(B,), dtype=torch.float64, device=self.device,
                              generator=self.generator).add(1).div(self.timesteps)
        else:
            t = torch.rand((B,), dtype=torch.float64, device=self.device, generator=self.generator)
        noise = torch.empty_like(x).normal_(generator=self.generator)
        return t, noise

    def step(self, x, y, update=True, t=None, noise=None):
        """One micro-batch: returns the detached mean loss (device tensor; no host sync here).  ``t`` / ``noise`` replace the draw from
        the per-rank generator (tests: the same rows through one rank and through two must give the same update)."""
        flat = self.flat
        if t is None or noise is None:
            t, noise = self.draw(x)
        if flat.cls_range is not None:
            # did this micro-batch reach the class embedding?  One float behind the gradients: it is summed over ranks with the last
            # gradient bucket (and over micro-batches with the accumulation buffer), and vd_adamw_ema_flagged reads it on the device --
            # whether the class embedding is updated is ONE decision for all replicas (the gradients are averaged over ranks; a
            # rank-local decision would let replicas that saw y = None skip an update the others apply), and no host waits for another
            flat.flag.fill_(0.0 if y is None else 1.0)
        loss = self.diffusion.train_loss(self.model, x_0=x, t=t, y=y, noise=noise).mean()
        self.reducer.start()
        self.model._grads_ready_hook = self.reducer.ready if self.reducer.active else None     # (no listener: backward joins its side stream once, at the end)
        (loss / (self.num_accum * self.world)).backward()          # 1/world folds DDP's gradient averaging into the seed
        self.model._grads_ready_hook = None
        self.reducer.finish()
        if self.num_accum > 1:                                     # the engine overwrites flat.g: keep the running sum (train_utils.py:154,
            if getattr(self, "_acc", None) is None:                # :257 -- loss / num_accum, .grad accumulated over the micro-batches)
                self._acc = torch.zeros_like(flat.g_all)
            self._acc.add_(flat.g_all)
            if update:
                flat.g_all.copy_(self._acc)
                self._acc.zero_()
        if update:
            flat.step_count += 1
            k = flat.step_count
            lr = self.lr * (min(k / self.warmup, 1.0) if self.warmup > 0 else 1.0)    # LambdaLR, train.py:160-162
            _hip.sumsq(flat.g, flat.gnorm_sq)
            decay = self.ema_decay
            if flat.ema is not None:
                flat.ema_updates += 1
                decay = min(self.ema_decay, (1 + flat.ema_updates) / (10 + flat.ema_updates))   # utils.py:145-146
            args = (flat.p, flat.g, flat.m, flat.v, flat.ema, flat.gnorm_sq, float(self.grad_norm), lr, self.betas[0], self.betas[1],
                    self.eps, self.wd, 1 - self.betas[0] ** k, 1 - self.betas[1] ** k, decay)
            if flat.cls_range is not None:
                # reference semantics of parameters without a gradient (torch.optim.AdamW skips them: no moment decay, no weight
                # decay, no update, their own step counter), decided and counted on the device: see vd_adamw_ema_flagged
                _hip.adamw_ema_flagged(*args, flat.cls_range[0], flat.cls_range[1], flat.flag, flat.cls_steps_dev)
            else:
                _hip.adamw_ema(*args)
        loss = loss.detach()
        if self.world > 1:                                         # train_utils.py:156-158: the leader reports the rank mean
            dist.reduce(loss, dst=self.leader, op=dist.ReduceOp.SUM, group=self.reducer.group)
            loss.div_(self.world)
        self.stats.update(x.shape[0], loss=loss * x.shape[0])      # train_utils.py:169, without its .item()
        return loss

    def step_uint8(self, u8_hwc, y, flip=None, **kw):
        """``step`` on a batch as the dataset holds it BEFORE the reference's CPU transforms (datasets.py:111-126: PIL image ->
        RandomHorizontalFlip -> ToTensor -> Normalize(0.5, 0.5), run by the DataLoader workers in float): ``u8_hwc`` (B, H, W, C) uint8
        on the device, ``flip`` (B,) bool / uint8 or None (the flip decisions; drawn by the loader, data not model state).  One launch
        (vd_images_from_uint8_hwc) produces the normalised fp32 NCHW batch q_sample and the loss read: the host ships a quarter of the
        bytes and does no float work -- what keeps 8 GPUs at > 10 k img/s fed without 8 x 4 transform workers."""
        from .functions import from_uint8_images
        return self.step(from_uint8_images(u8_hwc, flip), y, **kw)

    @property
    def current_stats(self):                                       # train_utils.py:305-307
        return self.stats.extract()

    # ------------------------------------------------------------------ EMA weights for sampling (utils.py:151-166)
    def ema_weights(self):
        """``with trainer.ema_weights(): ...`` runs the body on the EMA shadow.  The reference clones every parameter
        (``EMA.apply``) and copies back (``restore``); here the module's ``.data`` views are re-pointed from the flat
        parameter buffer to the flat shadow buffer and back -- no device traffic at all."""
        return _EmaSwap(self.flat)

    # ------------------------------------------------------------------ checkpoints (train_utils.py:309-348)
    def _lr_now(self):
        k = self.flat.step_count
        return self.lr * (min(k / self.warmup, 1.0) if self.warmup > 0 else 1.0)

    def state_dicts(self):
        """The reference's ``named_state_dicts`` payload: the same keys and per-entry formats ``Trainer.load_checkpoint``
        feeds to ``UNet.load_state_dict`` / ``torch.optim.AdamW.load_state_dict`` / ``EMA.load_state_dict`` /
        ``LambdaLR.load_state_dict`` (optimizer state indexed by position in ``model.parameters()``)."""
        flat, names = self.flat, [k for k, _ in self.model.named_parameters()]
        params = dict(self.model.named_parameters())

        def view(buf, k):
            return buf[flat.offsets[k]: flat.offsets[k] + params[k].numel()].view_as(params[k])

        out = {"model": {k: v.detach().clone() for k, v in self.model.state_dict().items()}}
        state = {}
        if flat.step_count > 0:
            for i, k in enumerate(names):
                steps = flat.cls_steps if k in flat.cls_names else flat.step_count
                if steps == 0:
                    continue                 # never received a gradient: torch.optim.AdamW holds no state for it
                state[i] = {"step": torch.tensor(float(steps)), "exp_avg": view(flat.m, k).clone(), "exp_avg_sq": view(flat.v, k).clone()}
        group = {"lr": self._lr_now(), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "initial_lr": self.lr, "params": list(range(len(names)))}
        out["optimizer"] = {"state": state, "param_groups": [group]}
        if flat.ema is not None:
            out["ema"] = {"decay": self.ema_decay, "shadow": {k: view(flat.ema, k).clone() for k in names},
                          "num_updates": flat.ema_updates}
        out["scheduler"] = {"base_lrs": [self.lr], "last_epoch": flat.step_count, "_step_count": flat.step_count + 1,
                            "_get_lr_called_within_step": False, "_last_lr": [self._lr_now()], "lr_lambdas": [None]}
        return out

    def gather_rng_states(self):
        """{rank: generator state} of EVERY rank (train_utils.py:277-291: all_gather as int64 on the device, back to uint8);
        a collective when world_size > 1 -- all ranks must call it."""
        state = self.generator.get_state()
        if self.world == 1:
            return {self.rank: state}
        mine = state.to(torch.int64).to(self.device)
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.reducer.group)
        return {r: q.cpu().to(torch.uint8) for r, q in enumerate(parts)}

    def save_checkpoint(self, path, **extra):
        """All ranks call this (the generator states are gathered: the only collective); only the leader -- the first rank of
        the group -- clones the state (4x the parameters on the device) and writes the file."""
        rng = self.gather_rng_states()
        if not self.is_leader:
            return
        ckpt = self.state_dicts()
        ckpt["rng"] = rng
        ckpt.update(extra)
        torch.save(ckpt, path)

    def load_checkpoint(self, path_or_dict, map_location=None):
        """Reads a checkpoint written by the reference ``Trainer.save_checkpoint`` (or by ``save_checkpoint`` above):
        weights, Adam moments and step, EMA shadow and update count, LR-schedule position, per-rank generator state.
        ``module.``-prefixed keys (saved from a DDP wrapper) are accepted as the reference does (:319-323)."""
        ckpt = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location=map_location or "cpu")
        flat, names = self.flat, [k for k, _ in self.model.named_parameters()]
        params = dict(self.model.named_parameters())

        def strip(d):
            return {(k[7:] if k.startswith("module.") else k): v for k, v in d.items()}

        def view(buf, k):
            return buf[flat.offsets[k]: flat.offsets[k] + params[k].numel()].view_as(params[k])

        self.model.load_state_dict(strip(ckpt["model"]))
        with torch.no_grad():
            opt = ckpt.get("optimizer")
            if opt is not None:
                st = opt["state"]
                ncls = len(flat.cls_names)
                assert len(st) in (0, len(names), len(names) - ncls), "optimizer state does not match the parameter list"
                flat.m.zero_(); flat.v.zero_()
                steps, csteps = set(), set()
                for i, k in enumerate(names):
                    e = st.get(i, st.get(str(i)))
                    if e is None:
                        continue
                    view(flat.m, k).copy_(e["exp_avg"])
                    view(flat.v, k).copy_(e["exp_avg_sq"])
                    (csteps if k in flat.cls_names else steps).add(int(e["step"]))
                # (only the class-embedding tensors may lag: they are the one group a step can leave without a gradient)
                assert len(steps) <= 1 and len(csteps) <= 1, "per-parameter step counts differ"
                flat.step_count = steps.pop() if steps else 0
                flat.cls_steps = csteps.pop() if csteps else 0
            ema = ckpt.get("ema")
            if ema is not None and flat.ema is not None:
                shadow = strip(ema["shadow"])
                missing = set(names) ^ set(shadow)
                if missing:
                    raise RuntimeError(f"EMA key mismatch: {sorted(missing)[:4]} ...")
                for k in names:
                    view(flat.ema, k).copy_(shadow[k])
                flat.ema_updates = int(ema["num_updates"])
                self.ema_decay = float(ema.get("decay", self.ema_decay))
            sch = ckpt.get("scheduler")
            if sch is not None and opt is None:
                flat.step_count = int(sch.get("last_epoch", flat.step_count))
                flat.cls_steps = flat.step_count
        if ckpt.get("rng") is not None:
            if int(self.rank) in ckpt["rng"]:
                self.generator.set_state(ckpt["rng"][int(self.rank)].cpu())
            else:
                import warnings
                warnings.warn(f"checkpoint holds no generator state for rank {self.rank} (has {sorted(ckpt['rng'])}): this rank "
                              f"continues from its fresh seed and will replay the t/noise stream it drew before the checkpoint")
        return ckpt.get("epoch", 0)


class _EmaSwap:
    def __init__(self, flat):
        self.flat = flat

    def _point(self, buf):
        flat = self.flat
        for k, q in flat.model.named_parameters():
            q.data = buf[flat.offsets[k]: flat.offsets[k] + q.numel()].view_as(q)

    def __enter__(self):
        assert self.flat.ema is not None, "trainer was built with use_ema=False"
        self._point(self.flat.ema)
        return self

    def __exit__(self, *exc):
        self._point(self.flat.p)
        return False
