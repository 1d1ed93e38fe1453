"""``UNet`` with the reference's constructor, parameter names/order and ``forward(x, t, y=None)`` contract
(reference v_diffusion/models/unet.py:151-322) -- executed by the HIP engine (``..engine``), MI355X only.

The module tree exists to own parameters under the reference's state_dict keys (``downsamples.level_1.0.0.conv1.weight``,
``middle.1.proj_in.weight``, ``out_conv.2.bias`` ...), so reference checkpoints load and DDP / EMA / optimizers see the
same ``parameters()`` order.  No sub-module computes anything by itself.
"""
import torch
import torch.nn as nn

from .. import _hip
from ..modules import Linear, Conv2d, Sequential, OneHot, GroupNorm32


class AttentionBlock(nn.Module):
    """GN -> 1x1 QKV -> softmax(QK^T/sqrt(d)) V -> 1x1 (zero-init) -> + x   (reference unet.py:33-81)"""

    def __init__(self, in_dim, head_dim=None, num_heads=None):
        super().__init__()
        if head_dim is None:
            assert num_heads is not None and in_dim % num_heads == 0
            head_dim = in_dim // num_heads
        if num_heads is None:
            assert head_dim is not None and in_dim % head_dim == 0
            num_heads = in_dim // head_dim
        self.head_dim, self.num_heads = head_dim, num_heads
        self.hid_dim = head_dim * num_heads
        self.norm = GroupNorm32(in_dim)
        self.proj_in = Conv2d(in_dim, 3 * self.hid_dim, 1)
        self.proj_out = Conv2d(self.hid_dim, in_dim, 1, init_scale=0.)


class ResidualBlock(nn.Module):
    """GN -> SiLU -> [resample] -> 3x3 -> FiLM(GN) -> SiLU -> dropout -> 3x3 (zero-init) + skip   (reference unet.py:106-148)"""

    def __init__(self, in_channels, out_channels, embed_dim, drop_rate=0., resampling="none"):
        super().__init__()
        assert resampling in ("none", "upsample", "downsample")
        self.in_channels, self.out_channels, self.resampling, self.drop_rate = in_channels, out_channels, resampling, drop_rate
        self.norm1 = GroupNorm32(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, 3, 1, 1)
        self.fc = Linear(embed_dim, 2 * out_channels)
        self.norm2 = GroupNorm32(out_channels)
        self.conv2 = Conv2d(out_channels, out_channels, 3, 1, 1, init_scale=0.)
        self.skip = nn.Identity() if in_channels == out_channels else Conv2d(in_channels, out_channels, 1)


class _UNetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward runs the engine and keeps its tape, backward runs the engine's
    hand-written backward and hands one gradient per parameter to autograd (so DDP hooks, ``.grad`` accumulation and
    optimizers behave as with the reference)."""

    @staticmethod
    def forward(ctx, model, x, t, y, *params):
        eng = model.engine()
        with torch.cuda.device(x.device):            # kernels go to the current stream of the tensor's device
            out, tape = eng.forward(x, t, y, model.training, save=True)
            ctx.model, ctx.tape, ctx.need_dx = model, tape, x.requires_grad
            return model._to_nchw(out)

    @staticmethod
    def backward(ctx, dout):
        model, tape = ctx.model, ctx.tape
        if tape is None:
            raise RuntimeError("UNet backward called twice (the tape is freed after the first backward)")
        ctx.tape = None
        eng = model.engine()
        B, co, Hh, Ww = dout.shape
        cop = (co + 3) // 4 * 4
        flat = model._flat_grad_views is not None
        G = model._grad_targets()
        with torch.cuda.device(dout.device):
            d4 = torch.empty((B, Hh, Ww, cop), dtype=torch.float32, device=dout.device)
            _hip.nchw_to_nhwc(dout.to(torch.float32).contiguous(), d4, B, co, Hh, Ww, cop)
            dx = eng.backward(tape, d4, G, need_dx=ctx.need_dx, progress=getattr(model, "_grads_ready_hook", None))
        if flat:
            # the kernels already wrote into the caller's flat gradient buffer (trainer.FlatState): autograd gets nothing
            # to accumulate, which avoids a 243 MB clone per step
            return (None, dx, None, None) + (None,) * len(G)
        # class-conditional network called without labels: the class embedding took no part in the forward, so -- as in the
        # reference, where autograd never reaches those parameters -- their gradient is None (torch optimizers then skip them)
        no_labels = bool(model.num_classes) and tape["embed"]["yn"] is None
        return (None, dx, None, None) + tuple(None if (no_labels and k.startswith("class_embed.")) else G[k]
                                              for k, _ in model.named_parameters())


class UNet(nn.Module):
    def __init__(self, in_channels, hid_channels, out_channels, ch_multipliers, num_res_blocks, apply_attn,
                 embedding_dim=None, drop_rate=0., head_dim=None, num_heads=None, num_classes=0, multitags=False,
                 resample_with_res=True, use_xformers=False):
        super().__init__()
        self.in_channels, self.hid_channels, self.out_channels = in_channels, hid_channels, out_channels
        self.embedding_dim = embedding_dim or 4 * hid_channels
        self.levels = levels = len(ch_multipliers)
        self.ch_multipliers = ch_multipliers
        if isinstance(apply_attn, bool):
            apply_attn = [apply_attn] * levels
        self.apply_attn = apply_attn
        self.num_res_blocks, self.drop_rate = num_res_blocks, drop_rate
        if head_dim is None and num_heads is None:
            num_heads = 1
        self.head_dim, self.num_heads = head_dim, num_heads
        self.num_classes, self.multitags = num_classes, multitags
        self.resample_with_res = resample_with_res
        if not resample_with_res:
            raise NotImplementedError("resample_with_res=False (strided-conv resampling, reference unet.py:258-260,279-282) "
                                      "is not on the hot path: no shipped config uses it")
        if use_xformers:
            print("xFormers not available! Resetting to False.")       # reference unet.py:192-194 (same fallback message)
        if hid_channels % 32 or any((hid_channels * k) % 32 for k in ch_multipliers):
            raise ValueError("channel counts must be multiples of 32 (GroupNorm(32, C))")

        emb = self.embedding_dim
        self.time_embed = nn.Sequential(Linear(hid_channels, emb), nn.SiLU(), Linear(emb, emb))
        if num_classes > 0:
            if multitags:
                self.class_embed = nn.Linear(num_classes, emb)
            else:
                self.class_embed = nn.Sequential(OneHot(num_classes, exclude_zero=True), Linear(num_classes, emb))
        self.in_conv = Conv2d(in_channels, hid_channels, 3, 1, 1)
        chs = [hid_channels * k for k in ch_multipliers]

        def block(level, cin, cout, resampling="none"):
            rb = ResidualBlock(cin, cout, emb, drop_rate, resampling)
            if apply_attn[level]:
                return Sequential(rb, AttentionBlock(cout, head_dim=head_dim, num_heads=num_heads))
            return rb

        downs = {}
        for i in range(levels):
            prev = chs[i - 1] if i else hid_channels
            mods = [block(i, prev, chs[i])] + [block(i, chs[i], chs[i]) for _ in range(num_res_blocks - 1)]
            if i != levels - 1:
                mods.append(block(i, chs[i], chs[i], "downsample"))
            downs[f"level_{i}"] = nn.ModuleList(mods)
        self.downsamples = nn.ModuleDict(downs)
        mid = chs[-1]
        self.middle = Sequential(ResidualBlock(mid, mid, emb, drop_rate),
                                 AttentionBlock(mid, head_dim=head_dim, num_heads=num_heads),
                                 ResidualBlock(mid, mid, emb, drop_rate))
        ups = {}
        for i in range(levels):
            nxt = hid_channels if i == 0 else chs[i - 1]
            prv = chs[-1] if i == levels - 1 else chs[i + 1]
            mods = [block(i, prv + chs[i], chs[i])] + [block(i, 2 * chs[i], chs[i]) for _ in range(num_res_blocks - 1)]
            mods.append(block(i, nxt + chs[i], chs[i]))
            if i != 0:
                mods.append(block(i, chs[i], chs[i], "upsample"))
            ups[f"level_{i}"] = nn.ModuleList(mods)
        self.upsamples = nn.ModuleDict(ups)
        self.out_conv = Sequential(GroupNorm32(chs[0]), nn.SiLU(), Conv2d(chs[0], out_channels, 3, 1, 1, init_scale=0.))
        self._engine = None
        self._flat_grad_views = None
        self._grads_ready_hook = None

    # ------------------------------------------------------------------ engine plumbing
    def engine(self):
        if self._engine is None:
            from ..engine import UNetEngine
            _hip.lib()                                   # fail loudly if the HIP library is missing
            self._engine = UNetEngine(self)
        return self._engine

    def _grad_targets(self):
        if self._flat_grad_views is not None:
            return self._flat_grad_views
        return {k: torch.empty_like(p) for k, p in self.named_parameters()}

    def _to_nchw(self, out_nhwc):
        B, Hh, Ww, cop = out_nhwc.shape
        y = torch.empty((B, self.out_channels, Hh, Ww), dtype=torch.float32, device=out_nhwc.device)
        _hip.nhwc_to_nchw(out_nhwc, cop, y, B, self.out_channels, Hh, Ww)
        return y

    def forward(self, x, t, y=None):
        """x (B,C,H,W) fp32, t (B,) fp64 in [0,1] (or step/T), y (B,) float labels 0..num_classes / (B,num_classes) tags."""
        if not x.is_cuda:
            raise RuntimeError("v_diffusion.UNet runs on an MI355X through libvdiff_hip.so only; there is no CPU path "
                               "(the CPU restatement lives under oracle/ and is test infrastructure)")
        if x.shape[0] == 0:                                   # empty batch: nothing to launch (F.conv2d returns an empty tensor too)
            return x.new_zeros((0, self.out_channels) + tuple(x.shape[2:]), dtype=torch.float32)
        params = list(self.parameters())
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _UNetFn.apply(self, x, t, y, *params)
        with torch.cuda.device(x.device):
            out, _ = self.engine().forward(x, t, y, self.training, save=False)
            return self._to_nchw(out)
