"""CPU oracle for ``UNet.forward`` -- TEST INFRASTRUCTURE, not a product path.

Functional restatement (pure torch, NCHW, fp32 unless the input dtype says otherwise) driven by a
plain ``state_dict`` that uses the reference's parameter names.  Follows

  * UNet.forward ................ reference v_diffusion/models/unet.py:286-322
  * level construction .......... unet.py:234-284 (which blocks exist, which carry attention)
  * ResidualBlock.forward ....... unet.py:137-148
  * AttentionBlock.forward ...... unet.py:55-81
  * get_timestep_embedding ...... v_diffusion/functions.py:11-29
  * OneHot(exclude_zero) ........ v_diffusion/modules.py:184-201

It is differentiable through torch autograd, so gradients of the reference are restated too.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

GN_GROUPS = 32     # unet.py:28-30
GN_EPS = 1e-6


def normalize_cfg(cfg: dict) -> dict:
    c = dict(cfg)
    c.setdefault("embedding_dim", None)
    c["embedding_dim"] = c["embedding_dim"] or 4 * c["hid_channels"]          # unet.py:176
    levels = len(c["ch_multipliers"])
    if isinstance(c["apply_attn"], bool):
        c["apply_attn"] = [c["apply_attn"]] * levels                          # unet.py:179-180
    c.setdefault("head_dim", None)
    c.setdefault("num_heads", None)
    if c["head_dim"] is None and c["num_heads"] is None:
        c["num_heads"] = 1                                                    # unet.py:184-185
    c.setdefault("num_classes", 0)
    c.setdefault("multitags", False)
    c.setdefault("drop_rate", 0.0)
    return c


def attn_dims(cfg, channels):
    """(head_dim, num_heads) exactly as BaseAttentionBlock.__init__ resolves them (unet.py:36-53)."""
    hd, nh = cfg["head_dim"], cfg["num_heads"]
    if hd is None:
        hd = channels // nh
    if nh is None:
        nh = channels // hd
    return hd, nh


def block_plan(cfg):
    """List of (prefix, kind, cin, cout, resampling, has_attn, consumes_skip) in execution order."""
    cfg = normalize_cfg(cfg)
    hid, mult, nrb = cfg["hid_channels"], cfg["ch_multipliers"], cfg["num_res_blocks"]
    levels = len(mult)
    chs = [hid * m for m in mult]
    plan = []
    for i in range(levels):                                                   # unet.py:250-263
        attn = cfg["apply_attn"][i]
        prev = chs[i - 1] if i else hid
        for j in range(nrb):
            plan.append((f"downsamples.level_{i}.{j}", "down", prev if j == 0 else chs[i], chs[i], "none", attn, False))
        if i != levels - 1:
            plan.append((f"downsamples.level_{i}.{nrb}", "down", chs[i], chs[i], "downsample", attn, False))
    plan.append(("middle.0", "mid", chs[-1], chs[-1], "none", False, False))
    plan.append(("middle.1", "midattn", chs[-1], chs[-1], "none", True, False))
    plan.append(("middle.2", "mid", chs[-1], chs[-1], "none", False, False))
    for i in range(levels - 1, -1, -1):                                       # unet.py:265-284
        attn = cfg["apply_attn"][i]
        nxt = hid if i == 0 else chs[i - 1]
        prv = chs[-1] if i == levels - 1 else chs[i + 1]
        cins = [prv + chs[i]] + [2 * chs[i]] * (nrb - 1) + [nxt + chs[i]]
        for j, cin in enumerate(cins):
            plan.append((f"upsamples.level_{i}.{j}", "up", cin, chs[i], "none", attn, True))
        if i != 0:
            plan.append((f"upsamples.level_{i}.{nrb + 1}", "up", chs[i], chs[i], "upsample", attn, False))
    return plan


def param_shapes(cfg) -> "OrderedDict[str, tuple]":
    """Names and shapes in the reference's ``parameters()`` / ``state_dict()`` order."""
    cfg = normalize_cfg(cfg)
    hid, emb = cfg["hid_channels"], cfg["embedding_dim"]
    sh = OrderedDict()

    def lin(p, i, o):
        sh[p + ".weight"] = (o, i)
        sh[p + ".bias"] = (o,)

    def conv(p, i, o, k):
        sh[p + ".weight"] = (o, i, k, k)
        sh[p + ".bias"] = (o,)

    def gn(p, c):
        sh[p + ".weight"] = (c,)
        sh[p + ".bias"] = (c,)

    def res(p, cin, cout):                        # registration order of ResidualBlock.__init__ (unet.py:118-134)
        gn(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout, 3)
        lin(p + ".fc", emb, 2 * cout)
        gn(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip", cin, cout, 1)

    def attn(p, c):                               # AttentionBlock.__init__ (unet.py:52,70-71)
        hd, nh = attn_dims(cfg, c)
        gn(p + ".norm", c)
        conv(p + ".proj_in", c, 3 * hd * nh, 1)
        conv(p + ".proj_out", hd * nh, c, 1)

    lin("time_embed.0", hid, emb)
    lin("time_embed.2", emb, emb)
    if cfg["num_classes"] > 0:
        lin("class_embed" if cfg["multitags"] else "class_embed.1", cfg["num_classes"], emb)
    conv("in_conv", cfg["in_channels"], hid, 3)
    plan = block_plan(cfg)
    # execution visits the up levels from deepest to level 0, but the ModuleDict registers them as
    # level_0, level_1, ... (unet.py:228-229), which is what parameters()/state_dict() follow
    ups = [e for e in plan if e[1] == "up"]
    ups.sort(key=lambda e: int(e[0].split("level_")[1].split(".")[0]))
    plan = [e for e in plan if e[1] != "up"] + ups
    for prefix, kind, cin, cout, _, has_attn, _ in plan:
        if kind == "midattn":
            attn(prefix, cout)
        elif kind == "mid":
            res(prefix, cin, cout)
        elif has_attn:
            res(prefix + ".0", cin, cout)
            attn(prefix + ".1", cout)
        else:
            res(prefix, cin, cout)
    gn("out_conv.0", hid * cfg["ch_multipliers"][0])
    conv("out_conv.2", hid * cfg["ch_multipliers"][0], cfg["out_channels"], 3)
    return sh


def timestep_embedding(t, dim, scale=1000.0, out_dtype=torch.float32):
    """functions.py:11-29 -- arithmetic in t's dtype (fp64 on the hot path), then cast."""
    t = scale * t.reshape(-1)
    half = dim // 2
    step = math.log(10000) / (half - 1)
    freq = torch.exp(-torch.arange(half, dtype=t.dtype, device=t.device) * step)
    arg = t[:, None] * freq[None, :]
    e = torch.cat([arg.sin(), arg.cos()], dim=1).to(out_dtype)
    if dim % 2 == 1:
        e = F.pad(e, [0, 1])
    return e


def _gn(x, sd, p):
    return F.group_norm(x, GN_GROUPS, sd[p + ".weight"], sd[p + ".bias"], GN_EPS)


def res_block(x, temb_act, sd, p, resampling="none", drop_rate=0.0, drop_mask=None):
    """unet.py:137-148.  ``drop_mask`` (same shape as the conv2 input, values 0 or 1/(1-p)) makes the
    dropout explicit so the HIP path's counter-based mask can be replayed through the oracle."""
    if resampling == "downsample":
        rs = lambda z: F.avg_pool2d(z, 2)
    elif resampling == "upsample":
        rs = lambda z: F.interpolate(z, scale_factor=2, mode="nearest")
    else:
        rs = lambda z: z
    skip = rs(x)
    if p + ".skip.weight" in sd:
        skip = F.conv2d(skip, sd[p + ".skip.weight"], sd[p + ".skip.bias"])
    h = F.conv2d(rs(F.silu(_gn(x, sd, p + ".norm1"))), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    film = F.linear(temb_act, sd[p + ".fc.weight"], sd[p + ".fc.bias"])[:, :, None, None]
    shift, scale = film.chunk(2, dim=1)                       # shift is the FIRST half (unet.py:145)
    h = (1 + scale) * _gn(h, sd, p + ".norm2") + shift
    h = F.silu(h)
    if drop_mask is not None:
        h = h * drop_mask
    elif drop_rate > 0:
        h = F.dropout(h, drop_rate, training=True)
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    return h + skip


def attn_block(x, sd, p, head_dim, num_heads):
    """unet.py:55-81 (channel order of proj_in output: q heads, k heads, v heads)."""
    B, C, H, W = x.shape
    L = H * W
    qkv = F.conv2d(_gn(x, sd, p + ".norm"), sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    q, k, v = qkv.reshape(B, 3 * num_heads, head_dim, L).chunk(3, dim=1)       # each (B, nh, hd, L)
    logits = torch.einsum("bncl,bncm->bnlm", q, k) / math.sqrt(head_dim)
    w = torch.softmax(logits, dim=-1)
    o = torch.einsum("bnlm,bncm->bncl", w, v).reshape(B, num_heads * head_dim, H, W)
    o = F.conv2d(o, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return o + x


def embed(sd, cfg, t, y):
    cfg = normalize_cfg(cfg)
    # (fp32 as in the reference; a float64 state_dict -- the tests' fp32-vs-fp64 spread measurements -- gets the embedding in its dtype)
    te = timestep_embedding(t, cfg["hid_channels"], out_dtype=sd["time_embed.0.weight"].dtype)
    te = F.linear(te, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    te = F.linear(F.silu(te), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if cfg["num_classes"] and y is not None:
        if cfg["multitags"]:                                                   # unet.py:290-294
            nnz = torch.count_nonzero(y, dim=1).clamp(min=1.0).sqrt().unsqueeze(1)
            te = te + F.linear(y / nnz, sd["class_embed.weight"], sd["class_embed.bias"])
        else:                                                                  # modules.py:190-201
            yl = y.long()
            oh = F.one_hot((yl - 1).clamp(min=0), cfg["num_classes"]).to(te.dtype)
            oh = oh * (yl != 0).to(te.dtype)[:, None]
            te = te + F.linear(oh, sd["class_embed.1.weight"], sd["class_embed.1.bias"])
    return te


def unet_forward(sd, cfg, x, t, y=None, train=False, drop_masks=None, taps=None):
    """Returns model_out (B, out_channels, H, W).  ``taps`` (dict) collects named intermediates."""
    cfg = normalize_cfg(cfg)
    p_drop = cfg["drop_rate"] if train else 0.0
    te = embed(sd, cfg, t, y)
    ta = F.silu(te)                                       # every block applies act1 to t_emb (unet.py:142)
    if taps is not None:
        taps["t_emb"] = te
    hs = [F.conv2d(x, sd["in_conv.weight"], sd["in_conv.bias"], padding=1)]
    h = None
    for prefix, kind, cin, cout, resampling, has_attn, consumes in block_plan(cfg):
        if kind == "down":
            inp = hs[-1]
        elif kind in ("mid", "midattn"):
            inp = hs[-1] if h is None else h
        else:
            inp = torch.cat([h, hs.pop()], dim=1) if consumes else h            # unet.py:315 (h first)
        if kind == "midattn":
            hd, nh = attn_dims(cfg, cout)
            out = attn_block(inp, sd, prefix, hd, nh)
        else:
            rp = prefix + ".0" if has_attn else prefix
            dm = None if drop_masks is None else drop_masks.get(rp)
            out = res_block(inp, ta, sd, rp, resampling, p_drop, dm)
            if has_attn:
                hd, nh = attn_dims(cfg, cout)
                out = attn_block(out, sd, prefix + ".1", hd, nh)
        if taps is not None:
            taps[prefix] = out
        if kind == "down":
            hs.append(out)
        else:
            h = out
    h = F.silu(_gn(h, sd, "out_conv.0"))
    return F.conv2d(h, sd["out_conv.2.weight"], sd["out_conv.2.bias"], padding=1)
