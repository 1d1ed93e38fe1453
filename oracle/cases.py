"""Shared case table for golden generation and oracle/parity tests (test infrastructure)."""
import torch
from . import detrand
from .unet_ref import param_shapes, normalize_cfg

CIFAR_COND = dict(in_channels=3, hid_channels=256, out_channels=3, ch_multipliers=[1, 1, 1], num_res_blocks=3,
                  apply_attn=[False, True, True], drop_rate=0.2, num_heads=1, num_classes=10, multitags=False)
CIFAR_UNCOND = dict(CIFAR_COND, num_classes=0)
# configs/celeba.json merged with configs/defaults.json (defaults inject num_heads=1) and --model-out-type v
CELEBA = dict(in_channels=3, hid_channels=192, out_channels=3, ch_multipliers=[1, 2, 3, 4], num_res_blocks=3,
              apply_attn=[False, True, True, True], embedding_dim=768, drop_rate=0.1, head_dim=64, num_heads=1,
              num_classes=40, multitags=True)

TINY = {
    # Cin != Cout skip conv, plain downsample block, attention on the up-sampling block, nh=2, single-label classes
    "tinyA": dict(cfg=dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2], num_res_blocks=1,
                           apply_attn=[False, True], num_heads=2, num_classes=10), B=3, R=16, label="single"),
    # head_dim and num_heads both given (hid_dim != C), multitags, out=6, attention on a down-sampling block
    "tinyB": dict(cfg=dict(in_channels=3, hid_channels=32, out_channels=6, ch_multipliers=[1, 1, 2], num_res_blocks=1,
                           apply_attn=[True, True, True], head_dim=16, num_heads=2, embedding_dim=96,
                           num_classes=40, multitags=True), B=2, R=16, label="multi"),
    # unconditional, two res blocks per level, 8x8
    "tinyC": dict(cfg=dict(in_channels=3, hid_channels=64, out_channels=3, ch_multipliers=[1, 1], num_res_blocks=2,
                           apply_attn=[False, True], num_heads=1, num_classes=0), B=2, R=8, label=None),
}


def make_inputs(cfg, B, R, label, seed=0):
    cfg = normalize_cfg(cfg)
    x = detrand.normal("x", (B, cfg["in_channels"], R, R), seed)
    t = detrand.uniform("t", (B,), seed, dtype=torch.float64)
    if label == "single":
        y = detrand.randint("y", (B,), 0, cfg["num_classes"] + 1, seed).float()   # 0 = "no label"
    elif label == "multi":
        y = (detrand.uniform("y", (B, cfg["num_classes"]), seed) < 0.2).float()
        y[0] = 0                                                                   # nnz = 0 row (clamp(min=1) branch)
    else:
        y = None
    return x, t, y


def full_size_inputs(cfg, B, R, label):
    """inputs of the full-size UNet fixtures (tests/golden/unet_cifar10_cond.npz, unet_celeba.npz): ONE statement for the generator and
    the tests that read them.  Single-label models: every row labelled except the last, which is UNLABELLED (class 0: the reference
    adds no class embedding, unet.py:289-295); multi-tag models: make_inputs' all-zero first row."""
    x, t, y = make_inputs(cfg, B, R, label)
    if label == "single":
        y = y.clamp(min=1)
        y[-1] = 0
    return x, t, y


_WEIGHT_CACHE = {}


def make_weights(cfg, seed=0):
    """deterministic weights of a config (oracle/detrand.py); generated once per process and config -- the integer hash costs ~20 s for
    the 267 M parameters of the CelebA model -- and handed out as clones (callers flip requires_grad in place)"""
    key = (repr(sorted(normalize_cfg(cfg).items())), seed)
    if key not in _WEIGHT_CACHE:
        _WEIGHT_CACHE[key] = detrand.fill_state_dict(param_shapes(cfg), seed)
    return {k: v.clone() for k, v in _WEIGHT_CACHE[key].items()}


def kl_case():
    """inputs of the variational-bound fixtures (tests/golden/ext_kl.npz; oracle/make_goldens_ext.py builds the same tensors)"""
    case = TINY["tinyA"]
    cfg = case["cfg"]
    x0, t, y = make_inputs(cfg, 6, case["R"], case["label"], seed=3)
    x0 = (x0.clamp(-1, 1) * 127.5).round() / 127.5          # 8-bit data, as the discretised likelihood assumes
    x0[0, :, :2] = 1.0                                       # both saturation branches of the decoder likelihood
    x0[1, :, :2] = -1.0
    t[0] = 0.05                                              # snaps to t = 1/T, s = 0: the decoder-NLL row
    noise = detrand.normal("noise", tuple(x0.shape), 3)
    out = detrand.normal("out", tuple(x0.shape), 4) * 0.5
    return cfg, x0, t, y, noise, out
