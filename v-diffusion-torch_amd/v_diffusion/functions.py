"""Hot-path helpers of reference v_diffusion/functions.py (only the ones on the path: SURVEY 2.1 #3)."""
import torch

from . import _hip

DEFAULT_DTYPE = torch.float32


def get_timestep_embedding(timesteps, embed_dim: int, dtype: torch.dtype = DEFAULT_DTYPE, scale: float = 1000.):
    """Sinusoidal embedding [sin | cos] of ``scale * t`` (reference functions.py:11-29), one HIP kernel.

    The arithmetic is fp64, as on the reference hot path where ``t`` is fp64 (train_utils.py:141-145,
    diffusion.py:399); an fp32/integer ``t`` is widened first (documented deviation: the reference would
    evaluate it in that narrower dtype)."""
    t = timesteps.reshape(-1)
    if t.dtype != torch.float64:
        t = t.to(torch.float64)
    out = torch.empty((t.shape[0], embed_dim), dtype=torch.float32, device=t.device)
    _hip.timestep_embedding(t.contiguous(), out, t.shape[0], embed_dim, scale)
    return out if dtype == torch.float32 else out.to(dtype)


def flat_mean(x, start_dim=1):
    """reference functions.py:102-104"""
    return torch.mean(x, dim=list(range(start_dim, x.ndim)))


def to_uint8_images(x):
    """(B,C,H,W) fp32 in [-1,1] -> (B,H,W,C) uint8 on the device: ``(x*127.5+127.5).clamp(0,255).to(uint8).permute(0,2,3,1)``
    of reference generate.py:149 as one kernel (quantise + NCHW->HWC pack; a 4x smaller D2H copy follows)."""
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), dtype=torch.uint8, device=x.device)
    _hip.images_to_uint8_hwc(x.to(torch.float32).contiguous(), out, B, C, H * W)
    return out


def from_uint8_images(u8, flip=None):
    """(B,H,W,C) uint8 -> (B,C,H,W) fp32 in [-1,1]: RandomHorizontalFlip (per-image mask ``flip``) + ToTensor +
    Normalize(0.5,0.5) of reference datasets.py:115-120 as one kernel, so raw uint8 batches can be uploaded as they are."""
    B, H, W, C = u8.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=u8.device)
    fm = None if flip is None else flip.to(torch.uint8).contiguous()
    _hip.images_from_uint8_hwc(u8.contiguous(), fm, out, B, C, H, W)
    return out
