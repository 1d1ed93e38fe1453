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
