"""MI355X-native v-diffusion hot path behind the call surface of tqch/v-diffusion-torch
(reference v_diffusion/__init__.py:1-21 re-exports; only the hot-path names exist here)."""
from .diffusion import GaussianDiffusion, get_logsnr_schedule
from .models.unet import UNet

__all__ = ["GaussianDiffusion", "get_logsnr_schedule", "UNet"]
