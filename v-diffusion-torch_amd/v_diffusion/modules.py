"""Parameter containers with the reference's names, shapes and initialisation (v_diffusion/modules.py:25-144,184-208).

On the hot path these modules do not compute: ``UNet.forward`` hands their parameters to the HIP engine
(``engine.py``).  Their state_dict keys / shapes / ``parameters()`` order are the drop-in contract
(reference checkpoints load unchanged, optimizer state is indexed by parameter order).
"""
import math
from collections.abc import Iterable
from itertools import repeat

import torch
import torch.nn as nn

DEFAULT_DTYPE = torch.float32


def lecun_normal_(tensor, scale: float = 1.):
    """N(0,1) truncated at +-2, times sqrt(scale / fan_in); scale == 0 gives zeros (reference modules.py:25-35)."""
    assert tensor.ndim >= 2
    fan_in = tensor.shape[1] * (math.prod(tensor.shape[2:]) if tensor.ndim > 2 else 1)
    nn.init.trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.)
    with torch.no_grad():
        tensor.mul_(math.sqrt(scale / fan_in))
    return tensor


DEFAULT_INITIALIZER = lecun_normal_


def pair(x):
    return tuple(x) if isinstance(x, Iterable) else tuple(repeat(x, 2))


class Linear(nn.Module):
    """weight (out, in), bias (out) -- reference modules.py:55-84"""

    def __init__(self, in_features, out_features, bias=True, init_scale=1.):
        super().__init__()
        self.in_features, self.out_features, self.init_scale = in_features, out_features, init_scale
        self.weight = nn.Parameter(torch.empty((out_features, in_features), dtype=DEFAULT_DTYPE))
        if bias:
            self.bias = nn.Parameter(torch.empty((out_features,), dtype=DEFAULT_DTYPE))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        DEFAULT_INITIALIZER(self.weight, scale=self.init_scale)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class Conv2d(nn.Module):
    """weight OIHW, bias (O) -- reference modules.py:87-144.  Only the two shapes the UNet uses are executable on
    the HIP path: 3x3 stride 1 pad 1 and 1x1."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", init_scale=1.):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = pair(kernel_size)
        self.stride, self.dilation, self.groups = pair(stride), pair(dilation), groups
        self.padding = padding if isinstance(padding, str) else pair(padding)
        self.padding_mode, self.init_scale = padding_mode, init_scale
        self.weight = nn.Parameter(torch.empty((out_channels, in_channels // groups, *self.kernel_size), dtype=DEFAULT_DTYPE))
        if bias:
            self.bias = nn.Parameter(torch.empty((out_channels,), dtype=DEFAULT_DTYPE))
        else:
            self.register_parameter("bias", None)
        self.reset_parameter()

    def reset_parameter(self):
        DEFAULT_INITIALIZER(self.weight, scale=self.init_scale)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}")


class GroupNorm32(nn.Module):
    """Affine parameters of nn.GroupNorm(32, C, eps=1e-6) (reference unet.py:28-30): keys ``weight`` / ``bias``."""

    def __init__(self, num_channels, num_groups=32, eps=1e-6):
        super().__init__()
        assert num_channels % num_groups == 0
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def extra_repr(self):
        return f"{self.num_groups}, {self.num_channels}, eps={self.eps}"


class OneHot(nn.Module):
    """Parameter-free placeholder keeping the reference's ``class_embed.1`` key (modules.py:184-201); the lookup is
    fused with the linear layer in one HIP kernel (vd_class_embed)."""

    def __init__(self, num_classes=-1, exclude_zero=False):
        super().__init__()
        self.num_classes, self.exclude_zero = num_classes, exclude_zero


class Sequential(nn.Sequential):
    """Container only (reference modules.py:204-208 forwards **kwargs); execution order is the engine's."""
