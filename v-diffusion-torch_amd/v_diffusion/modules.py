"""Parameter containers with the reference's names, shapes and initialisation (v_diffusion/modules.py:25-144,184-208).

On the hot path these modules do not compute: ``UNet.forward`` hands their parameters to the HIP engine
(``engine.py``).  Their state_dict keys / shapes / ``parameters()`` order are the drop-in contract
(reference checkpoints load unchanged, optimizer state is indexed by parameter order).
"""
import math

import torch
import torch.nn as nn

DEFAULT_DTYPE = torch.float32


def lecun_normal_(tensor, scale: float = 1.):
    """In place: N(0,1) truncated at +-2 sigma, times sqrt(scale / fan_in), no variance correction; scale == 0 gives
    exact zeros (reference modules.py:25-35 -- the zero-initialised conv2 / proj_out / out_conv tensors rely on it)."""
    if tensor.ndim < 2:
        raise ValueError("lecun_normal_ needs a tensor with a fan-in dimension")
    fan_in = tensor[0].numel()
    with torch.no_grad():
        nn.init.trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.).mul_(math.sqrt(scale / fan_in))
    return tensor


DEFAULT_INITIALIZER = lecun_normal_


def pair(v):
    """int -> (int, int); sequences pass through as tuples"""
    try:
        return tuple(v)
    except TypeError:
        return (v, v)


class _WeightBias(nn.Module):
    """``weight`` (+ optional ``bias``) holder shared by Linear and Conv2d: lecun-normal weight, zero bias."""

    def _make(self, weight_shape, with_bias, init_scale):
        self.init_scale = init_scale
        self.weight = nn.Parameter(torch.empty(weight_shape, dtype=DEFAULT_DTYPE))
        self.bias = nn.Parameter(torch.empty(weight_shape[:1], dtype=DEFAULT_DTYPE)) if with_bias else None
        self._init()

    def _init(self):
        DEFAULT_INITIALIZER(self.weight, scale=self.init_scale)
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()


class Linear(_WeightBias):
    """weight (out, in), bias (out) -- reference modules.py:55-84"""

    def __init__(self, in_features, out_features, bias=True, init_scale=1.):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self._make((out_features, in_features), bias, init_scale)

    reset_parameters = _WeightBias._init

    def extra_repr(self):
        return f"{self.in_features} -> {self.out_features}, bias={self.bias is not None}"


class Conv2d(_WeightBias):
    """weight OIHW, bias (O) -- reference modules.py:87-144.  Only the two shapes the UNet uses are executable on the HIP
    path (3x3 stride 1 pad 1, and 1x1); the other constructor arguments are recorded for repr / compatibility."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", init_scale=1.):
        super().__init__()
        self.in_channels, self.out_channels, self.groups, self.padding_mode = in_channels, out_channels, groups, padding_mode
        self.kernel_size, self.stride, self.dilation = pair(kernel_size), pair(stride), pair(dilation)
        self.padding = padding if isinstance(padding, str) else pair(padding)
        self._make((out_channels, in_channels // groups) + self.kernel_size, bias, init_scale)

    reset_parameter = _WeightBias._init            # (sic: the reference spells it without the plural on Conv2d)

    def extra_repr(self):
        return f"{self.in_channels} -> {self.out_channels}, k={self.kernel_size}, s={self.stride}, p={self.padding}"


class GroupNorm32(nn.Module):
    """Affine parameters of nn.GroupNorm(32, C, eps=1e-6) (reference unet.py:28-30): keys ``weight`` / ``bias``."""

    def __init__(self, num_channels, num_groups=32, eps=1e-6):
        super().__init__()
        if num_channels % num_groups:
            raise ValueError(f"{num_channels} channels cannot form {num_groups} groups")
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def extra_repr(self):
        return f"groups={self.num_groups}, channels={self.num_channels}, eps={self.eps}"


class OneHot(nn.Module):
    """Parameter-free placeholder keeping the reference's ``class_embed.1`` key (modules.py:184-201); the lookup is
    fused with the linear layer in one HIP kernel (vd_class_embed)."""

    def __init__(self, num_classes=-1, exclude_zero=False):
        super().__init__()
        self.num_classes, self.exclude_zero = num_classes, exclude_zero


class Sequential(nn.Sequential):
    """Container only (reference modules.py:204-208 forwards **kwargs); execution order is the engine's."""
