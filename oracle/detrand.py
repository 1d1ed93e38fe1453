"""Deterministic, platform-independent tensor generator (test infrastructure).

Values come from a splitmix64 integer hash of (key, element index); four 16-bit uniforms are
summed (Irwin-Hall) to give an approximately normal variate using only exactly-representable
float64 arithmetic, so the same numbers appear on every host and fixtures only need to store
reference OUTPUTS, never weights or inputs.
"""
import zlib
import numpy as np
import torch

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def _key(name: str, seed: int) -> np.uint64:
    return np.uint64((zlib.crc32(name.encode()) << 20) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF))


def normal(name: str, shape, seed: int = 0, std: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + _key(name, seed)
    h = _splitmix64(idx)
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += ((h >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.float64)
    # each term uniform on {0..65535}: mean 32767.5, var (65536^2-1)/12
    z = (s - 4 * 32767.5) / np.sqrt(4 * (65536.0 ** 2 - 1) / 12.0)
    return torch.from_numpy((z * std).reshape(shape)).to(dtype)


def uniform(name: str, shape, seed: int = 0, lo: float = 0.0, hi: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + _key(name, seed)
    h = _splitmix64(idx)
    u = ((h >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
    return torch.from_numpy((lo + (hi - lo) * u).reshape(shape)).to(dtype)


def randint(name: str, shape, lo: int, hi: int, seed: int = 0) -> torch.Tensor:
    """integers in [lo, hi)"""
    u = uniform(name, shape, seed, dtype=torch.float64)
    return (u * (hi - lo)).floor().long() + lo


def sign_patterns(name: str, n: int, k: int = 4, start: int = 0) -> np.ndarray:
    """[k][n] values of +-1 (float64): k sign patterns for elements start .. start + n - 1 of a tensor, bits 40 .. 40 + k - 1 of ONE hash per
    element (the gradient digests of the full-size goldens project every tensor on them)"""
    with np.errstate(over="ignore"):
        idx = np.arange(start, start + n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + _key("proj:" + name, 0)
    h = _splitmix64(idx)
    return np.stack([1.0 - 2.0 * ((h >> np.uint64(40 + j)) & np.uint64(1)).astype(np.float64) for j in range(k)])


def _i64(c: int) -> int:
    """a uint64 constant as the int64 with the same bits (torch has no uint64 arithmetic; int64 products wrap to the same low 64 bits)"""
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    return (x >> s) & ((1 << (64 - s)) - 1)                # logical shift right of int64 bit patterns


def projections(name: str, g: torch.Tensor, k: int = 4, chunk: int = 1 << 24) -> np.ndarray:
    """k deterministic +-1 projections of a tensor, float64: sum_i s_j(i) g_i for the sign patterns of ``sign_patterns(name, ...)`` -- the same
    splitmix64 hash in torch int64 arithmetic, on the tensor's own device (a GPU test projects 267 M gradient elements in a second; numpy
    needs half a minute).  Unlike a norm the projections see the position and the sign of every element: a permuted, transposed or
    sign-flipped block moves them by the size of that block."""
    v = g.detach().double().flatten()
    key = int(_key("proj:" + name, 0))
    out = torch.zeros(k, dtype=torch.float64, device=v.device)
    for s0 in range(0, v.numel(), chunk):
        part = v[s0:s0 + chunk]
        x = torch.arange(s0, s0 + part.numel(), dtype=torch.int64, device=v.device) * _i64(0x2545F4914F6CDD1D) + _i64(key)
        x = x + _i64(0x9E3779B97F4A7C15)
        z = (x ^ _lsr(x, 30)) * _i64(0xBF58476D1CE4E5B9)
        z = (z ^ _lsr(z, 27)) * _i64(0x94D049BB133111EB)
        h = z ^ _lsr(z, 31)
        for j in range(k):
            sgn = 1.0 - 2.0 * ((h >> (40 + j)) & 1).to(torch.float64)
            out[j] += (sgn * part).sum()
    return out.cpu().numpy()


def fill_state_dict(shapes: dict, seed: int = 0) -> dict:
    """Random-looking weights for every tensor of a UNet state_dict.

    Conv/linear weights ~ N(0, 1/fan_in) (no zero-initialised tensors, so no path is hidden
    behind zeros -- reference trap at unet.py:71,125,232); norm weights ~ 1 + 0.1 N; biases 0.1 N.
    """
    out = {}
    for name, shp in shapes.items():
        shp = tuple(shp)
        if len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            out[name] = normal(name, shp, seed, std=fan_in ** -0.5)
        elif "norm" in name and name.endswith("weight") or name.startswith("out_conv.0.weight"):
            out[name] = 1.0 + normal(name, shp, seed, std=0.1)
        else:
            out[name] = normal(name, shp, seed, std=0.1)
    return out
