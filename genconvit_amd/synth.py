"""Deterministic synthetic weights / frames / eps (counter-based, no torch RNG).

There are no pretrained checkpoints in the container (reference ``weight/`` holds
only ``.gitkeep``; README.md:116-125 points to a download), so parity and
benchmarks run on synthetic "trained-like" parameters.

Every value is a pure function of ``(seed, tensor name, flat index)``:
``u64 = splitmix64(seed ^ H(name) + index)`` -> top 24 bits -> exact fp32 in
[0,1).  Only integer arithmetic plus one exact int->float conversion and one
fp32 multiply/add are involved for the uniform draws, so the numbers are
bit-identical here and on the GPU box.  Normal draws (VAE ``eps``) use
Box-Muller evaluated in float64 and rounded to fp32.
"""
from __future__ import annotations

import hashlib
import math

import numpy as np
import torch

DEFAULT_SEED = 0x47434F4E  # "GCON"

_M64 = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


def _s64(x: int) -> int:
    """two's-complement reinterpretation of a u64 python int as i64."""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _name_key(seed: int, name: str) -> int:
    h = int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], "little")
    return (seed * 0xD1342543DE82EF95 + h) & _M64


def _lsr(z: torch.Tensor, n: int) -> torch.Tensor:
    # logical shift right on int64 storage
    return (z >> n) & ((1 << (64 - n)) - 1)


def _splitmix_bits24(key: int, start: int, count: int, device="cpu") -> torch.Tensor:
    """top-24-bit splitmix64 output for counters start..start+count-1 (int64 tensor)."""
    idx = torch.arange(start, start + count, dtype=torch.int64, device=device)
    z = idx * _s64(_GOLDEN) + _s64(key)
    z = z + _s64(_GOLDEN)
    z = (z ^ _lsr(z, 30)) * _s64(_C1)
    z = (z ^ _lsr(z, 27)) * _s64(_C2)
    z = z ^ _lsr(z, 31)
    return _lsr(z, 40)


def splitmix_bits24_numpy(key: int, start: int, count: int) -> np.ndarray:
    """numpy-uint64 restatement of :func:`_splitmix_bits24` (used by tests)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start, start + count, dtype=np.uint64)
        z = idx * np.uint64(_GOLDEN) + np.uint64(key)
        z = z + np.uint64(_GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.int64)


_CHUNK = 1 << 24


def uniform01(name: str, numel: int, seed: int = DEFAULT_SEED, device="cpu") -> torch.Tensor:
    """fp32 uniform [0,1) of ``numel`` elements, deterministic in (seed, name)."""
    key = _name_key(seed, name)
    out = torch.empty(numel, dtype=torch.float32, device=device)
    for s in range(0, numel, _CHUNK):
        n = min(_CHUNK, numel - s)
        bits = _splitmix_bits24(key, s, n, device)
        out[s:s + n] = bits.to(torch.float32) * (1.0 / (1 << 24))
    return out


def uniform(name: str, shape, lo: float, hi: float, seed: int = DEFAULT_SEED, device="cpu") -> torch.Tensor:
    numel = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, numel, seed, device)
    u.mul_(float(hi - lo))
    u.add_(float(lo))
    return u.reshape(shape)


def normal(name: str, shape, seed: int = DEFAULT_SEED) -> torch.Tensor:
    """fp32 N(0,1): Box-Muller in float64 over two independent uniform streams."""
    numel = int(np.prod(shape))
    u1 = uniform01(name + "#bm1", numel, seed).double()
    u2 = uniform01(name + "#bm2", numel, seed).double()
    u1 = (u1 + 0.5 / (1 << 24))                       # (0,1): avoid log(0)
    r = torch.sqrt(-2.0 * torch.log(u1))
    z = r * torch.cos(2.0 * math.pi * u2)
    return z.float().reshape(shape)


def _fan_in(shape, kind):
    if kind == "linear":
        return shape[1]
    if kind == "conv":
        return shape[1] * shape[2] * shape[3]
    if kind == "convT":                               # (Cin, Cout, kh, kw), k == stride
        return shape[0]
    raise ValueError(kind)


def make_param(name: str, shape, kind: str, seed: int = DEFAULT_SEED, device="cpu") -> torch.Tensor:
    """"Trained-like" synthetic parameter: unit-gain weights, LN
    scale around 1, layer-scale gamma well above the 1e-6 init so every residual
    branch matters."""
    if kind in ("linear", "conv", "convT"):
        a = math.sqrt(3.0 / _fan_in(shape, kind))
        return uniform(name, shape, -a, a, seed, device)
    if kind == "bias":
        return uniform(name, shape, -0.035, 0.035, seed, device)
    if kind == "ln_w":
        return uniform(name, shape, 0.5, 1.5, seed, device)
    if kind == "gamma":
        return uniform(name, shape, 0.05, 0.5, seed, device)
    if kind == "bn_mean":
        return uniform(name, shape, -0.17, 0.17, seed, device)
    if kind == "bn_var":
        return uniform(name, shape, 0.5, 1.5, seed, device)
    if kind == "relpos":
        return uniform(name, shape, -0.035, 0.035, seed, device)
    raise ValueError(f"unknown parameter kind {kind!r}")


def make_state_dict(spec, seed: int = DEFAULT_SEED, tag: str = "", device="cpu"):
    """``{name: fp32 tensor}`` for a spec from :mod:`genconvit_amd.spec`.

    ``tag`` salts the stream so the ED and VAE networks get different backbones.
    """
    return {name: make_param(tag + name, shape, kind, seed, device) for name, shape, kind in spec}


# ImageNet statistics used by the reference's "vid" transform (dataset/loader.py:64-65,77)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def make_uint8_frames(batch: int, seed: int = DEFAULT_SEED, name: str = "frames") -> torch.Tensor:
    """uint8 (B,224,224,3) uniform[0,255] face-crop stand-ins."""
    key = _name_key(seed, name)
    n = batch * 224 * 224 * 3
    out = torch.empty(n, dtype=torch.uint8)
    for s in range(0, n, _CHUNK):
        m = min(_CHUNK, n - s)
        out[s:s + m] = (_splitmix_bits24(key, s, m) >> 16).to(torch.uint8)
    return out.reshape(batch, 224, 224, 3)


def make_frames(batch: int, seed: int = DEFAULT_SEED, name: str = "frames") -> torch.Tensor:
    """Normalised fp32 NCHW frames with ``preprocess_frame`` semantics
    (model/pred_func.py:95-108): uint8 NHWC -> float -> NCHW -> /255 -> (x-mean)/std."""
    u8 = make_uint8_frames(batch, seed, name)
    x = u8.float().permute(0, 3, 1, 2) / 255.0
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(1, 3, 1, 1)
    return ((x - mean) / std).contiguous()


def make_eps(batch: int, latent: int = 12544, seed: int = DEFAULT_SEED, name: str = "eps") -> torch.Tensor:
    """The VAE's single N(0,1) draw (genconvit_vae.py:46), made explicit."""
    return normal(name, (batch, latent), seed)
