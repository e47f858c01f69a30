"""Closed-form synthetic weights (TEST INFRASTRUCTURE).

Parity tests at the real model sizes need the reference and the HIP path to
hold identical weights without committing 160 MB checkpoints.  Every tensor is
filled from a counter-based integer hash (splitmix64) of ``crc32(key)`` and the
element index, so the values are bit-reproducible on any platform and depend
only on the state-dict key and shape.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform_pm1(key: str, numel: int, salt: int = 0) -> np.ndarray:
    """float32 values in [-1, 1), exact multiples of 2^-23."""
    seed = np.uint64(zlib.crc32(key.encode()) + (salt << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(numel, dtype=np.uint64)
        h = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + _splitmix64(np.array([seed], dtype=np.uint64))[0])
    u24 = (h >> np.uint64(40)).astype(np.int64)          # 24 random bits
    return ((u24 - (1 << 23)).astype(np.float32)) / np.float32(1 << 23)


def make_tensor(key: str, shape, salt: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    numel = int(np.prod(shape)) if len(shape) else 1
    u = uniform_pm1(key, numel, salt)
    if key.endswith("gamma"):
        v = 1.0 + 0.25 * u
    elif key.endswith("relative_attention_bias.weight"):
        v = 0.75 * u
    elif key.endswith("null_cond_emb"):
        v = 1.7 * u
    elif key.endswith("bias"):
        v = 0.1 * u
    elif key.endswith("weight"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        if "ConvTranspose" in key:
            fan_in = shape[0]
        v = u * np.float32(np.sqrt(3.0 / max(fan_in, 1)) * 1.4)
    else:
        v = u
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).reshape(shape))


def fill_state_dict(template: Dict[str, torch.Tensor], salt: int = 0) -> Dict[str, torch.Tensor]:
    """Return a new state dict with every floating tensor of ``template`` replaced by
    generated values.  Non-trainable frequency buffers are kept as they are."""
    out = {}
    for k, t in template.items():
        if k.endswith("inv_freq") or k.endswith("freqs") or not torch.is_floating_point(t):
            out[k] = t.clone()
        else:
            out[k] = make_tensor(k, t.shape, salt).to(t.dtype)
    return out


def synthetic_input(key: str, shape, scale: float = 1.0) -> torch.Tensor:
    """Roughly unit-variance deterministic activations (sum of 3 uniforms)."""
    numel = int(np.prod(shape))
    u = uniform_pm1(key, numel, 1) + uniform_pm1(key, numel, 2) + uniform_pm1(key, numel, 3)
    return torch.from_numpy((u * np.float32(scale)).reshape(tuple(shape)))
