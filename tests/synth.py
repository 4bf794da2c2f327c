"""Deterministic synthetic inputs for the parity tests and the golden generator.

Everything is integer arithmetic on a splitmix64 counter stream followed by an exact
power-of-two scaling, so the same (seed, shape) gives the same fp32 bits on every machine,
numpy version and torch version -- fixtures store seeds instead of input tensors.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def normal_like(shape, seed: int) -> np.ndarray:
    """Bell-shaped fp32 values in (-4, 4), std ~1.15: sum of four 16-bit uniforms, centred,
    times 2**-15 (exact in fp32)."""
    n = int(np.prod(shape))
    w = _splitmix64(np.arange(n, dtype=np.uint64), seed)
    s = np.zeros(n, dtype=np.int64)
    for k in range(4):
        s += ((w >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.int64)
    s -= 131070
    return (s.astype(np.float32) * np.float32(2.0 ** -15)).reshape(shape)


def uniform01(shape, seed: int) -> np.ndarray:
    """fp32 uniform on [0,1) with 24-bit resolution (exact)."""
    n = int(np.prod(shape))
    w = _splitmix64(np.arange(n, dtype=np.uint64), seed)
    return ((w >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(shape)


def small_ints(shape, seed: int, lo: int = 1, hi: int = 5) -> np.ndarray:
    """fp32 integers in [lo, hi] (token sizes)."""
    n = int(np.prod(shape))
    w = _splitmix64(np.arange(n, dtype=np.uint64), seed)
    return (lo + (w % np.uint64(hi - lo + 1)).astype(np.int64)).astype(np.float32).reshape(shape)


def clustered(shape, seed: int, noise: float = 0.25) -> np.ndarray:
    """Keys that look like a real ViT layer's: every token = one shared direction plus a small
    perturbation, so cosine similarities crowd near 1 (mean node_max ~0.93)."""
    n, T, D = shape
    base = normal_like((n, 1, D), seed ^ 0x5EED)
    pert = normal_like((n, T, D), seed)
    return (base + np.float32(noise) * pert).astype(np.float32)


def to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16, returned as uint16 bit patterns (finite inputs)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16)


def bf16_round(a: np.ndarray) -> np.ndarray:
    """fp32 array whose values are exactly representable in bf16."""
    return (to_bf16_bits(a).astype(np.uint32) << np.uint32(16)).view(np.float32).reshape(a.shape)


def fill_parameters(module, base_seed: int, scale: float = 0.05) -> list:
    """Overwrite every parameter of a torch module with deterministic values keyed by its NAME (sorted),
    so two implementations with the same parameter names get bit-identical weights without shipping a
    state_dict.  LayerNorm-like gains are set around 1, everything else around 0.  Returns the names."""
    import zlib

    import torch
    names = []
    with torch.no_grad():
        for name, p in sorted(module.named_parameters(), key=lambda kv: kv[0]):
            vals = normal_like(tuple(p.shape), base_seed + zlib.crc32(name.encode()))
            is_gain = p.ndim == 1 and name.endswith("weight") and ("norm" in name)
            t = torch.from_numpy(vals * np.float32(0.1 if is_gain else scale))
            if is_gain:
                t = t + 1.0
            p.copy_(t.to(p.dtype))
            names.append(name)
    return names
