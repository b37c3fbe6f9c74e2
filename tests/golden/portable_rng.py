"""Portable, version-independent pseudo-random inputs for fixtures whose inputs are too large to
commit (the large-codebook VQ case).  Pure integer hashing (splitmix64 finaliser) on uint64 with
wrap-around, so the same bytes come out under any numpy/torch/libm."""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform01(n: int, seed: int) -> np.ndarray:
    """n float32 in [0,1) with 24 random bits each."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x51ED27)) * _G
    bits = (_mix(z) >> np.uint64(40)).astype(np.float64)
    return (bits / 16777216.0).astype(np.float32)


def normalish(n: int, seed: int) -> np.ndarray:
    """Roughly N(0,1): centred sum of 12 uniforms (only exact float64 additions)."""
    acc = np.zeros(n, dtype=np.float64)
    for j in range(12):
        acc += uniform01(n, seed * 131 + j).astype(np.float64)
    return (acc - 6.0).astype(np.float32)


def vq_case(N: int, D: int, K: int, seed: int):
    """X ~ N(0,1) rows and a default-init-scale codebook U(-1/K, 1/K) (models.py:125)."""
    x = normalish(N * D, seed).reshape(N, D)
    e = ((uniform01(K * D, seed + 7919).astype(np.float64) * 2.0 - 1.0) / K).astype(np.float32).reshape(K, D)
    return x, e
