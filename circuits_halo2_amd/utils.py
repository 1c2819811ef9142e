"""Host-side helpers: seeded synthetic inputs and byte-layout utilities.
Nothing here computes the MSM/NTT path."""
from __future__ import annotations

import numpy as np

R_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_R_LIMBS = np.array([(R_MODULUS >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
DEFAULT_SEED = 0x53554D4D41  # "SUMMA"
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _splitmix_block(start_state: int, count: int) -> np.ndarray:
    """outputs 1..count of SplitMix64 started at start_state (vectorised)"""
    with np.errstate(over="ignore"):
        s = np.uint64(start_state & 0xFFFFFFFFFFFFFFFF) + _GOLD * np.arange(1, count + 1, dtype=np.uint64)
        z = (s ^ (s >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _lt_r(limbs: np.ndarray) -> np.ndarray:
    """limbs: (n,4) uint64 LE -> bool mask value < r"""
    lt = np.zeros(limbs.shape[0], dtype=bool)
    eq = np.ones(limbs.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (limbs[:, i] < _R_LIMBS[i])
        eq &= limbs[:, i] == _R_LIMBS[i]
    return lt


def _mix(s: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (s ^ (s >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_fr_canonical(seed: int, n: int) -> np.ndarray:
    """n uniform values in [0, r) as canonical 32-B little-endian integers (numpy uint8,
    length 32 n).  Same stream rule as the oracle's generators (oracle/pyref.py::random_fr):
    element i = SplitMix64 outputs 4i+1..4i+4 (limb 3 masked to 62 bits), a rejected candidate
    (~24 %) is redrawn from the stream seeded (seed ^ (i+1)) + (j << 32), j = 0, 1, ...
    Fully vectorised."""
    limbs = _splitmix_block(seed, 4 * n).reshape(n, 4).copy()
    limbs[:, 3] &= np.uint64((1 << 62) - 1)
    bad = np.nonzero(~_lt_r(limbs))[0]
    j = 0
    with np.errstate(over="ignore"):
        while bad.size:
            base = (np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (bad.astype(np.uint64) + np.uint64(1))) + np.uint64((j << 32) & 0xFFFFFFFFFFFFFFFF)
            w = _mix(base[:, None] + _GOLD * np.arange(1, 5, dtype=np.uint64)[None, :])
            w[:, 3] &= np.uint64((1 << 62) - 1)
            ok = _lt_r(w)
            limbs[bad[ok]] = w[ok]
            bad = bad[~ok]
            j += 1
    return limbs.view(np.uint8).reshape(-1)


def random_fr_secure(n: int) -> np.ndarray:
    """n uniform values in [0, r) from the OS entropy source (blinding factors of a proof): 254-bit candidates,
    rejection-sampled; canonical 32-B little-endian integers"""
    import os
    out = np.zeros((n, 4), dtype=np.uint64)
    todo = np.arange(n)
    while todo.size:
        cand = np.frombuffer(os.urandom(32 * todo.size), dtype=np.uint64).reshape(-1, 4).copy()
        cand[:, 3] &= np.uint64((1 << 62) - 1)
        ok = _lt_r(cand)
        out[todo[ok]] = cand[ok]
        todo = todo[~ok]
    return out.view(np.uint8).reshape(-1)


def to_montgomery_host(canon: np.ndarray) -> np.ndarray:
    """canonical -> Montgomery with Python integers (small inputs / CPU-only tests)."""
    out = bytearray(canon.size)
    raw = canon.tobytes()
    for i in range(0, len(raw), 32):
        v = (int.from_bytes(raw[i:i + 32], "little") << 256) % R_MODULUS
        out[i:i + 32] = v.to_bytes(32, "little")
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def ints_to_fr(values) -> np.ndarray:
    """list of Python ints -> Montgomery Fr buffer"""
    return np.frombuffer(b"".join(((v % R_MODULUS) << 256).__mod__(R_MODULUS).to_bytes(32, "little") for v in values),
                         dtype=np.uint8).copy()
