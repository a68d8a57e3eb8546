"""Synthetic control streams (oracle side, NumPy).  TEST INFRASTRUCTURE ONLY.

The reference has no synthetic-control generator for batched rollouts; its only
template is the AR(1) thruster command of
training/train_sim_brov2_koopmanEDMDc.py:160-164.  SURVEY.md §8(d) config 2 fixes the
stream used by the benchmark so that the HIP fill kernel (csrc/controls.hip) and this
file produce the same numbers:

  counter(b, t, j) = (b*T + t)*nu + j                      (layout independent)
  bits(c)          = splitmix64 output number c of the sequence seeded with `seed`
                   = mix(seed + (c+1)*0x9E3779B97F4A7C15)  (mod 2^64)
  uniform(c)       = (bits(c) >> 11) * 2^-53               in [0, 1)
  dist A (iid)     : u = 2*uniform - 1                     in [-1, 1)   -- bit exact
  dist B (AR(1))   : xi = sqrt(-2 ln(1-u1)) * cos(2 pi u2),
                     (u1, u2) = uniform(2c), uniform(2c+1) of a second stream
                     (seed ^ 0xA5A5A5A5A5A5A5A5);
                     u_t = clip(0.98*u_{t-1} + 0.02*xi_t, -1, 1), u_{-1} = 0
                     -- libm dependent, equal to the device fill only to ~1e-15.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
AR1_STREAM_XOR = 0xA5A5A5A5A5A5A5A5


def splitmix64_at(seed: int, counter: np.ndarray) -> np.ndarray:
    """Random-access splitmix64: output number `counter` (0-based) for `seed`."""
    with np.errstate(over="ignore"):
        c = np.asarray(counter, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (c + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform01_at(seed: int, counter: np.ndarray) -> np.ndarray:
    return (splitmix64_at(seed, counter) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def controls_iid(seed: int, b0: int, nb: int, T: int, nu: int = 8, t0: int = 0, nt: int = None) -> np.ndarray:
    """dist A, trajectories b0..b0+nb-1, steps t0..t0+nt-1, layout [nb, nt, nu]."""
    nt = T - t0 if nt is None else nt
    b = np.arange(b0, b0 + nb, dtype=np.uint64)[:, None, None]
    t = np.arange(t0, t0 + nt, dtype=np.uint64)[None, :, None]
    j = np.arange(nu, dtype=np.uint64)[None, None, :]
    c = (b * np.uint64(T) + t) * np.uint64(nu) + j
    return 2.0 * uniform01_at(seed, c) - 1.0


def controls_ar1(seed: int, b0: int, nb: int, T: int, nu: int = 8,
                 alpha: float = 0.98, sigma: float = 0.02) -> np.ndarray:
    """dist B (mirrors training/train_sim_brov2_koopmanEDMDc.py:160-164), [nb, T, nu]."""
    b = np.arange(b0, b0 + nb, dtype=np.uint64)[:, None, None]
    t = np.arange(T, dtype=np.uint64)[None, :, None]
    j = np.arange(nu, dtype=np.uint64)[None, None, :]
    c = (b * np.uint64(T) + t) * np.uint64(nu) + j
    s2 = seed ^ AR1_STREAM_XOR
    u1 = uniform01_at(s2, np.uint64(2) * c)
    u2 = uniform01_at(s2, np.uint64(2) * c + np.uint64(1))
    xi = np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * np.pi * u2)
    out = np.empty((nb, T, nu))
    prev = np.zeros((nb, nu))
    for k in range(T):
        prev = np.clip(alpha * prev + sigma * xi[:, k, :], -1.0, 1.0)
        out[:, k, :] = prev
    return out
