"""Deterministic synthetic camera frames for tests and benchmarks.

Integer-only (a counter-based splitmix64 hash and an integer sine table), so the
build container and the GPU box produce bit-identical inputs from a seed.

  noise_frames   i.i.d. uniform uint16 (BASELINE config 0: 256x256x64, seed 1)
  plasma_frames  sum of 4 drifting sinusoids + sensor noise; 12-bit values in
                 the low bits of uint16 (use shift=4) or full 16-bit (shift=0).
                 Frame 0 doubles as the delta frame, as encode.cc does.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on a uint64 array (wraps mod 2^64)."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _hash_grid(seed, frame, npix):
    with np.errstate(over="ignore"):
        idx = np.arange(npix, dtype=np.uint64)
        key = np.uint64((seed * 0x100000001B3 + frame * 0x9E3779B1 + 12345) & 0xFFFFFFFFFFFFFFFF)
        return _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + key)


def noise_frames(n, xs, ys, seed=1):
    """Uniform 16-bit noise, shape (n, ys, xs) uint16."""
    out = np.empty((n, ys * xs), dtype=np.uint16)
    for f in range(n):
        out[f] = (_hash_grid(seed, f, xs * ys) >> np.uint64(24)).astype(np.uint16)
    return out.reshape(n, ys, xs)


def _isin_table():
    """int32 sine table, 65536 phases, amplitude 32767 (Bhaskara I, integers)."""
    half = 32768
    p = np.arange(half, dtype=np.int64)
    t = p * (half - p)
    s = (16 * t * 32767) // (5 * half * half - 4 * t)
    return np.concatenate([s, -s]).astype(np.int32)


_ISIN = None

# (x step, y step, frame step) of each sinusoid in 1/65536 turns:
# periods of about 23..51 pixels, about 0.05 rad per frame.
_WAVES = ((2849, 311, 521), (-463, 1771, -389), (1285, 1093, 457), (911, -1423, 263))


def plasma_frames(n, xs, ys, bits=12, seed=7, first=0):
    """Plasma-like frames (n, ys, xs) uint16, values in [0, 2^bits)."""
    global _ISIN
    if _ISIN is None:
        _ISIN = _isin_table()
    full = (1 << bits) - 1
    amp = (full * 95 // 100) // 8          # 4 waves of +-amp around mid-range
    k = 5 if bits <= 12 else 77            # noise: sum of 4 uniforms in [-k, k]
    x = np.arange(xs, dtype=np.int64)[None, :]
    y = np.arange(ys, dtype=np.int64)[:, None]
    out = np.empty((n, ys, xs), dtype=np.uint16)
    for f in range(n):
        t = first + f
        v = np.full((ys, xs), full // 2, dtype=np.int64)
        for (fx, fy, ft) in _WAVES:
            ph = (fx * x + fy * y + ft * t) & 0xFFFF
            v += (_ISIN[ph].astype(np.int64) * amp) >> 15
        h = _hash_grid(seed, t, xs * ys).reshape(ys, xs)
        m = np.uint64(2 * k + 1)
        for j in range(4):
            v += ((h >> np.uint64(16 * j)) & np.uint64(0xFFFF)).astype(np.int64) % int(m) - k
        out[f] = np.clip(v, 0, full).astype(np.uint16)
    return out
