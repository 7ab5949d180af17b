"""Synthetic benchmark inputs: random-obstacle voxel grids and weld-point sets.

The occupancy is iid Bernoulli(occ_prob) from a splitmix64 stream in raster (z, y, x) order
(occupied iff u < occ_prob, u = top 53 bits / 2^53), then the 2x2x2 corner blocks at (0..1)^3 and
(n-2..n-1)^3 are forced free.  wall = 0, precision = 1, node coordinate = index.  This is input
generation only (numpy on the host); tests check it equals the oracle's generator bit for bit."""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_block(seed, count, start=0):
    """outputs start .. start+count-1 of the splitmix64 stream seeded with `seed`"""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_grid(n, seed=2024, occ_prob=0.10):
    """returns (free uint8[n^3], cx, cy, cz float32[n], precision=1.0, wall=0)"""
    tot = n * n * n
    free = np.empty(tot, np.uint8)
    step = 1 << 22
    for s in range(0, tot, step):
        c = min(step, tot - s)
        u = (splitmix64_block(seed, c, s) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        free[s:s + c] = (u >= occ_prob).astype(np.uint8)
    f3 = free.reshape(n, n, n)
    f3[:2, :2, :2] = 1
    f3[n - 2:, n - 2:, n - 2:] = 1
    ax = np.arange(n, dtype=np.float32)
    return free, ax, ax.copy(), ax.copy(), np.float32(1.0), 0


def synth_weld_points(free, n, count, seed=7):
    """`count` distinct free voxels chosen by the same PRNG (C5: 64 weld points); returns ids"""
    ids, k = [], 0
    seen = set()
    while len(ids) < count:
        v = int(splitmix64_block(seed, 1, k)[0] % np.uint64(n * n * n))
        k += 1
        if free[v] and v not in seen:
            seen.add(v)
            ids.append(v)
    return np.array(ids, np.int64)
