"""Synthetic benchmark inputs: random-triangle soups and incoherent rays.

Both come from a counter-based generator, u(k) = (splitmix64(seed * 0x9E3779B97F4A7C15 + k) >> 11) * 2^-53,
so every process (and the CPU checker) regenerates identical inputs without files.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(seed, start, count):
    """u(k) for k = start .. start+count-1, float64 in [0, 1)."""
    with np.errstate(over="ignore"):
        k = np.arange(start, start + count, dtype=np.uint64)
        z = _mix(np.uint64(seed) * _GOLD + k)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def soup(n, sigma, seed=12345, chunk=1 << 20):
    """Soup(N, sigma, seed): triangle i has centre c_i ~ U[-1,1]^3 and vertices c_i + sigma*N(0, I3).

    Sixteen counters per triangle: 3 for the centre, 12 for six Box-Muller pairs (nine normals used).
    Returns the rt_mesh tables: one object, one face per triangle, a flat normal per face."""
    v = np.empty((n, 3, 3), dtype=np.float64)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        u = uniform(seed, s * 16, m * 16).reshape(m, 16)
        c = 2.0 * u[:, 0:3] - 1.0
        u1 = np.maximum(u[:, 3:9], 1e-300)
        u2 = u[:, 9:15]
        r = np.sqrt(-2.0 * np.log(u1))
        g = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)], axis=1)[:, :9]
        v[s:s + m] = c[:, None, :] + sigma * g.reshape(m, 3, 3)
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 1]
    nrm = np.cross(e1, e2)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    idx = np.arange(3 * n, dtype=np.int64)
    corners = np.stack([idx, np.full(3 * n, -1, dtype=np.int64), idx // 3], axis=1)
    return {
        "v": v.reshape(-1, 3), "vt": np.zeros((0, 3)), "vn": nrm, "corners": corners,
        "face_begin": np.arange(0, 3 * n + 1, 3, dtype=np.uint64),
        "object_begin": np.array([0, n], dtype=np.uint64), "names": ["soup"],
    }


def incoherent_rays(m, seed=777, t_min=1e-3, t_max=np.inf, start=0, chunk=1 << 22):
    """Rays(M, seed): origin U[-1,1]^3, direction uniform on the sphere (z = 1-2u1, phi = 2 pi u2).
    Six counters per ray (one spare).  float32 [M, 8] = ox oy oz tmin dx dy dz tmax."""
    out = np.empty((m, 8), dtype=np.float32)
    for s in range(0, m, chunk):
        k = min(chunk, m - s)
        u = uniform(seed, (start + s) * 6, k * 6).reshape(k, 6)
        z = 1.0 - 2.0 * u[:, 3]
        r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
        phi = 2.0 * np.pi * u[:, 4]
        out[s:s + k, 0:3] = 2.0 * u[:, 0:3] - 1.0
        out[s:s + k, 3] = t_min
        out[s:s + k, 4] = r * np.cos(phi)
        out[s:s + k, 5] = r * np.sin(phi)
        out[s:s + k, 6] = z
        out[s:s + k, 7] = t_max
    return out
