"""numpy restatement of rayGenAOKernel (src/rt/ray/RayGenKernels.cu:129-236) for testing the
device ray generators to a floating-point tolerance (the reference builds these kernels with
-use_fast_math, so no bit-exact target exists for them)."""
import numpy as np


def _jenkins(a, b, c):
    M = np.uint32
    with np.errstate(over="ignore"):
        a = a - b; a = a - c; a = a ^ (c >> M(13))
        b = b - c; b = b - a; b = b ^ (a << M(8))
        c = c - a; c = c - b; c = c ^ (b >> M(13))
        a = a - b; a = a - c; a = a ^ (c >> M(12))
        b = b - c; b = b - a; b = b ^ (a << M(16))
        c = c - a; c = c - b; c = c ^ (b >> M(5))
        a = a - b; a = a - c; a = a ^ (c >> M(3))
        b = b - c; b = b - a; b = b ^ (a << M(10))
        c = c - a; c = c - b; c = c ^ (b >> M(15))
    return a, b, c


def ao_rays(in_rays, in_results, normals, num_samples, max_dist, kernel_seed, first=0, count=None):
    count = in_rays.shape[0] - first if count is None else count
    r = in_rays[first:first + count]
    res = in_results[first:first + count]
    o = np.stack([r["ox"], r["oy"], r["oz"]], 1).astype(np.float64)
    d = np.stack([r["dx"], r["dy"], r["dz"]], 1).astype(np.float64)
    back = np.maximum(res["t"].astype(np.float64) - 1.0e-4, 0.0)
    origin = o + d * back[:, None]
    tri = res["id"]
    n = np.where((tri != -1)[:, None], normals[np.maximum(tri, 0)].astype(np.float64), np.array([1.0, 0.0, 0.0]))
    flip = (n * d).sum(1) > 0
    n[flip] = -n[flip]
    na = np.abs(n)
    nm = na.max(1)
    perp = np.stack([n[:, 1], -n[:, 0], np.zeros(count)], 1)
    zc = nm == na[:, 2]
    xc = ~zc & (nm == na[:, 0])
    perp[zc] = np.stack([np.zeros(count), n[:, 2], -n[:, 1]], 1)[zc]
    perp[xc] = np.stack([-n[:, 2], np.zeros(count), n[:, 0]], 1)[xc]
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    biperp = np.cross(n, perp)
    task = np.arange(count, dtype=np.uint32)
    a, b, c = _jenkins(np.uint32(kernel_seed) + task, np.full(count, 0x9e3779b9, np.uint32), np.full(count, 0x9e3779b9, np.uint32))
    a, b, c = _jenkins(a, b, c)
    angle = 2.0 * np.pi * c.astype(np.float32).astype(np.float64) * 2.0 ** -32
    t0 = perp * np.cos(angle)[:, None] + biperp * np.sin(angle)[:, None]
    t1 = perp * -np.sin(angle)[:, None] + biperp * np.cos(angle)[:, None]
    out_o = np.repeat(origin, num_samples, axis=0)
    out_d = np.zeros((count * num_samples, 3))
    for i in range(num_samples):
        x, xadd, h2 = 0.0, 1.0, i + 1
        while h2:
            xadd *= 0.5
            if h2 & 1:
                x += xadd
            h2 >>= 1
        y, yadd, h3 = 0.0, 1.0, i + 1
        while h3:
            yadd *= 1.0 / 3.0
            y += (h3 % 3) * yadd
            h3 //= 3
        ang = 2.0 * np.pi * y
        rr = np.sqrt(x)
        sx, sy = rr * np.cos(ang), rr * np.sin(ang)
        sz = np.sqrt(max(1.0 - sx * sx - sy * sy, 0.0))
        v = sx * t0 + sy * t1 + sz * n
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        out_d[i::num_samples] = v
    tmax = np.repeat(np.where(tri == -1, -1.0, max_dist), num_samples)
    return out_o, out_d, tmax
