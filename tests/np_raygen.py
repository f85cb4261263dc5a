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


def shadow_rays(in_rays, in_results, num_samples, light_pos, light_radius, kernel_seed, first=0, count=None):
    """rayGenShadowKernel (src/rt/ray/RayGenKernels.cu:240-301) in float64: (origins, unit directions, tmax) of count * num_samples rays."""
    count = in_rays.shape[0] - first if count is None else count
    r = in_rays[first:first + count]
    res = in_results[first:first + count]
    o = np.stack([r["ox"], r["oy"], r["oz"]], 1).astype(np.float64)
    d = np.stack([r["dx"], r["dy"], r["dz"]], 1).astype(np.float64)
    back = np.maximum(res["t"].astype(np.float64) - 1.0e-2, 0.0)
    origin = o + d * back[:, None]
    M = np.uint32
    with np.errstate(over="ignore"):
        a = (M(kernel_seed) + np.arange(count, dtype=np.uint32)).astype(np.uint32)
    b = np.full(count, 0x9e3779b9, dtype=np.uint32)
    c = np.full(count, 0x9e3779b9, dtype=np.uint32)
    a, b, c = _jenkins(a, b, c)
    a, b, c = _jenkins(a, b, c)
    off = np.stack([a, b, c], 1).astype(np.float32).astype(np.float64) * 2.0 ** -32    # (F32)hash * exp2(-32)
    ro = np.zeros((count, num_samples, 3))
    rd = np.zeros((count, num_samples, 3))
    rt = np.zeros((count, num_samples))
    lp = np.asarray(light_pos, dtype=np.float64)
    for i in range(num_samples):
        r1, r2, v1, v2, k = 0, 0, 1 << 31, 3 << 30, i
        while k:
            if k & 1:
                r1 ^= v1
                r2 ^= (v2 << 1) & 0xFFFFFFFF
            v1 |= v1 >> 1
            v2 ^= v2 >> 1
            k >>= 1
        pos = np.array([float(np.float32(r1)) * 2.0 ** -32, float(np.float32(r2)) * 2.0 ** -32, (i + 0.5) / num_samples])[None, :] + off
        pos = np.where(pos >= 1.0, pos - 1.0, pos)
        pos = pos * 2.0 - 1.0
        direction = lp[None, :] + light_radius * pos - origin
        length = np.linalg.norm(direction, axis=1)
        ro[:, i] = origin
        rd[:, i] = direction / length[:, None]
        rt[:, i] = np.where(res["id"] == -1, -1.0, length)
    return ro.reshape(-1, 3), rd.reshape(-1, 3), rt.reshape(-1)
