"""Seeded procedural scenes and host-side ray batches for tests and bench.py.

The scene files BASELINE.json names (Cornell box, Crytek Sponza, Conference, Hairball,
San Miguel) are not in the reference checkout (`.MISSING_LARGE_BLOBS` lists
data/models/Sponza/sponza.obj) and there is no network, so stand-ins with the same
triangle counts are generated from fixed seeds.  Everything is
float32 / int32 numpy; nothing here is on the timed path.
"""
import math

import numpy as np

from ._capi import RAY_DTYPE


# ------------------------------------------------------------------------------------
# mesh helpers
# ------------------------------------------------------------------------------------
class MeshBuilder:
    def __init__(self):
        self.pos = []
        self.tri = []
        self.nv = 0

    def add(self, pos, tri):
        pos = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
        tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
        self.pos.append(pos)
        self.tri.append(tri + self.nv)
        self.nv += pos.shape[0]

    def num_tris(self):
        return sum(t.shape[0] for t in self.tri)

    def finish(self):
        pos = np.concatenate(self.pos, axis=0).astype(np.float32)
        tri = np.concatenate(self.tri, axis=0).astype(np.int32)
        return tri, pos

    # parametric patch: f(u,v) -> xyz on an (nu+1) x (nv+1) lattice, 2*nu*nv triangles
    def patch(self, f, nu, nv, flip=False):
        u = np.linspace(0.0, 1.0, nu + 1)
        v = np.linspace(0.0, 1.0, nv + 1)
        uu, vv = np.meshgrid(u, v, indexing="ij")
        p = np.stack(f(uu, vv), axis=-1).reshape(-1, 3)
        i = np.arange(nu)[:, None] * (nv + 1) + np.arange(nv)[None, :]
        a, b, c, d = i, i + (nv + 1), i + (nv + 1) + 1, i + 1
        t0 = np.stack([a, b, c], -1).reshape(-1, 3)
        t1 = np.stack([a, c, d], -1).reshape(-1, 3)
        t = np.concatenate([t0, t1], 0)
        if flip:
            t = t[:, ::-1]
        self.add(p, t)

    def quad(self, p0, p1, p2, p3, nu=1, nv=1):
        p0, p1, p2, p3 = [np.asarray(p, dtype=np.float64) for p in (p0, p1, p2, p3)]

        def f(u, v):
            q = (p0[None, None] * ((1 - u) * (1 - v))[..., None] + p1[None, None] * (u * (1 - v))[..., None]
                 + p2[None, None] * (u * v)[..., None] + p3[None, None] * ((1 - u) * v)[..., None])
            return q[..., 0], q[..., 1], q[..., 2]
        self.patch(f, nu, nv)

    def box(self, lo, hi, n=1):
        x0, y0, z0 = lo
        x1, y1, z1 = hi
        self.quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), n, n)
        self.quad((x0, y0, z1), (x0, y1, z1), (x1, y1, z1), (x1, y0, z1), n, n)
        self.quad((x0, y0, z0), (x0, y1, z0), (x0, y1, z1), (x0, y0, z1), n, n)
        self.quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), n, n)
        self.quad((x0, y0, z0), (x0, y0, z1), (x1, y0, z1), (x1, y0, z0), n, n)
        self.quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), n, n)

    def cylinder(self, base, radius, height, seg, rings, profile=None):
        bx, by, bz = base

        def f(u, v):
            r = radius * (profile(v) if profile is not None else 1.0)
            ang = u * (2.0 * math.pi)
            return bx + r * np.cos(ang), by + v * height, bz + r * np.sin(ang)
        self.patch(f, seg, rings)

    def sphere(self, center, radius, nu, nv, bump=None):
        cx, cy, cz = center

        def f(u, v):
            th = u * (2.0 * math.pi)
            ph = v * math.pi
            r = radius * (1.0 + (bump(th, ph) if bump is not None else 0.0))
            return cx + r * np.sin(ph) * np.cos(th), cy + r * np.cos(ph), cz + r * np.sin(ph) * np.sin(th)
        self.patch(f, nu, nv)


# ------------------------------------------------------------------------------------
# scenes
# ------------------------------------------------------------------------------------
def cornell_box():
    """Procedural Cornell box: 5 walls + 2 boxes, ~555-unit cube (config 1)."""
    m = MeshBuilder()
    s = 555.0
    m.quad((0, 0, 0), (s, 0, 0), (s, 0, s), (0, 0, s))          # floor
    m.quad((0, s, 0), (0, s, s), (s, s, s), (s, s, 0))          # ceiling
    m.quad((0, 0, s), (s, 0, s), (s, s, s), (0, s, s))          # back wall
    m.quad((0, 0, 0), (0, 0, s), (0, s, s), (0, s, 0))          # left wall
    m.quad((s, 0, 0), (s, s, 0), (s, s, s), (s, 0, s))          # right wall
    m.box((130, 0, 65), (295, 165, 230))                        # short box
    m.box((265, 0, 295), (430, 330, 460))                       # tall box
    tri, pos = m.finish()
    cam = dict(eye=(278.0, 273.0, -800.0), target=(278.0, 273.0, 0.0), up=(0.0, 1.0, 0.0), fov_deg=40.0, far=3000.0)
    return tri, pos, cam


def random_soup(num_tris, seed, extent=10.0, size=0.6, walls=True):
    """N small random triangles in a box, optionally with 5 large wall quads."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, size=(num_tris, 1, 3))
    p = c + rng.normal(0.0, size, size=(num_tris, 3, 3))
    pos = p.reshape(-1, 3).astype(np.float32)
    tri = np.arange(num_tris * 3, dtype=np.int32).reshape(-1, 3)
    if walls:
        m = MeshBuilder()
        m.add(pos, tri)
        e = extent * 1.5
        m.quad((-e, -e, -e), (e, -e, -e), (e, -e, e), (-e, -e, e))
        m.quad((-e, -e, e), (e, -e, e), (e, e, e), (-e, e, e))
        m.quad((-e, -e, -e), (-e, -e, e), (-e, e, e), (-e, e, -e))
        m.quad((e, -e, -e), (e, e, -e), (e, e, e), (e, -e, e))
        m.quad((-e, e, -e), (-e, e, e), (e, e, e), (e, e, -e))
        tri, pos = m.finish()
    cam = dict(eye=(0.0, 0.0, -extent * 1.4), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov_deg=60.0,
               far=extent * 12.0)
    return tri, pos, cam


ATRIUM_TRIS = 262267  # triangle count of Crytek Sponza


def atrium(num_tris=ATRIUM_TRIS, seed=262267):
    """'atrium-262k': colonnaded two-storey hall standing in for Crytek Sponza (config 2).

    Long axis = x.  Mix of large architectural polygons and dense small detail
    (tessellated columns, arches, balusters, drapes, ornaments, foliage) like the original.
    Generated deterministically from `seed` to exactly `num_tris` triangles.
    """
    rng = np.random.default_rng(seed)
    m = MeshBuilder()
    X0, X1 = -1800.0, 1800.0      # hall length
    Z0, Z1 = -1100.0, 1100.0      # hall width (outer walls)
    A0, A1 = -520.0, 520.0        # open atrium strip between the colonnades
    Y1, Y2 = 520.0, 1000.0        # first-floor slab, roof line

    # floor, outer walls, aisle ceilings
    m.quad((X0, 0, Z0), (X1, 0, Z0), (X1, 0, Z1), (X0, 0, Z1), 72, 44)
    m.quad((X0, 0, Z0), (X0, Y2, Z0), (X1, Y2, Z0), (X1, 0, Z0), 36, 10)
    m.quad((X0, 0, Z1), (X1, 0, Z1), (X1, Y2, Z1), (X0, Y2, Z1), 36, 10)
    m.quad((X0, 0, Z0), (X0, 0, Z1), (X0, Y2, Z1), (X0, Y2, Z0), 22, 10)
    m.quad((X1, 0, Z0), (X1, Y2, Z0), (X1, Y2, Z1), (X1, 0, Z1), 22, 10)
    for (za, zb) in ((Z0, A0), (A1, Z1)):
        m.box((X0, Y1 - 40.0, za), (X1, Y1, zb), 12)             # first-floor slab over the aisles
        m.quad((X0, Y2, za), (X0, Y2, zb), (X1, Y2, zb), (X1, Y2, za), 36, 6)   # aisle roof

    # colonnades: two tiers of columns with capitals/bases + arches along both sides
    ncol = 12
    xs = np.linspace(X0 + 150.0, X1 - 150.0, ncol)

    def entasis(v):
        return 1.0 - 0.12 * v + 0.05 * np.sin(v * math.pi)

    for zc in (A0, A1):
        for (ybase, h, r) in ((0.0, Y1 - 100.0, 42.0), (Y1, Y2 - Y1 - 80.0, 34.0)):
            for x in xs:
                m.cylinder((x, ybase + 30.0, zc), r, h - 30.0, 28, 20, entasis)
                m.box((x - r * 1.4, ybase, zc - r * 1.4), (x + r * 1.4, ybase + 30.0, zc + r * 1.4), 2)
                m.box((x - r * 1.5, ybase + h, zc - r * 1.5), (x + r * 1.5, ybase + h + 40.0, zc + r * 1.5), 2)
            # arches between neighbouring columns
            for xa, xb in zip(xs[:-1], xs[1:]):
                cx, rad = 0.5 * (xa + xb), 0.5 * (xb - xa) - 30.0

                def arch(u, v, cx=cx, rad=rad, y=ybase + h + 40.0, zc=zc):
                    ang = u * math.pi
                    return cx - rad * np.cos(ang), y - 60.0 + 60.0 * np.sin(ang) * 1.0 + 0.0 * v, zc - 30.0 + 60.0 * v
                m.patch(arch, 24, 4)

    # balustrade along the first-floor edge: rail + turned balusters
    for zc in (A0 + 20.0, A1 - 20.0):
        m.box((X0 + 100.0, Y1 + 95.0, zc - 8.0), (X1 - 100.0, Y1 + 110.0, zc + 8.0), 6)

        def baluster(v):
            return 0.55 + 0.45 * np.sin(v * 3.0 * math.pi) ** 2
        for x in np.linspace(X0 + 120.0, X1 - 120.0, 110):
            m.cylinder((x, Y1, zc), 9.0, 95.0, 10, 12, baluster)

    # drapes hanging in the atrium (wavy cloth, dense regular tessellation)
    for k, x in enumerate(np.linspace(X0 + 500.0, X1 - 500.0, 6)):
        ph = rng.uniform(0.0, 2.0 * math.pi)
        zc = A0 + 60.0 if (k & 1) else A1 - 60.0

        def cloth(u, v, x=x, zc=zc, ph=ph):
            return (x - 220.0 + 440.0 * u, Y2 - 80.0 - 520.0 * v,
                    zc + 35.0 * np.sin(u * 9.0 * math.pi + ph) * (0.3 + v) + 12.0 * np.sin(v * 7.0 + ph))
        m.patch(cloth, 72, 56)

    # ornaments: bumpy "lion head" blobs on the end walls and vases on the floor
    for k in range(10):
        cx = X0 + 60.0 if k < 5 else X1 - 60.0
        cz = np.linspace(A0 + 80.0, A1 - 80.0, 5)[k % 5]
        f1, f2, ph = rng.integers(3, 9), rng.integers(2, 7), rng.uniform(0, 6.28)

        def bump(th, phi, f1=f1, f2=f2, ph=ph):
            return 0.18 * np.sin(f1 * th + ph) * np.sin(f2 * phi) + 0.05 * np.sin(17.0 * th) * np.sin(13.0 * phi)
        m.sphere((cx, 330.0, cz), 70.0, 56, 40, bump)
    for x in np.linspace(X0 + 400.0, X1 - 400.0, 8):
        for zc in (A0 + 140.0, A1 - 140.0):
            def vase(v):
                return 0.45 + 0.55 * np.sin(v * math.pi) ** 1.5 + 0.08 * np.sin(v * 23.0)
            m.cylinder((x, 0.0, zc), 55.0, 150.0, 32, 24, vase)

    # foliage: clusters of small leaf triangles above the vases
    budget = num_tris - m.num_tris()
    if budget < 0:
        raise ValueError("atrium: base geometry (%d tris) exceeds the requested %d" % (m.num_tris(), num_tris))
    nclusters = 16
    centers = [(x, 230.0, zc) for x in np.linspace(X0 + 400.0, X1 - 400.0, 8) for zc in (A0 + 140.0, A1 - 140.0)]
    per = [budget // nclusters + (1 if i < budget % nclusters else 0) for i in range(nclusters)]
    for (cx, cy, cz), n in zip(centers, per):
        if n == 0:
            continue
        c = np.array([cx, cy, cz]) + rng.normal(0.0, 1.0, size=(n, 1, 3)) * np.array([70.0, 60.0, 70.0])
        leaf = rng.normal(0.0, 9.0, size=(n, 3, 3))
        p = (c + leaf).reshape(-1, 3)
        m.add(p, np.arange(n * 3).reshape(-1, 3))

    tri, pos = m.finish()
    assert tri.shape[0] == num_tris, (tri.shape[0], num_tris)
    cam = dict(eye=(X0 + 260.0, 170.0, -40.0), target=(X1, 330.0, 60.0), up=(0.0, 1.0, 0.0), fov_deg=60.0,
               far=3.0 * float(np.linalg.norm([X1 - X0, Y2, Z1 - Z0])))
    return tri, pos, cam


def hairball(num_tris=2800000, seed=2800000):
    """'hairball' stand-in (config 4): thin random triangles along seeded, wobbling strands in a ball."""
    rng = np.random.default_rng(seed)
    strands = max(num_tris // 200, 1)
    per = (num_tris + strands - 1) // strands
    d = rng.normal(size=(strands, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = np.linspace(0.05, 1.0, per)[None, :, None]
    wob = rng.normal(0, 0.04, size=(strands, per, 3)).cumsum(axis=1) * 0.2
    c = (d[:, None, :] * t + wob).reshape(-1, 3)[:num_tris] * 100.0
    a = rng.normal(0, 0.15, size=(num_tris, 3))
    b = rng.normal(0, 0.6, size=(num_tris, 3))
    pos = np.stack([c, c + a, c + b], axis=1).reshape(-1, 3).astype(np.float32)
    tri = np.arange(num_tris * 3, dtype=np.int32).reshape(-1, 3)
    cam = dict(eye=(0.0, 0.0, -260.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov_deg=50.0, far=1000.0)
    return tri, pos, cam


def conference_room(num_tris=331000, seed=331000):
    """'room-331k' stand-in for Conference (config 3): the atrium generator at 331 k triangles."""
    return atrium(num_tris=num_tris, seed=seed)


def courtyard(num_tris=10000000, seed=10000000):
    """'courtyard-10M' stand-in for San Miguel (config 5): atrium architecture + ~9.8 M foliage triangles."""
    return atrium(num_tris=num_tris, seed=seed)


# ------------------------------------------------------------------------------------
# ray batches
# ------------------------------------------------------------------------------------
def pixel_table(w, h):
    """indexToPixel of PixelTable::recalculate (src/rt/ray/PixelTable.cpp:57-143):
    8x8 pixel blocks in Morton order (bit-swizzled inside a block), then the bottom and
    right edge stripes."""
    bh, bw = h & ~7, w & ~7
    maxdim = max(bw, bh)
    for s in (1, 2, 4, 8, 16):
        maxdim |= maxdim >> s
    maxdim = (maxdim + 1) >> 1
    w8, h8 = bw >> 3, bh >> 3
    i = np.arange(maxdim * maxdim, dtype=np.int64)
    tx = np.zeros_like(i)
    ty = np.zeros_like(i)
    for b in range(16):
        tx |= ((i >> (2 * b)) & 1) << b
        ty |= ((i >> (2 * b + 1)) & 1) << b
    keep = (tx < w8) & (ty < h8)
    tx, ty = tx[keep], ty[keep]
    inner = np.arange(64, dtype=np.int64)
    ix = ((inner & 1) >> 0) | ((inner & 4) >> 1) | ((inner & 16) >> 2)
    iy = ((inner & 2) >> 1) | ((inner & 8) >> 2) | ((inner & 32) >> 3)
    pos = (ty[:, None] * 8 + iy[None, :]) * w + (tx[:, None] * 8 + ix[None, :])
    parts = [pos.reshape(-1)]
    if bh < h:  # horizontal stripe below the bulk: px outer, py inner
        px, py = np.meshgrid(np.arange(bw), np.arange(bh, h), indexing="ij")
        parts.append((px + py * w).reshape(-1))
    if bw < w:  # vertical stripe + corner: py outer, px inner
        py, px = np.meshgrid(np.arange(h), np.arange(bw, w), indexing="ij")
        parts.append((px + py * w).reshape(-1))
    out = np.concatenate(parts).astype(np.int32)
    assert out.shape[0] == w * h
    return out


def camera_basis(cam):
    eye = np.asarray(cam["eye"], dtype=np.float64)
    fwd = np.asarray(cam["target"], dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(cam["up"], dtype=np.float64))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    return eye, fwd, right, up


def primary_rays(cam, w, h, order="pixeltable"):
    """Pinhole primary rays, one per pixel: origin = eye, normalised direction, tmin = 0,
    tmax = far (rayGenPrimaryKernel semantics, src/rt/ray/RayGenKernels.cu:77-125, with
    randomSeed = 0), emitted in PixelTable order.  Returns (rays[RAY_DTYPE], slotToID)."""
    eye, fwd, right, up = camera_basis(cam)
    idx = pixel_table(w, h) if order == "pixeltable" else np.arange(w * h, dtype=np.int32)
    px = (idx % w).astype(np.float64)
    py = (idx // w).astype(np.float64)
    nx = 2.0 * (px + 0.5) / w - 1.0
    ny = 2.0 * (py + 0.5) / h - 1.0
    th = math.tan(math.radians(cam["fov_deg"]) * 0.5)
    aspect = w / float(h)
    d = fwd[None, :] + (nx * th * aspect)[:, None] * right[None, :] - (ny * th)[:, None] * up[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(w * h, dtype=RAY_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = np.float32(eye[0]), np.float32(eye[1]), np.float32(eye[2])
    rays["tmin"] = 0.0
    rays["dx"], rays["dy"], rays["dz"] = d[:, 0].astype(np.float32), d[:, 1].astype(np.float32), d[:, 2].astype(np.float32)
    rays["tmax"] = np.float32(cam["far"])
    return rays, idx


def random_rays(n, seed, extent=10.0, tmax=1e30):
    """Incoherent rays with random origins/directions (parity stress)."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-extent, extent, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(n, dtype=RAY_DTYPE)
    for k, a in zip(("ox", "oy", "oz"), o.T):
        rays[k] = a.astype(np.float32)
    for k, a in zip(("dx", "dy", "dz"), d.T):
        rays[k] = a.astype(np.float32)
    rays["tmin"] = 0.0
    rays["tmax"] = np.float32(tmax)
    return rays


def box_rays(pos, n, seed, tmax=None):
    """Incoherent rays through a scene: origins uniform in its bounding box, directions uniform on the sphere, extent = the
    box diagonal unless given.  What a diffuse bounce looks like to the memory system: every ray in its own part of the BVH."""
    rng = np.random.default_rng(seed)
    lo, hi = pos.min(0).astype(np.float64), pos.max(0).astype(np.float64)
    o = lo + rng.uniform(0.0, 1.0, size=(n, 3)) * (hi - lo)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(n, dtype=RAY_DTYPE)
    for k, a in zip(("ox", "oy", "oz"), o.T):
        rays[k] = a.astype(np.float32)
    for k, a in zip(("dx", "dy", "dz"), d.T):
        rays[k] = a.astype(np.float32)
    rays["tmin"] = 0.0
    rays["tmax"] = np.float32(tmax if tmax is not None else np.linalg.norm(hi - lo))
    return rays


def nscreen_to_world(cam, w, h):
    """Row-major 4x4 mapping normalised screen (nx, ny, 0, 1) to a world point on the image plane
    at unit distance: the `nscreenToWorld` input of rayGenPrimaryKernel for this pinhole camera
    (the reference builds it as invert(fitToView * perspective * worldToCamera), Renderer.cpp:473-477)."""
    eye, fwd, right, up = camera_basis(cam)
    th = math.tan(math.radians(cam["fov_deg"]) * 0.5)
    aspect = w / float(h)
    m = np.zeros((4, 4), dtype=np.float64)
    m[:3, 0] = th * aspect * right
    m[:3, 1] = -th * up
    m[:3, 3] = eye + fwd
    m[3, 3] = 1.0
    return m.astype(np.float32)


def tri_normals(tri, pos):
    """Scene::getTriNormalBuffer contents (src/rt/Scene.cpp:127-134): normalised geometric normals."""
    a, b, c = pos[tri[:, 0]].astype(np.float64), pos[tri[:, 1]].astype(np.float64), pos[tri[:, 2]].astype(np.float64)
    n = np.cross(b - a, c - a)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-300), n)
    return n.astype(np.float32)
