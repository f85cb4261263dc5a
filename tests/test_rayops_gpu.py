"""reconstructKernel and the secondary-ray Morton sort vs numpy restatements (bit-exact: both are
plain IEEE binary32 arithmetic and integer work)."""
import numpy as np
import pytest

import ntrace_amd as nt
from ntrace_amd import scenes

pytestmark = pytest.mark.gpu
F = np.float32


def np_from_abgr(c):
    c = c.astype(np.uint32)
    k = F(1.0) / F(255.0)
    return np.stack([(c & 0xFF).astype(F) * k, ((c >> 8) & 0xFF).astype(F) * k, ((c >> 16) & 0xFF).astype(F) * k,
                     (c >> 24).astype(F) * k], -1).astype(F)


def np_to_abgr(v):
    b = (np.minimum(np.maximum(v, F(0)), F(1)) * F(255.0)).astype(np.uint32)
    return b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16) | (b[:, 3] << 24)


def np_reconstruct(ray_type, n_per, first, num, p_slot_to_id, p_res, b_id_to_slot, b_res, mat, shaded, pixels):
    bg = np.array([0.2, 0.4, 0.8, 1.0], dtype=F)
    for task in range(num):
        pslot = first + task
        pid = p_slot_to_id[pslot]
        slots = b_id_to_slot[pid:pid + n_per] if ray_type == 0 else b_id_to_slot[task * n_per:(task + 1) * n_per]
        col = np.zeros(4, dtype=F)
        for s in slots:
            tri = b_res["id"][s]
            if tri == -1:
                add = bg if ray_type == 0 else np.ones(4, dtype=F)
            elif ray_type == 1:
                add = np.array([0, 0, 0, 1], dtype=F)
            else:
                add = np_from_abgr(shaded[tri:tri + 1])[0]
            col = (col + add).astype(F)
        col = (col * (F(1.0) / F(n_per))).astype(F)
        ptri = p_res["id"][pslot]
        if ray_type == 1 and ptri == -1:
            col = bg.copy()
        if ray_type == 2:
            col = (col * (bg if ptri == -1 else np_from_abgr(mat[ptri:ptri + 1])[0])).astype(F)
        pixels[pid] = np_to_abgr(col[None])[0]
    return pixels


@pytest.mark.parametrize("ray_type", [0, 1, 2])
def test_reconstruct_matches_numpy(ray_type):
    import torch
    from gpu_util import up
    rng = np.random.default_rng(ray_type)
    w, h, ns, ntri = 24, 16, (1 if ray_type == 0 else 4), 50
    n = w * h
    slot_to_id = scenes.pixel_table(w, h)
    p_res = np.zeros(n, dtype=nt.RESULT_DTYPE)
    p_res["id"] = rng.integers(-1, ntri, n)
    first, num = (0, n) if ray_type == 0 else (64, 200)
    nb = n if ray_type == 0 else num * ns
    b_res = np.zeros(nb, dtype=nt.RESULT_DTYPE)
    b_res["id"] = rng.integers(-1, ntri, nb)
    b_id_to_slot = rng.permutation(nb).astype(np.int32)
    mat = rng.integers(0, 2 ** 32, ntri, dtype=np.uint64).astype(np.uint32)
    shaded = rng.integers(0, 2 ** 32, ntri, dtype=np.uint64).astype(np.uint32)
    d_pix = torch.full((n,), 0x11223344, dtype=torch.int32, device="cuda:0")
    bufs = [up(slot_to_id), up(p_res), up(b_id_to_slot), up(b_res), up(mat), up(shaded)]
    nt.reconstruct(ray_type, ns, first, num, *[b.data_ptr() for b in bufs], d_pix.data_ptr())
    torch.cuda.synchronize()
    got = d_pix.cpu().numpy().view(np.uint32)
    exp = np_reconstruct(ray_type, ns, first, num, slot_to_id, p_res, b_id_to_slot, b_res, mat, shaded,
                         np.full(n, 0x11223344, dtype=np.uint32))
    assert np.array_equal(got, exp)


def np_ray_box(rays):
    o = np.stack([rays["ox"], rays["oy"], rays["oz"]], 1).astype(F)
    d = np.stack([rays["dx"], rays["dy"], rays["dz"]], 1).astype(F)
    e = (o + d * rays["tmax"][:, None]).astype(F)
    return np.minimum(o.min(0), e.min(0)).astype(F), np.maximum(o.max(0), e.max(0)).astype(F)


def np_ray_keys(rays, box=None):
    """the 192-bit sort keys (RayBuffer.cpp:103-165) as Python integers; `box`: the (lo, hi) of the batch the rays are a sample of"""
    o = np.stack([rays["ox"], rays["oy"], rays["oz"]], 1).astype(F)
    d = np.stack([rays["dx"], rays["dy"], rays["dz"]], 1).astype(F)
    lo, hi = box if box is not None else np_ray_box(rays)
    with np.errstate(all="ignore"):
        a = ((o - lo) / (hi - lo)).astype(F)
        ln = np.sqrt(((d[:, 0] * d[:, 0]).astype(F) + (d[:, 1] * d[:, 1]).astype(F)).astype(F) + (d[:, 2] * d[:, 2]).astype(F)).astype(F)
        inv = (F(1.0) * (F(1.0) / ln)).astype(F)
        b = (((d * inv[:, None]).astype(F) + F(1.0)).astype(F) * F(0.5)).astype(F)
    comp = [(a[:, k] * F(256.0) * F(65536.0)).astype(F) for k in range(3)] + [(b[:, k] * F(32.0) * F(65536.0)).astype(F) for k in range(3)]
    comp = [c.astype(np.int64).astype(np.uint64) & 0xFFFFFFFF for c in comp]
    key = np.zeros(rays.shape[0], dtype=object)
    big = [int(0)] * rays.shape[0]
    for k in range(6):
        ck = comp[k]
        for i in range(32):
            bit = ((ck >> np.uint64(i)) & np.uint64(1)).astype(np.uint64)
            pos = k + 6 * i
            for r in np.nonzero(bit)[0]:
                big[r] |= 1 << pos
    return big


def test_ray_morton_sort_matches_numpy():
    import torch
    from gpu_util import up
    tri, pos, cam = scenes.random_soup(2000, seed=4)
    rays = scenes.random_rays(3000, seed=12, tmax=5.0)
    rays[100:110] = rays[50]          # identical rays: ties keep slot order
    n = rays.shape[0]
    slot_to_id = np.random.default_rng(1).permutation(n).astype(np.int32)
    d_in, d_s2i = up(rays), up(slot_to_id)
    d_out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0")
    d_i2s = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    d_s2i_out = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    sec = nt.ray_morton_sort(n, d_in.data_ptr(), d_s2i.data_ptr(), d_out.data_ptr(), d_i2s.data_ptr(), d_s2i_out.data_ptr())
    assert sec > 0
    keys = np_ray_keys(rays)
    order = sorted(range(n), key=lambda i: (keys[i], i))
    got = d_out.cpu().numpy().view(nt.RAY_DTYPE)
    assert np.array_equal(got, rays[order])
    s2i = d_s2i_out.cpu().numpy()
    assert np.array_equal(s2i, slot_to_id[order])
    assert np.array_equal(d_i2s.cpu().numpy()[s2i], np.arange(n))


def test_ray_morton_sort_large_batch_ticket_path():
    """3 M rays: more one-sweep tiles (1 465 of 2 048 keys) than the device holds at once, so the passes take their tiles by ticket
    (radix_sort.h onesweep_launch) -- the 2^20-ray batches of the frames never do.  Checked: the output is a permutation, the keys of
    sampled neighbours (numpy restatement, the whole batch's box) do not decrease, and rays with equal keys keep their slot order."""
    import torch
    from gpu_util import up
    n = 3_000_000
    rays = scenes.random_rays(n, seed=77, tmax=3.0)
    rays[1_000_000:1_000_600] = rays[17]          # 600 identical rays: ties keep slot order
    d_in = up(rays)
    d_id = torch.arange(n, dtype=torch.int32, device="cuda:0")
    d_out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0")
    d_i2s = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    d_s2i = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    nt.ray_morton_sort(n, d_in.data_ptr(), d_id.data_ptr(), d_out.data_ptr(), d_i2s.data_ptr(), d_s2i.data_ptr())
    s2i = d_s2i.cpu().numpy()
    assert np.array_equal(np.sort(s2i), np.arange(n, dtype=np.int32))
    assert np.array_equal(d_i2s.cpu().numpy()[s2i], np.arange(n))
    got = d_out.cpu().numpy().view(nt.RAY_DTYPE)
    assert np.array_equal(got, rays[s2i])
    box = np_ray_box(rays)
    rng = np.random.default_rng(5)
    at = np.unique(np.concatenate([rng.integers(0, n - 1, 1500), np.nonzero(s2i == 17)[0], np.nonzero(s2i == 17)[0] + 300]))
    at = at[at < n - 1]
    ka, kb = np_ray_keys(got[at], box), np_ray_keys(got[at + 1], box)
    for i, a, b in zip(at, ka, kb):
        assert a < b or (a == b and s2i[i] < s2i[i + 1]), int(i)
    tie = np.nonzero(s2i == 17)[0][0]
    assert np.array_equal(s2i[tie + 1:tie + 601], np.arange(1_000_000, 1_000_600))


def test_scratch_is_kept_between_sorts_and_builds_and_can_be_returned():
    """ntr_ray_morton_sort and ntr_lbvh_build keep their temporaries per device between calls; ntr_lbvh_release_workspace returns both, and
    the next call allocates again: same order, same tree, before and after, and with a larger batch in between (the scratch grows)."""
    import torch
    from gpu_util import up
    tri, pos, cam = scenes.random_soup(9000, seed=21)
    d_tri, d_pos = up(tri), up(pos)
    capn, capw, capi = nt.lbvh_capacity(tri.shape[0])
    bufs = [torch.zeros(c, dtype=torch.uint8, device="cuda:0") for c in (capn, capw, capi)]
    mn, mx = pos.min(0), pos.max(0)

    def build():
        r = nt.lbvh_build(tri.shape[0], d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, bufs[0].data_ptr(), capn, bufs[1].data_ptr(), capw,
                          bufs[2].data_ptr(), capi)
        torch.cuda.synchronize()
        return [b[:k].clone() for b, k in zip(bufs, (r.nodesBytes, r.triWoopBytes, r.triIndexBytes))]

    def sort(n, seed):
        rays = scenes.random_rays(n, seed=seed, tmax=6.0)
        d_in = up(rays)
        ident = torch.arange(n, dtype=torch.int32, device="cuda:0")
        d_out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0")
        a = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        b = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        nt.ray_morton_sort(n, d_in.data_ptr(), ident.data_ptr(), d_out.data_ptr(), a.data_ptr(), b.data_ptr())
        return b.cpu().numpy()

    t0, s0 = build(), sort(5000, 1)
    big = sort(70000, 2)                      # the scratch grows
    assert np.array_equal(np.sort(big), np.arange(70000))
    assert np.array_equal(sort(5000, 1), s0)  # ... and a smaller batch in the larger scratch sorts as before
    nt.lbvh_release_workspace()
    t1, s1 = build(), sort(5000, 1)
    assert np.array_equal(s1, s0)
    for x, y in zip(t0, t1):
        assert torch.equal(x, y)
    nt.lbvh_release_workspace()


def test_sorted_ao_batch_traces_to_same_hits():
    """Sorting changes slots, not results: per ray id the hit record is identical."""
    import torch
    from gpu_util import DeviceBvh, gpu_trace, up
    tri, pos, cam = scenes.random_soup(8000, seed=9)
    dbvh = DeviceBvh(nt.sah_build(tri, pos))
    rays = scenes.random_rays(20000, seed=3, tmax=8.0)
    n = rays.shape[0]
    ident = np.arange(n, dtype=np.int32)
    d_in, d_id = up(rays), up(ident)
    d_out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0")
    d_i2s = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    d_s2i = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    nt.ray_morton_sort(n, d_in.data_ptr(), d_id.data_ptr(), d_out.data_ptr(), d_i2s.data_ptr(), d_s2i.data_ptr())
    sorted_rays = d_out.cpu().numpy().view(nt.RAY_DTYPE)
    a, _ = gpu_trace("fermi_speculative_while_while", dbvh, rays, True)
    b, _ = gpu_trace("fermi_speculative_while_while", dbvh, sorted_rays, True)
    i2s = d_i2s.cpu().numpy()
    assert np.array_equal(a["id"], b["id"][i2s]) and np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32)[i2s])
