"""Device ray generators vs numpy restatements (floating-point tolerance 1e-5: the reference
compiles these kernels with -use_fast_math, so there is no bit-exact target)."""
import numpy as np
import pytest

import ntrace_amd as nt
from ntrace_amd import scenes

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_pixel_table_and_primary_rays():
    import torch
    tri, pos, cam = scenes.cornell_box()
    for (w, h) in ((64, 48), (70, 45), (1920, 1080)):
        d_tab = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
        d_inv = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
        nt.pixel_table(w, h, d_tab.data_ptr(), d_inv.data_ptr())
        tab = d_tab.cpu().numpy()
        assert np.array_equal(tab, scenes.pixel_table(w, h))
        assert np.array_equal(d_inv.cpu().numpy()[tab], np.arange(w * h))
        d_rays = torch.zeros(w * h * 8, dtype=torch.float32, device="cuda:0")
        d_i2s = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
        d_s2i = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
        nt.raygen_primary(d_rays.data_ptr(), d_i2s.data_ptr(), d_s2i.data_ptr(), d_tab.data_ptr(), cam["eye"],
                          scenes.nscreen_to_world(cam, w, h), w, h, cam["far"])
        torch.cuda.synchronize()
        got = d_rays.cpu().numpy().reshape(-1, 8)
        ref, idx = scenes.primary_rays(cam, w, h)
        refa = ref.view(np.float32).reshape(-1, 8)
        assert np.abs(got - refa).max() < TOL * max(1.0, np.abs(refa[:, :3]).max())
        assert np.array_equal(d_s2i.cpu().numpy(), idx)
        assert np.array_equal(d_i2s.cpu().numpy()[idx], np.arange(w * h))
        assert np.allclose(np.linalg.norm(got[:, 4:7], axis=1), 1.0, atol=1e-5)


def test_ao_rays_match_numpy_restatement():
    import torch
    import np_raygen
    from gpu_util import DeviceBvh, gpu_trace, up
    tri, pos, cam = scenes.random_soup(5000, seed=3)
    dbvh = DeviceBvh(nt.sah_build(tri, pos))
    rays, _ = scenes.primary_rays(cam, 96, 64)
    res, _ = gpu_trace("fermi_speculative_while_while", dbvh, rays, False)
    normals = scenes.tri_normals(tri, pos)
    ns, first, count, seed, maxd = 8, 128, 4000, 0x12345678, 5.0
    d_out = torch.zeros(count * ns * 8, dtype=torch.float32, device="cuda:0")
    d_a = torch.zeros(count * ns, dtype=torch.int32, device="cuda:0")
    d_b = torch.zeros(count * ns, dtype=torch.int32, device="cuda:0")
    d_rays, d_res, d_nrm = up(rays), up(res), up(normals)
    nt.raygen_ao(d_out.data_ptr(), d_a.data_ptr(), d_b.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(),
                 first, count, ns, maxd, seed)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().reshape(-1, 8)
    ro, rd, rt = np_raygen.ao_rays(rays, res, normals, ns, maxd, seed, first, count)
    assert np.abs(got[:, :3] - ro).max() < 1e-4 * max(1.0, np.abs(ro).max())
    assert np.abs(got[:, 4:7] - rd).max() < 5e-5
    assert np.array_equal(got[:, 7], rt.astype(np.float32))
    assert (got[:, 3] == 0).all()
    assert np.array_equal(d_a.cpu().numpy(), np.arange(count * ns))
    assert nt.count_hits(d_res.data_ptr(), rays.shape[0]) == int((res["id"] >= 0).sum())


def test_shadow_rays_match_numpy_restatement():
    """rayGenShadowKernel (RayGenKernels.cu:240-301): per input ray numSamples rays towards quasi-random points around the light; rays of
    missed inputs are degenerate; the batch traces as an any-hit batch and equals the oracle's records."""
    import torch
    import np_raygen
    from gpu_util import DeviceBvh, assert_parity, gpu_trace, up
    from oracle import oracle
    tri, pos, cam = scenes.random_soup(5000, seed=3)
    dbvh = DeviceBvh(nt.sah_build(tri, pos))
    rays, _ = scenes.primary_rays(cam, 96, 64)
    res, _ = gpu_trace("fermi_speculative_while_while", dbvh, rays, False)
    ns, first, count, seed = 6, 100, 4100, 0x9ABCDEF1
    res = res.copy()
    res["id"][first + 5::7] = -1          # some inputs missed: their shadow rays must be degenerate (tmax = -1)
    light, radius = (float(pos[:, 0].mean()), float(pos[:, 1].max()) * 0.9, float(pos[:, 2].mean())), 0.75
    d_out = torch.zeros(count * ns * 8, dtype=torch.float32, device="cuda:0")
    d_a = torch.zeros(count * ns, dtype=torch.int32, device="cuda:0")
    d_b = torch.zeros(count * ns, dtype=torch.int32, device="cuda:0")
    d_rays, d_res = up(rays), up(res)
    nt.raygen_shadow(d_out.data_ptr(), d_a.data_ptr(), d_b.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), first, count, ns, light, radius, seed)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().reshape(-1, 8)
    ro, rd, rt = np_raygen.shadow_rays(rays, res, ns, light, radius, seed, first, count)
    scale = max(1.0, np.abs(ro).max())
    assert np.abs(got[:, :3] - ro).max() < 1e-4 * scale
    assert np.abs(got[:, 4:7] - rd).max() < 5e-5
    miss = rt < 0
    assert miss.any() and (~miss).any()
    assert np.array_equal(got[miss, 7], np.full(int(miss.sum()), -1.0, dtype=np.float32))
    assert np.abs(got[~miss, 7] - rt[~miss]).max() < 1e-4 * scale
    assert (got[:, 3] == 0).all() and np.allclose(np.linalg.norm(got[:, 4:7], axis=1), 1.0, atol=1e-5)
    assert np.array_equal(d_a.cpu().numpy(), np.arange(count * ns)) and np.array_equal(d_b.cpu().numpy(), np.arange(count * ns))
    srays = d_out.cpu().numpy().view(nt.RAY_DTYPE).reshape(-1)
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, srays, any_hit=True, threads=8)
    for kernel in nt.KERNELS:
        g, _ = gpu_trace(kernel, dbvh, srays, True)
        assert_parity(g, ref, "shadow batch, " + kernel)
