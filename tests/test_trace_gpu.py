"""GPU parity: HIP tracer (through the C-ABI) vs the CPU oracle, bit-exact on (id, t)."""
import numpy as np
import pytest

import ntrace_amd as nt
from ntrace_amd import scenes
from oracle import oracle
from ray_sets import edge_rays

pytestmark = pytest.mark.gpu

KERNELS = list(nt.KERNELS)   # every kernel name the reference selects by, the tesla_* while-while persistent bodies included


@pytest.fixture(scope="module")
def soup():
    from gpu_util import DeviceBvh
    tri, pos, cam = scenes.random_soup(20000, seed=11)
    bvh = nt.sah_build(tri, pos)
    return DeviceBvh(bvh), cam


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("any_hit", [False, True])
def test_soup_primary_and_random(soup, kernel, any_hit):
    from gpu_util import assert_parity, gpu_trace
    dbvh, cam = soup
    rays = np.concatenate([scenes.primary_rays(cam, 256, 256)[0], scenes.random_rays(50000, seed=3)])
    got, _ = gpu_trace(kernel, dbvh, rays, any_hit)
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=any_hit, threads=8)
    assert_parity(got, ref, "%s anyHit=%d" % (kernel, any_hit))


@pytest.mark.parametrize("kernel", KERNELS)
def test_cornell(kernel):
    from gpu_util import DeviceBvh, assert_parity, gpu_trace
    tri, pos, cam = scenes.cornell_box()
    dbvh = DeviceBvh(nt.sah_build(tri, pos))
    rays, _ = scenes.primary_rays(cam, 200, 120)
    for any_hit in (False, True):
        got, _ = gpu_trace(kernel, dbvh, rays, any_hit)
        ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=any_hit)
        assert_parity(got, ref, "cornell %s anyHit=%d" % (kernel, any_hit))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("any_hit", [False, True])
def test_edge_rays(soup, kernel, any_hit):
    from gpu_util import assert_parity, gpu_trace
    dbvh, _ = soup
    rays = edge_rays()
    got, _ = gpu_trace(kernel, dbvh, rays, any_hit)
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=any_hit)
    assert_parity(got, ref, "edge %s anyHit=%d" % (kernel, any_hit))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 257, 1000])
def test_ragged_sizes(soup, kernel, n):
    from gpu_util import assert_parity, gpu_trace
    dbvh, cam = soup
    rays = scenes.random_rays(max(n, 1), seed=n)[:n]
    got, sec = gpu_trace(kernel, dbvh, rays, False)
    if n == 0:
        assert sec == 0.0  # CudaBVHTracer.cpp:92-94
        return
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays)
    assert_parity(got, ref, "ragged %s n=%d" % (kernel, n))


def test_full_size_properties():
    """BASELINE size (1920x1080 primary on atrium-262k): size-independent properties --
    persistent == per-ray kernel bit for bit, any-hit hits iff closest-hit hits, t within
    [tmin, tmax], and EVERY record (closest hit and any hit) checked against the oracle on all host cores."""
    from gpu_util import DeviceBvh, assert_parity, gpu_trace
    tri, pos, cam = scenes.atrium()
    dbvh = DeviceBvh(nt.sah_build(tri, pos))
    rays, _ = scenes.primary_rays(cam, 1920, 1080)
    a, _ = gpu_trace("fermi_speculative_while_while", dbvh, rays, False)
    b, _ = gpu_trace("kepler_dynamic_fetch", dbvh, rays, False)
    assert_parity(a, b, "perray vs persistent")
    ah, _ = gpu_trace("kepler_dynamic_fetch", dbvh, rays, True)
    assert np.array_equal(ah["id"] >= 0, a["id"] >= 0)
    hit = a["id"] >= 0
    assert (a["t"][hit] > rays["tmin"][hit]).all() and (a["t"][hit] < rays["tmax"][hit]).all()
    assert np.array_equal(a["t"][~hit].view(np.uint32), rays["tmax"][~hit].view(np.uint32))
    import os
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, threads=os.cpu_count() or 8)
    assert_parity(a, ref, "atrium 1080p primary, every record")
    refa, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=True, threads=os.cpu_count() or 8)
    assert_parity(ah, refa, "atrium 1080p any hit, every record")


@pytest.mark.parametrize("any_hit", [False, True])
def test_gpu_traversal_counters_equal_oracle(soup, any_hit):
    """The instrumented kernel visits exactly the nodes / triangles the CPU tracer visits
    (same per-ray order), so the RayStats counters -- and hence the algorithmic bytes used
    for the roofline -- are identical."""
    import torch
    from gpu_util import up
    dbvh, cam = soup
    rays = np.concatenate([scenes.primary_rays(cam, 128, 128)[0], scenes.random_rays(20000, seed=8)])
    d_rays = up(rays)
    d_res = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device="cuda:0")
    st = dbvh.view.trace_stats("kepler_dynamic_fetch", rays.shape[0], any_hit, d_rays.data_ptr(), d_res.data_ptr())
    ref, rst = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=any_hit, threads=8)
    assert st.as_dict() == {k: v for k, v in rst.as_dict().items() if k != "maxStackDepth"}
    assert st.algorithmic_bytes() == rst.algorithmic_bytes()
    got = d_res.cpu().numpy().view(nt.RESULT_DTYPE)
    assert np.array_equal(got["id"], ref["id"])


@pytest.mark.parametrize("kernel", KERNELS)
def test_golden_fixtures_on_gpu(kernel):
    import os
    from gpu_util import DeviceBvh, assert_parity, gpu_trace
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for f in sorted(x for x in os.listdir(gold) if x.endswith(".npz")):
        g = np.load(os.path.join(gold, f))
        dbvh = DeviceBvh(nt.HostBvh(g["nodes"], g["woop"], g["tri_index"]))
        rays = g["rays"].view(nt.RAY_DTYPE).reshape(-1)
        for any_hit, key in ((False, "closest"), (True, "any")):
            got, _ = gpu_trace(kernel, dbvh, rays, any_hit)
            assert_parity(got, g["res_" + key].view(nt.RESULT_DTYPE).reshape(-1), "%s %s %s" % (f, kernel, key))


def test_fast_division_equals_hardware_division():
    """FAST path precondition: inside the FASTDIV range the refactored divide (per-ray correctly rounded
    reciprocal + 3 ops per quotient) is bit-identical to the hardware `/`; and the hardware `/`
    is IEEE correctly rounded (checked against numpy binary32 division on a subsample)."""
    import torch
    rng = np.random.default_rng(123)

    def rnd(n, emin, emax):
        m = rng.uniform(1.0, 2.0, n)
        e = rng.integers(emin, emax + 1, n)
        s = rng.choice([-1.0, 1.0], n)
        return (s * m * np.exp2(e.astype(np.float64))).astype(np.float32)
    x = np.concatenate([rnd(60000, -93, 55), rnd(20000, -10, 12), np.array([0.0, -0.0, 2.0 ** -93, 2.0 ** 55 * 1.999], np.float32),
                        (rng.integers(1, 2 ** 24, 5000) * 2.0 ** -10).astype(np.float32)])
    d = np.concatenate([rnd(1500, -40, 19), rnd(500, -3, 3), np.array([2.0 ** -40, 2.0 ** 20, -1.0, 1.0, 3.0, 1.0 / 3.0], np.float32),
                        np.nextafter(np.float32(2.0), np.float32(0.0)).reshape(1)])
    d_x, d_d = torch.from_numpy(x).cuda(), torch.from_numpy(d).cuda()
    assert nt.selftest_division(d_x.data_ptr(), x.shape[0], d_d.data_ptr(), d.shape[0]) == 0
    # hardware `/` itself vs numpy binary32 division (torch's elementwise divide is the same instruction sequence)
    q = (d_x[:20000, None] / d_d[None, :256]).cpu().numpy()
    with np.errstate(all="ignore"):
        ref = x[:20000, None] / d[None, :256]
    assert np.array_equal(q.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("x_exp,d_exp", [(0, 0), (-84, -20), (51, 16), (10, -40), (-30, 12)])
def test_fast_division_on_the_hardest_quotients(x_exp, d_exp):
    """One residual correction is enough only because the reciprocal is correctly rounded -- and only if NO quotient close to a rounding
    boundary goes wrong.  Those quotients are enumerable (scripts/studies/div_one_correction_check.py checks all 46.5 M significand pairs
    in exact integer arithmetic); here the device forms every one of them, at 16 exponent combinations and both signs, and compares the
    FAST divide with the hardware `/`."""
    pairs, bad = nt.selftest_division_hard(x_exp, d_exp)
    assert pairs == 46_517_418 * 32, pairs      # every significand pair of the exact check x 16 exponent pairs x 2 signs
    assert bad == 0, (pairs, bad)


def test_hard_quotient_enumeration_matches_the_exact_check():
    """The device enumerates the near-midpoint significand pairs itself: their number must be the one the exact-arithmetic script finds
    (profiles/r04_div_one_correction_check.txt: 46 517 418)."""
    pairs, bad = nt.selftest_division_hard(-3, 5)
    assert bad == 0
    assert pairs == 46_517_418 * 32


@pytest.mark.parametrize("kernel", KERNELS)
def test_generic_and_fast_paths_agree(soup, kernel):
    """bvh_flags = 0 forces the GENERIC path; both must give identical hit records."""
    from gpu_util import assert_parity, gpu_trace
    dbvh, cam = soup
    assert dbvh.flags & nt.BVH_FASTDIV
    rays = np.concatenate([scenes.primary_rays(cam, 200, 200)[0], scenes.random_rays(30000, seed=13)])
    for any_hit in (False, True):
        a, _ = gpu_trace(kernel, dbvh, rays, any_hit)
        b, _ = gpu_trace(kernel, dbvh, rays, any_hit, flags=0)
        assert_parity(a, b, "fast vs generic %s" % kernel)


@pytest.mark.parametrize("any_hit", [False, True])
def test_sched_hint_changes_order_only(soup, any_hit):
    """ntr_trace_bvh_hinted: every generation of the hint (cost recording launch, first derived order,
    refreshed orders), a rebind to another ray count and a reset give the oracle's hit records."""
    import torch
    from gpu_util import assert_parity, up
    dbvh, cam = soup
    rays = np.concatenate([scenes.primary_rays(cam, 160, 120)[0], scenes.random_rays(30011, seed=12), edge_rays()])
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=any_hit, threads=8)
    d_rays = up(rays)
    hint = nt.SchedHint()
    for gen in range(12):
        n = rays.shape[0] if gen != 6 else 777          # generation 6 rebinds the hint to a shorter batch
        if gen == 9:
            hint.reset()
        d_res = torch.full((rays.shape[0] * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
        dbvh.view.trace("fermi_speculative_while_while", n, any_hit, d_rays.data_ptr(), d_res.data_ptr(), hint=hint)
        torch.cuda.synchronize()
        assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE)[:n], ref[:n], "hint generation %d" % gen)
    # kernels without a block order accept the hint and ignore it
    d_res = torch.full((rays.shape[0] * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
    dbvh.view.trace("kepler_dynamic_fetch", rays.shape[0], any_hit, d_rays.data_ptr(), d_res.data_ptr(), hint=hint)
    torch.cuda.synchronize()
    assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), ref, "hint + persistent kernel")
    hint.close()


@pytest.mark.parametrize("n", [1, 255, 257, 16385, 70001])
def test_dispatch_order_prediction_changes_order_only(soup, n, monkeypatch):
    """Closest-hit launches are dispatched in predicted-cost order (sched_kernels.hip): the per-ray kernel maps workgroup i to block
    order[i], the persistent kernels hand their pool out in that order (every head walks its share of it).  Forced on for small, ragged
    launches here (block counts that are not multiples of 64 or of the number of pool heads, a single block)."""
    from gpu_util import assert_parity, gpu_trace
    dbvh, cam = soup
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_NODES", "1")
    nt.set_tunables()
    rays = np.concatenate([scenes.primary_rays(cam, 300, 240)[0], edge_rays()])[:n]
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=False, threads=8)
    kernels = ("fermi_speculative_while_while", "tesla_persistent_while_while", "kepler_dynamic_fetch")
    for kernel in kernels:
        for rep in range(3):  # the class counters alternate between two sets
            got, _ = gpu_trace(kernel, dbvh, rays, False)
            assert_parity(got, ref, "prediction %s n=%d rep=%d" % (kernel, n, rep))
    monkeypatch.setenv("NTR_TRACE_POOL_HEADS", "8")     # few heads: long per-head shares of the order
    monkeypatch.setenv("NTR_TRACE_CHUNK", "32")
    nt.set_tunables()
    for kernel in kernels[1:]:
        got, _ = gpu_trace(kernel, dbvh, rays, False)
        assert_parity(got, ref, "prediction %s n=%d, 8 heads, 32-ray chunks" % (kernel, n))
    monkeypatch.setenv("NTR_TRACE_PREDICT", "0")
    nt.set_tunables()
    for kernel in kernels:
        got, _ = gpu_trace(kernel, dbvh, rays, False)
        assert_parity(got, ref, "prediction off %s n=%d" % (kernel, n))


@pytest.mark.parametrize("octant", [0, 1])
def test_octant_slabs_change_no_record(soup, monkeypatch, octant):
    """The octant-specialised slab test (waves whose rays share their direction signs, NTR_BVH_ORDERED) is an instruction-count
    change only: coherent primary rays (the specialised path for most waves), the edge-case ray set and random rays (mixed
    octants inside a wave: the general path) give the oracle's records."""
    from gpu_util import assert_parity, gpu_trace
    dbvh, cam = soup
    assert dbvh.flags & nt.BVH_ORDERED
    monkeypatch.setenv("NTR_TRACE_OCTANT", str(octant))
    nt.set_tunables()
    rays = np.concatenate([scenes.primary_rays(cam, 320, 200)[0], edge_rays(), scenes.random_rays(20000, seed=11)])
    for any_hit in (False, True):
        ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=any_hit, threads=8)
        for kernel in ("fermi_speculative_while_while", "tesla_persistent_while_while"):
            got, _ = gpu_trace(kernel, dbvh, rays, any_hit)
            assert_parity(got, ref, "octant=%d %s anyHit=%d" % (octant, kernel, any_hit))


@pytest.mark.parametrize("waves", [1, 2, 4])
def test_workgroup_size_changes_no_record(soup, monkeypatch, waves):
    """The per-ray kernel runs in workgroups of 1, 2 or 4 waves (NTR_TRACE_CLOSEST_WAVES / NTR_TRACE_ANYHIT_WAVES); the dispatch
    order (prediction forced on) and the scheduling hint keep their 256-ray units.  Ragged ray counts, both ray kinds."""
    from gpu_util import assert_parity, gpu_trace
    dbvh, cam = soup
    monkeypatch.setenv("NTR_TRACE_CLOSEST_WAVES", str(waves))
    monkeypatch.setenv("NTR_TRACE_ANYHIT_WAVES", str(waves))
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_NODES", "1")
    nt.set_tunables()
    allrays = np.concatenate([scenes.primary_rays(cam, 300, 240)[0], edge_rays(), scenes.random_rays(5000, seed=4)])
    for n in (1, 63, 65, 257, 70001):
        rays = allrays[:n]
        for any_hit in (False, True):
            ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=any_hit, threads=8)
            got, _ = gpu_trace("fermi_speculative_while_while", dbvh, rays, any_hit)
            assert_parity(got, ref, "waves=%d n=%d anyHit=%d" % (waves, n, any_hit))


def test_validate_flags_an_inverted_box_as_unordered():
    """NTR_BVH_ORDERED is withheld when any child box has lo > hi on an axis (the octant-specialised slab test relies on lo <= hi);
    such a tree is still traced like the CPU tracer traces it."""
    from gpu_util import DeviceBvh, assert_parity, gpu_trace
    tri, pos, cam = scenes.random_soup(400, seed=5)
    bvh = nt.sah_build(tri, pos)
    nodes = bvh.nodes.copy()
    f = nodes.view(np.float32).reshape(-1, 16)
    lo, hi = f[0, 0], f[0, 1]
    f[0, 0], f[0, 1] = hi, lo                                  # root, child 0: lo.x <-> hi.x
    dbvh = DeviceBvh(nt.HostBvh(nodes, bvh.woop, bvh.tri_index))
    assert not (dbvh.flags & nt.BVH_ORDERED)
    good = DeviceBvh(bvh)
    assert good.flags & nt.BVH_ORDERED
    rays = scenes.primary_rays(cam, 160, 100)[0]
    ref, _ = oracle.trace(nodes, bvh.woop, bvh.tri_index, rays, any_hit=False, threads=8)
    got, _ = gpu_trace("fermi_speculative_while_while", dbvh, rays, False)
    assert_parity(got, ref, "inverted box")


@pytest.mark.parametrize("kernel", ["fermi_speculative_while_while", "kepler_dynamic_fetch"])
def test_trace_launch_can_be_captured_in_a_hip_graph_and_replayed(soup, monkeypatch, kernel):
    """An asynchronous ntr_trace_bvh (dispatch-order prediction forced on; the persistent kernel's ray-pool counters)
    captured into a HIP graph gives the oracle's records on every replay: all per-launch counters are cleared by
    kernels inside the captured work (memset nodes were observed not to re-execute on replay)."""
    import torch
    from gpu_util import assert_parity, up
    dbvh, cam = soup
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_NODES", "1")
    nt.set_tunables()
    rays = np.concatenate([scenes.primary_rays(cam, 200, 150)[0], scenes.random_rays(5000, seed=3)])
    n = rays.shape[0]
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=False, threads=8)
    d_rays = up(rays)
    d_res = torch.full((n * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):  # warm-up: scratch buffers and the top-of-tree table are allocated outside the capture
            dbvh.view.trace(kernel, n, False, d_rays.data_ptr(), d_res.data_ptr(), s.cuda_stream, False)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        dbvh.view.trace(kernel, n, False, d_rays.data_ptr(), d_res.data_ptr(),
                        torch.cuda.current_stream().cuda_stream, False)
    for rep in range(4):
        d_res.fill_(0xCD)
        g.replay()
        torch.cuda.synchronize()
        assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), ref, "%s graph replay %d" % (kernel, rep))


@pytest.mark.parametrize("tree", ["sah leaves of 1", "sah leaves of up to 6", "device lbvh"])
def test_unified_step_loop_changes_no_record(soup, monkeypatch, tree):
    """The unified-step loop (every live lane advances by one inner node OR one triangle per iteration; kepler_dynamic_fetch always, the
    per-ray kernel on trees flagged NTR_BVH_WIDE_LEAVES) interleaves the lanes differently from the while-while loop and nothing
    else: forced on and off for both kernels, on one-triangle leaves, multi-triangle leaves and a device-built LBVH; coherent, edge-case
    and random rays; ragged counts; closest hit and any hit; dynamic-fetch thresholds from 'never refill' to 'refill at once'."""
    import torch
    from gpu_util import DeviceBvh, assert_parity, gpu_trace, up
    tri, pos, cam = scenes.random_soup(6000, seed=23)
    if tree == "sah leaves of 1":
        dbvh = DeviceBvh(nt.sah_build(tri, pos, 1, 1))
        assert not (dbvh.flags & nt.BVH_WIDE_LEAVES)
    elif tree == "sah leaves of up to 6":
        dbvh = DeviceBvh(nt.sah_build(tri, pos, 3, 6))
        assert dbvh.flags & nt.BVH_WIDE_LEAVES
    else:
        n = tri.shape[0]
        capn, capw, capi = nt.lbvh_capacity(n)
        d_tri, d_pos = up(tri), up(pos)
        bufs = [torch.zeros(c, dtype=torch.uint8, device="cuda:0") for c in (capn, capw, capi)]
        mn, mx = oracle.scene_bbox(pos)
        res = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, bufs[0].data_ptr(), capn, bufs[1].data_ptr(), capw,
                            bufs[2].data_ptr(), capi)
        torch.cuda.synchronize()
        host = nt.HostBvh(bufs[0].cpu().numpy()[:res.nodesBytes].copy(), bufs[1].cpu().numpy()[:res.triWoopBytes].copy(),
                          bufs[2].cpu().numpy()[:res.triIndexBytes].view(np.int32).copy())
        dbvh = DeviceBvh(host)
        assert dbvh.flags & nt.BVH_WIDE_LEAVES
    allrays = np.concatenate([scenes.primary_rays(cam, 200, 160)[0], edge_rays(), scenes.random_rays(20000, seed=5)])
    for any_hit in (False, True):
        ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, allrays, any_hit=any_hit, threads=8)
        for env in ({"NTR_TRACE_PERRAY_UNIFIED": "1", "NTR_TRACE_FLAT_FETCH": "1"}, {"NTR_TRACE_PERRAY_UNIFIED": "1", "NTR_TRACE_FLAT_FETCH": "0"},
                    {"NTR_TRACE_PERRAY_UNIFIED": "0"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            nt.set_tunables()
            for n in (allrays.shape[0], 1, 65, 4097):
                got, _ = gpu_trace("fermi_speculative_while_while", dbvh, allrays[:n], any_hit)
                assert_parity(got, ref[:n], "%s per-ray %s n=%d anyHit=%d" % (tree, env, n, any_hit))
        for env in ({"NTR_TRACE_UNIFIED": "1", "NTR_TRACE_FETCH_THRESHOLD": "48", "NTR_TRACE_FLAT_FETCH": "1"},
                    {"NTR_TRACE_UNIFIED": "1", "NTR_TRACE_FETCH_THRESHOLD": "48", "NTR_TRACE_FLAT_FETCH": "0"}, {"NTR_TRACE_UNIFIED": "1", "NTR_TRACE_FETCH_THRESHOLD": "64"},
                    {"NTR_TRACE_UNIFIED": "1", "NTR_TRACE_FETCH_THRESHOLD": "1"}, {"NTR_TRACE_UNIFIED": "0", "NTR_TRACE_FETCH_THRESHOLD": "24"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            nt.set_tunables()
            for n in (allrays.shape[0], 63, 4097):
                got, _ = gpu_trace("kepler_dynamic_fetch", dbvh, allrays[:n], any_hit)
                assert_parity(got, ref[:n], "%s dynamic fetch %s n=%d anyHit=%d" % (tree, env, n, any_hit))


@pytest.mark.parametrize("tree", ["sah leaves of 1", "device lbvh"])
def test_wave_private_mini_pool_changes_no_record(monkeypatch, tree):
    """The wave-private mini-pool (a wave owns K x 64 rays of a 256-ray block and refills its finished lanes from them; closest-hit
    launches of the per-ray kernel, K decided on the device) changes which lane traces which ray and when, nothing else: K forced to
    1 ... 16 (pools that end inside a block, that span blocks of the dispatch order, whose last group is short) and left to the device, refill thresholds from 'only when the wave is empty' to 'at once', ragged counts around the
    64 / 128 / 256 boundaries, natural, predicted and measured (hinted) dispatch orders, coherent + edge-case + random rays."""
    import torch
    from gpu_util import DeviceBvh, assert_parity, gpu_trace, up
    tri, pos, cam = scenes.random_soup(6000, seed=29)
    if tree == "sah leaves of 1":
        dbvh = DeviceBvh(nt.sah_build(tri, pos, 1, 1))
    else:
        n = tri.shape[0]
        capn, capw, capi = nt.lbvh_capacity(n)
        d_tri, d_pos = up(tri), up(pos)
        bufs = [torch.zeros(c, dtype=torch.uint8, device="cuda:0") for c in (capn, capw, capi)]
        mn, mx = oracle.scene_bbox(pos)
        res = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, bufs[0].data_ptr(), capn, bufs[1].data_ptr(), capw,
                            bufs[2].data_ptr(), capi)
        torch.cuda.synchronize()
        dbvh = DeviceBvh(nt.HostBvh(bufs[0].cpu().numpy()[:res.nodesBytes].copy(), bufs[1].cpu().numpy()[:res.triWoopBytes].copy(),
                                    bufs[2].cpu().numpy()[:res.triIndexBytes].view(np.int32).copy()))
    allrays = np.concatenate([scenes.primary_rays(cam, 200, 160)[0], edge_rays(), scenes.random_rays(20000, seed=7)])
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, allrays, any_hit=False, threads=8)
    counts = (allrays.shape[0], 1, 63, 65, 129, 255, 257, 4097)
    for predict in ("0", "1"):
        monkeypatch.setenv("NTR_TRACE_PREDICT", predict)
        monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
        monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_NODES", "1")
        for env in ({"NTR_TRACE_MINIPOOL": "1"}, {"NTR_TRACE_MINIPOOL": "2"}, {"NTR_TRACE_MINIPOOL": "4"}, {"NTR_TRACE_MINIPOOL": "3"},
                    {"NTR_TRACE_MINIPOOL": "5"}, {"NTR_TRACE_MINIPOOL": "7", "NTR_TRACE_MINIPOOL_THRESHOLD": "33"}, {"NTR_TRACE_MINIPOOL": "16"},
                    {"NTR_TRACE_MINIPOOL": "4", "NTR_TRACE_MINIPOOL_THRESHOLD": "1"}, {"NTR_TRACE_MINIPOOL": "2", "NTR_TRACE_MINIPOOL_THRESHOLD": "64"},
                    {"NTR_TRACE_MINIPOOL": "-1", "NTR_TRACE_MINIPOOL_WIDE": "4"}, {"NTR_TRACE_MINIPOOL": "-1", "NTR_TRACE_MINIPOOL_WIDE": "2"},
                    # the pools with the two-descriptor fetch (what a BVH whose buffers lie more than 4 GiB apart gets)
                    {"NTR_TRACE_MINIPOOL": "4", "NTR_TRACE_FLAT_FETCH": "0"}, {"NTR_TRACE_MINIPOOL": "-1", "NTR_TRACE_FLAT_FETCH": "0"},
                    {"NTR_TRACE_MINIPOOL": "0"}):
            for k in ("NTR_TRACE_MINIPOOL", "NTR_TRACE_MINIPOOL_THRESHOLD", "NTR_TRACE_MINIPOOL_WIDE", "NTR_TRACE_FLAT_FETCH"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            nt.set_tunables()
            for n in counts:
                got, _ = gpu_trace("fermi_speculative_while_while", dbvh, allrays[:n], False)
                assert_parity(got, ref[:n], "%s mini-pool %s predict=%s n=%d" % (tree, env, predict, n))
    # measured order: the same buffers traced repeatedly keep a hint (and their K in it) -- random rays first, so that the device picks K > 1
    monkeypatch.setenv("NTR_TRACE_AUTO_HINT_MIN_RAYS", "1")
    monkeypatch.delenv("NTR_TRACE_MINIPOOL", raising=False)
    monkeypatch.delenv("NTR_TRACE_MINIPOOL_THRESHOLD", raising=False)
    nt.set_tunables()
    # (first with the batch's first launch predicted, then with batches too small to be predicted: their K comes from the coherence probe
    # of the hint's refresh launches)
    rnd = scenes.random_rays(30000, seed=11)
    rnd_ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rnd, any_hit=False, threads=8)
    for min_rays in ("1", "100000000"):
        monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", min_rays)
        nt.set_tunables()
        for rays, want in ((rnd, rnd_ref), (allrays, ref)):
            d_rays = up(rays)
            d_res = torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device="cuda:0")
            for rep in range(5):
                d_res.zero_()
                dbvh.view.trace("fermi_speculative_while_while", rays.shape[0], False, d_rays.data_ptr(), d_res.data_ptr())
                torch.cuda.synchronize()
                assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), want, "%s mini-pool, hinted launch %d, predict from %s rays" % (tree, rep, min_rays))
    # the rays of a batch change under its hint (same buffer, same count): camera rays become scattered ones and back; the hint's refresh
    # launches re-estimate the coherence words, and whatever K a launch runs with, the records are the oracle's
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_SCHED_REFRESH_EVERY", "2")
    nt.set_tunables()
    m = rnd.shape[0]
    cam_rays, cam_ref = allrays[:m], ref[:m]
    d_rays = up(cam_rays)
    d_res = torch.zeros(m * 16, dtype=torch.uint8, device="cuda:0")
    for phase, (rays, want) in enumerate(((cam_rays, cam_ref), (rnd, rnd_ref), (cam_rays, cam_ref))):
        d_rays.copy_(up(rays))
        for rep in range(6):
            d_res.zero_()
            dbvh.view.trace("fermi_speculative_while_while", m, False, d_rays.data_ptr(), d_res.data_ptr())
            torch.cuda.synchronize()
            assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), want, "%s mini-pool, batch contents changed (phase %d), launch %d" % (tree, phase, rep))


@pytest.mark.parametrize("tree", ["sah leaves of 1", "device lbvh"])
def test_uniform_prologue_changes_no_record(monkeypatch, tree):
    """Wave-uniform prologue of the per-ray kernels (round 5): while every live lane of a fresh wave holds the same inner node the node
    is fetched once through the scalar cache.  Same arithmetic, same visiting order: records equal the oracle's with the prologue on
    and off -- camera tiles (uniform for many levels), bundles of rays from ONE origin (uniform until the directions part), waves that
    are half degenerate (the first live lane is not lane 0), edge-case rays (non-finite values take the generic slab path through the
    prologue too), any hit and closest hit, and a node buffer that is only 16-byte aligned (the prologue then stands down)."""
    import torch
    from gpu_util import DeviceBvh, assert_parity, gpu_trace, up
    tri, pos, cam = scenes.random_soup(6000, seed=41)
    if tree == "sah leaves of 1":
        host = nt.sah_build(tri, pos, 1, 1)
    else:
        n = tri.shape[0]
        capn, capw, capi = nt.lbvh_capacity(n)
        d_tri, d_pos = up(tri), up(pos)
        bufs = [torch.zeros(c, dtype=torch.uint8, device="cuda:0") for c in (capn, capw, capi)]
        mn, mx = oracle.scene_bbox(pos)
        res = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, bufs[0].data_ptr(), capn, bufs[1].data_ptr(), capw,
                            bufs[2].data_ptr(), capi)
        torch.cuda.synchronize()
        host = nt.HostBvh(bufs[0].cpu().numpy()[:res.nodesBytes].copy(), bufs[1].cpu().numpy()[:res.triWoopBytes].copy(),
                          bufs[2].cpu().numpy()[:res.triIndexBytes].view(np.int32).copy())
    dbvh = DeviceBvh(host)
    prim = scenes.primary_rays(cam, 256, 192)[0]
    bundle = scenes.random_rays(64 * 300, seed=3)
    for k in ("ox", "oy", "oz"):                       # 300 bundles of 64 rays from one origin each
        bundle[k] = np.repeat(bundle[k][::64], 64)
    allrays = np.concatenate([prim, bundle, edge_rays(), scenes.random_rays(5000, seed=4)])
    allrays["tmax"][0:6400:2] = -1.0                   # every other ray of the first hundred waves degenerate
    allrays["tmax"][64 * 200:64 * 200 + 37] = -1.0     # a wave whose first 37 lanes are dead
    # the same BVH behind a node pointer that is 16-byte but not 64-byte aligned
    raw = torch.zeros(host.nodes.nbytes + 64, dtype=torch.uint8, device="cuda:0")
    raw[16:16 + host.nodes.nbytes] = up(host.nodes)
    off_view = nt.BvhView(raw.data_ptr() + 16, host.nodes.nbytes, dbvh.woop.data_ptr(), host.woop.nbytes, dbvh.idx.data_ptr())
    off_view.validate()
    try:
        for any_hit in (False, True):
            ref, _ = oracle.trace(host.nodes, host.woop, host.tri_index, allrays, any_hit=any_hit, threads=8)
            for knob in ("1", "0"):
                monkeypatch.setenv("NTR_TRACE_UNIFORM_PROLOGUE", knob)
                nt.set_tunables()
                for n in (allrays.shape[0], 64, 65, 4097):
                    got, _ = gpu_trace("fermi_speculative_while_while", dbvh, allrays[:n], any_hit)
                    assert_parity(got, ref[:n], "%s prologue=%s any_hit=%s n=%d" % (tree, knob, any_hit, n))
                d_rays = up(allrays)
                d_res = torch.full((allrays.shape[0] * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
                off_view.trace("fermi_speculative_while_while", allrays.shape[0], any_hit, d_rays.data_ptr(), d_res.data_ptr())
                torch.cuda.synchronize()
                assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), ref, "%s prologue=%s any_hit=%s, node buffer at +16 bytes" % (tree, knob, any_hit))
    finally:
        monkeypatch.delenv("NTR_TRACE_UNIFORM_PROLOGUE", raising=False)
        nt.set_tunables()


@pytest.mark.parametrize("tree", ["atrium, sah leaves of 1", "soup, device lbvh"])
def test_ray_splitting_changes_no_record(monkeypatch, tree):
    """Ray splitting in the drain phase of the persistent kernels (trace_split.h, round 5): once the pool is dry, lanes without a ray
    traverse the bottom stack entries of the wave's live rays.  The reference's record depends on the traversal's history (a triangle a
    few ulp closer than the record can sit in a node the lone ray skips), so a helper's hit counts only under the bound the lone ray
    would bring along -- merging parts by t differs from the reference on ~5 of every 10^6 box rays of the atrium tree, which is why
    that tree is here at 2^20 rays.  Records must be the oracle's with the lanes looked at after every step, every 3 / 8 / 64 steps and
    never; ragged counts (waves that start with most lanes idle), edge-case rays (non-finite values: the helpers copy the generic slab
    path's ray too), closest hit and any hit (the record is then the FIRST hit in visiting order: a helper's hit counts only when
    everything before its entry ended without one), unified-step and while-while persistent kernels."""
    from gpu_util import DeviceBvh, assert_parity, gpu_trace
    if tree.startswith("atrium"):
        tri, pos, cam = scenes.atrium()
        dbvh = DeviceBvh(nt.sah_build(tri, pos, 1, 1))
        rays = np.concatenate([scenes.box_rays(pos, 1 << 20, seed=21), edge_rays()])
    else:
        import torch
        from gpu_util import up
        tri, pos, cam = scenes.random_soup(40000, seed=53)
        n = tri.shape[0]
        capn, capw, capi = nt.lbvh_capacity(n)
        d_tri, d_pos = up(tri), up(pos)
        bufs = [torch.zeros(c, dtype=torch.uint8, device="cuda:0") for c in (capn, capw, capi)]
        mn, mx = oracle.scene_bbox(pos)
        res = nt.lbvh_build(n, d_tri.data_ptr(), pos.shape[0], d_pos.data_ptr(), mn, mx, 8, 0.001, bufs[0].data_ptr(), capn, bufs[1].data_ptr(), capw,
                            bufs[2].data_ptr(), capi)
        torch.cuda.synchronize()
        dbvh = DeviceBvh(nt.HostBvh(bufs[0].cpu().numpy()[:res.nodesBytes].copy(), bufs[1].cpu().numpy()[:res.triWoopBytes].copy(),
                                    bufs[2].cpu().numpy()[:res.triIndexBytes].view(np.int32).copy()))
        rays = np.concatenate([scenes.box_rays(pos, 300000, seed=22), edge_rays(), scenes.primary_rays(cam, 320, 200)[0], scenes.random_rays(50000, seed=9)])
    try:
        for any_hit in (False, True):
            ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=any_hit, threads=8)
            for slice_ in (("1", "3", "8", "64", "0") if not any_hit else ("1", "8", "0")):
                monkeypatch.setenv("NTR_TRACE_SPLIT_SLICE", slice_)
                nt.set_tunables()
                for kernel in ("kepler_dynamic_fetch", "tesla_persistent_while_while"):
                    for n in ((rays.shape[0], 65, 4097) if kernel == "kepler_dynamic_fetch" else (rays.shape[0],)):
                        got, _ = gpu_trace(kernel, dbvh, rays[:n], any_hit)
                        assert_parity(got, ref[:n], "%s %s split slice %s any_hit=%s n=%d" % (tree, kernel, slice_, any_hit, n))
    finally:
        monkeypatch.delenv("NTR_TRACE_SPLIT_SLICE", raising=False)
        nt.set_tunables()


def test_buffers_more_than_4_gib_apart_take_the_descriptor_fetch():
    """The flat fetch of the unified-step loop addresses node and triangle buffers from ONE scalar base with 32-bit lane offsets (round 5),
    so the library uses it only when both buffers lie inside one 4 GiB window; buffers further apart get the two-descriptor fetch.
    Here 5 GiB of other allocations sit between the two: records equal the oracle's for every kernel name, closest hit and any hit."""
    import torch
    from gpu_util import assert_parity, up
    tri, pos, cam = scenes.random_soup(5000, seed=47)
    host = nt.sah_build(tri, pos, 1, 4)
    rays = np.concatenate([scenes.primary_rays(cam, 160, 120)[0], edge_rays(), scenes.random_rays(8000, seed=6)])
    d_nodes = up(host.nodes)
    # the allocator decides where things go: interleave spacers and copies of the triangle buffer until one copy lies > 4 GiB from the nodes
    spacer, d_woop = [], None
    for _ in range(6):
        spacer.append(torch.empty(3 << 30, dtype=torch.uint8, device="cuda:0"))
        cand = up(host.woop)
        if abs(cand.data_ptr() - d_nodes.data_ptr()) > (4 << 30) + host.woop.nbytes + host.nodes.nbytes:
            d_woop = cand
            break
        spacer.append(cand)
    if d_woop is None:
        pytest.skip("the allocator kept every copy within 4 GiB of the node buffer")
    d_idx = up(host.tri_index)
    view = nt.BvhView(d_nodes.data_ptr(), host.nodes.nbytes, d_woop.data_ptr(), host.woop.nbytes, d_idx.data_ptr())
    view.validate()
    d_rays = up(rays)
    for any_hit in (False, True):
        ref, _ = oracle.trace(host.nodes, host.woop, host.tri_index, rays, any_hit=any_hit, threads=8)
        for kernel in KERNELS:
            d_res = torch.full((rays.shape[0] * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
            view.trace(kernel, rays.shape[0], any_hit, d_rays.data_ptr(), d_res.data_ptr())
            torch.cuda.synchronize()
            assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), ref, "%s any_hit=%s, buffers %d MiB apart" % (kernel, any_hit, abs(d_woop.data_ptr() - d_nodes.data_ptr()) >> 20))
    del spacer


def test_stream_release_returns_a_streams_scheduling_state(monkeypatch):
    """ntr_stream_release: a stream's automatic hints and prediction scratch go back before the host destroys the stream; tracing on the
    stream afterwards simply starts over (first sighting, second sighting, hint), and other streams' entries are untouched."""
    import torch
    from gpu_util import DeviceBvh, assert_parity, up
    tri, pos, cam = scenes.random_soup(4000, seed=43)
    dbvh = DeviceBvh(nt.sah_build(tri, pos, 1, 1))
    rays = scenes.primary_rays(cam, 320, 240)[0]
    ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=False, threads=8)
    monkeypatch.setenv("NTR_TRACE_AUTO_HINT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_NODES", "1")
    nt.set_tunables()
    try:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        d_rays = up(rays)
        outs = [torch.zeros(rays.shape[0] * 16, dtype=torch.uint8, device="cuda:0") for _ in streams]
        torch.cuda.synchronize()
        for cycle in range(3):
            for rep in range(4):
                for s_, o in zip(streams, outs):
                    o.zero_()
                    torch.cuda.synchronize()
                    dbvh.view.trace("fermi_speculative_while_while", rays.shape[0], False, d_rays.data_ptr(), o.data_ptr(), s_.cuda_stream, False)
                torch.cuda.synchronize()
                for o in outs:
                    assert_parity(o.cpu().numpy().view(nt.RESULT_DTYPE), ref, "cycle %d launch %d" % (cycle, rep))
            nt.stream_release(streams[0].cuda_stream)       # stream 0 starts over in the next cycle, stream 1 keeps its hint
        nt.stream_release(streams[1].cuda_stream)
        nt.stream_release(streams[1].cuda_stream)           # releasing a stream that owns nothing is fine
    finally:
        for k in ("NTR_TRACE_AUTO_HINT_MIN_RAYS", "NTR_TRACE_PREDICT_MIN_RAYS", "NTR_TRACE_PREDICT_MIN_NODES"):
            monkeypatch.delenv(k, raising=False)
        nt.set_tunables()


def test_batch_coherence_estimate_separates_camera_rays_from_scattered_ones(soup, monkeypatch):
    """ntr_predict_batch_coherence (the words the dispatch-order prediction derives on the device): rays from one camera start together
    and point alike -- no incoherent block, K = 1; rays that start anywhere in the scene's box are incoherent in nearly every block --
    K = the wide pool (2 on a small tree, 4 when asked for); LONG rays that start together and point anywhere (a diffuse batch) are counted
    as direction-incoherent: the pool K stays 1 and the word's NTR_BATCH_DIVERGENT bit (16) is set -- SHORT ones (an AO batch) are not; an
    empty batch is coherent."""
    import torch
    from gpu_util import up
    dbvh, cam = soup
    out = torch.full((3,), 77, dtype=torch.int32, device="cuda:0")

    def query(rays):
        d = up(rays) if rays.shape[0] else None
        nt.predict_batch_coherence(rays.shape[0], d.data_ptr() if d is not None else 0, dbvh.nodes.data_ptr(), dbvh.host.nodes.nbytes, out.data_ptr())
        torch.cuda.synchronize()
        return out.cpu().tolist()

    prim = scenes.primary_rays(cam, 320, 200)[0]
    rnd = scenes.random_rays(64000, seed=3)
    fan = rnd.copy()
    for k in ("ox", "oy", "oz"):
        fan[k] = prim[k][0]
    blocks = rnd.shape[0] // 256
    assert query(prim) == [0, 0, 1]
    o, d, k = query(rnd)
    assert o >= 0.8 * blocks and k == (2 | 0x10000), (o, d, k)   # (a small tree: the wide pool is 2; scattered origins are "divergent" too)
    o, d, k = query(fan)
    assert o == 0 and d >= 0.6 * blocks and k == (1 | 0x10000), (o, d, k)
    short = fan.copy()
    short["tmin"] = 0.0
    short["tmax"] = 1e-3 * float(np.abs(prim["ox"][0]) + np.abs(prim["oy"][0]) + np.abs(prim["oz"][0]) + 1.0)
    assert query(short) == [0, 0, 1]
    monkeypatch.setenv("NTR_TRACE_MINIPOOL_WIDE", "4")
    nt.set_tunables()
    assert query(rnd)[2] == (4 | 0x10000) and query(fan)[2] == (1 | 0x10000) and query(prim)[2] == 1
    assert query(prim[:0]) == [0, 0, 1]


def test_captured_launches_own_their_scratch_and_release_returns_it(soup, monkeypatch):
    """HIP-graph resources: a captured launch gets prediction scratch / pool counters of its own -- two launches captured
    back to back on one stream, with no live launch in between, and live launches of other ray counts on the same stream while the
    graphs exist, all give the oracle's records on replay --, the stores are finite (the capture-time call says so instead of
    corrupting anything), and ntr_trace_graph_release_all() returns everything: a host that re-captures every frame keeps going."""
    import torch
    from gpu_util import assert_parity, up
    dbvh, cam = soup
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_NODES", "1")
    nt.set_tunables()
    nt.trace_graph_release_all()
    batches = []
    for (w, h, seed) in ((200, 150, 3), (160, 120, 4)):
        rays = np.concatenate([scenes.primary_rays(cam, w, h)[0], scenes.random_rays(3000, seed=seed)])
        ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=False, threads=8)
        batches.append((rays.shape[0], up(rays), torch.full((rays.shape[0] * 16,), 0xCD, dtype=torch.uint8, device="cuda:0"), ref))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):   # one live launch of the larger size provisions the spares
        dbvh.view.trace("fermi_speculative_while_while", batches[0][0], False, batches[0][1].data_ptr(), batches[0][2].data_ptr(), s.cuda_stream, False)
    torch.cuda.synchronize()
    graphs = []
    for (n, d_r, d_o, ref) in batches:   # two captures back to back
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            dbvh.view.trace("fermi_speculative_while_while", n, False, d_r.data_ptr(), d_o.data_ptr(), torch.cuda.current_stream().cuda_stream, False)
        graphs.append(g)
    with torch.cuda.stream(s):   # live launches on the capture stream must not disturb the graphs' scratch
        for k in (1, 257, 5000):
            tmp = torch.zeros(k * 16, dtype=torch.uint8, device="cuda:0")
            dbvh.view.trace("fermi_speculative_while_while", k, False, batches[0][1].data_ptr(), tmp.data_ptr(), s.cuda_stream, False)
    for rep in range(2):
        for g, (n, d_r, d_o, ref) in zip(graphs, batches):
            d_o.fill_(0xCD)
            g.replay()
        torch.cuda.synchronize()
        for (n, d_r, d_o, ref) in batches:
            assert_parity(d_o.cpu().numpy().view(nt.RESULT_DTYPE), ref, "captured launch, replay %d" % rep)
    # the persistent kernels' pinned pool counters: 192 per device, then a clear error; release_all makes room again
    # (prediction off: a captured persistent launch with a predicted pool order would run out of its four scratch spares first)
    n, d_r, d_o, ref = batches[1]
    del graphs
    nt.trace_graph_release_all()
    monkeypatch.setenv("NTR_TRACE_PREDICT", "0")
    nt.set_tunables()
    with torch.cuda.stream(s):
        dbvh.view.trace("kepler_dynamic_fetch", n, False, d_r.data_ptr(), d_o.data_ptr(), s.cuda_stream, False)
    torch.cuda.synchronize()
    held, failed = [], False
    for i in range(200):
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s):
                dbvh.view.trace("kepler_dynamic_fetch", n, False, d_r.data_ptr(), d_o.data_ptr(), torch.cuda.current_stream().cuda_stream, False)
            held.append(g)
        except nt.NtrError as e:
            failed = True
            assert "captured" in str(e)
            break
    assert failed and len(held) == 192
    torch.cuda.synchronize()
    del held
    nt.trace_graph_release_all()
    for frame in range(3):   # re-capture per frame with a release in between
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            dbvh.view.trace("kepler_dynamic_fetch", n, False, d_r.data_ptr(), d_o.data_ptr(), torch.cuda.current_stream().cuda_stream, False)
        d_o.fill_(0xCD)
        g.replay()
        torch.cuda.synchronize()
        assert_parity(d_o.cpu().numpy().view(nt.RESULT_DTYPE), ref, "re-captured frame %d" % frame)
        del g
        nt.trace_graph_release_all()


@pytest.mark.parametrize("any_hit", [False, True])
def test_automatic_scheduling_feedback_changes_no_record(soup, monkeypatch, any_hit):
    """Launches of the per-ray kernel on the same batch (stream, ray buffer, count, kind, BVH) dispatch their blocks in the order the
    previous launch measured (the library's own NtrSchedHint, NTR_TRACE_AUTO_HINT): every generation of the order -- natural or
    predicted, first measured, re-measured, new rays written into the same buffer, the feature switched off -- gives the oracle's
    records."""
    import torch
    from gpu_util import assert_parity, up
    dbvh, cam = soup
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_NODES", "1")
    monkeypatch.setenv("NTR_TRACE_AUTO_HINT_MIN_RAYS", "1000")
    nt.set_tunables()
    a = np.concatenate([scenes.primary_rays(cam, 400, 300)[0], edge_rays(), scenes.random_rays(30001, seed=12)])
    b = np.concatenate([scenes.random_rays(30001, seed=13), scenes.primary_rays(cam, 400, 300)[0], edge_rays()])   # same count, other rays
    n = a.shape[0]
    ref_a, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, a, any_hit=any_hit, threads=8)
    ref_b, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, b, any_hit=any_hit, threads=8)
    d_rays = up(a)
    d_res = torch.full((n * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.Stream()
    for stream in (0, s.cuda_stream):
        for gen in range(10):
            if gen == 6:   # new rays at the old address: the learned order is stale, never wrong
                d_rays.copy_(up(b))
                torch.cuda.synchronize()
            d_res.fill_(0xCD)
            torch.cuda.synchronize()
            dbvh.view.trace("fermi_speculative_while_while", n, any_hit, d_rays.data_ptr(), d_res.data_ptr(), stream, gen % 2 == 0)
            torch.cuda.synchronize()
            assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), ref_b if gen >= 6 else ref_a, "feedback generation %d stream %s" % (gen, stream))
        d_rays.copy_(up(a))
        torch.cuda.synchronize()
    monkeypatch.setenv("NTR_TRACE_AUTO_HINT", "0")
    nt.set_tunables()
    d_res.fill_(0xCD)
    dbvh.view.trace("fermi_speculative_while_while", n, any_hit, d_rays.data_ptr(), d_res.data_ptr())
    torch.cuda.synchronize()
    assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), ref_a, "feedback off")


def test_scheduling_feedback_with_two_bvhs_alternating_through_the_same_buffers(monkeypatch):
    """Two DIFFERENT trees and ray sets traced alternately through the SAME node / triangle / ray / result buffers (a host that rebuilds
    its BVH in place and regenerates its rays every frame): whatever the library remembers about a batch -- its top-of-tree table, the
    dispatch order it measured, the pool depth it derived -- belongs to contents that are gone one launch later.  Stale knowledge may cost
    time, never a record.  Also: more distinct batches than the library keeps entries for (it recycles them without waiting)."""
    import torch
    from gpu_util import assert_parity, up
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_NODES", "1")
    monkeypatch.setenv("NTR_TRACE_AUTO_HINT_MIN_RAYS", "1000")
    monkeypatch.setenv("NTR_SCHED_REFRESH_EVERY", "3")
    nt.set_tunables()
    trees, rays, refs = [], [], {}
    for seed, ntri in ((61, 9000), (62, 9000)):   # same triangle count: SAH trees with one-triangle leaves have equal buffer sizes
        tri, pos, cam = scenes.random_soup(ntri, seed=seed)
        bvh = nt.sah_build(tri, pos, 1, 1)
        trees.append(bvh)
        rays.append(np.concatenate([scenes.primary_rays(cam, 300, 200)[0], scenes.random_rays(40000, seed=seed)]))
    nbytes = [max(getattr(t, f).nbytes for t in trees) for f in ("nodes", "woop", "tri_index")]
    d_nodes = torch.zeros(nbytes[0], dtype=torch.uint8, device="cuda:0")
    d_woop = torch.zeros(nbytes[1], dtype=torch.uint8, device="cuda:0")
    d_idx = torch.zeros(nbytes[2], dtype=torch.uint8, device="cuda:0")
    n = rays[0].shape[0]
    d_rays = torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0")
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    for any_hit in (False, True):
        for k in (0, 1):
            refs[(k, any_hit)] = oracle.trace(trees[k].nodes, trees[k].woop, trees[k].tri_index, rays[k], any_hit=any_hit, threads=8)[0]
    for gen in range(14):
        k = gen % 2 if gen < 10 else (gen // 2) % 2     # strictly alternating, then in pairs
        t = trees[k]
        d_nodes[: t.nodes.nbytes].copy_(up(t.nodes))
        d_woop[: t.woop.nbytes].copy_(up(t.woop))
        d_idx[: t.tri_index.nbytes].copy_(up(t.tri_index))
        d_rays.copy_(up(rays[k]))
        torch.cuda.synchronize()
        view = nt.BvhView(d_nodes.data_ptr(), t.nodes.nbytes, d_woop.data_ptr(), t.woop.nbytes, d_idx.data_ptr())
        if gen % 3 == 0:
            view.validate()      # a host may or may not re-validate after a rebuild: the top-of-tree table is stale otherwise
        for any_hit in (False, True):
            d_res.fill_(0xCD)
            view.trace("fermi_speculative_while_while", n, any_hit, d_rays.data_ptr(), d_res.data_ptr(), 0, gen % 4 == 0)
            torch.cuda.synchronize()
            assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE), refs[(k, any_hit)], "tree %d generation %d anyHit=%d" % (k, gen, any_hit))
    # more distinct batches than the 96 feedback entries: ray counts that change every launch
    t = trees[0]
    d_nodes[: t.nodes.nbytes].copy_(up(t.nodes)); d_woop[: t.woop.nbytes].copy_(up(t.woop)); d_idx[: t.tri_index.nbytes].copy_(up(t.tri_index))
    d_rays.copy_(up(rays[0]))
    torch.cuda.synchronize()
    view = nt.BvhView(d_nodes.data_ptr(), t.nodes.nbytes, d_woop.data_ptr(), t.woop.nbytes, d_idx.data_ptr())
    view.validate()
    for i in range(130):
        m = n - 64 * i
        d_res.fill_(0xCD)
        for _ in range(2 if i % 5 == 0 else 1):
            view.trace("fermi_speculative_while_while", m, False, d_rays.data_ptr(), d_res.data_ptr(), 0, False)
        if i % 13 == 0:
            torch.cuda.synchronize()
            assert_parity(d_res.cpu().numpy().view(nt.RESULT_DTYPE)[:m], refs[(0, False)][:m], "changing count %d" % m)
    torch.cuda.synchronize()
    assert nt.trace_status() == 0


@pytest.mark.parametrize("route", ["1", "0"])
@pytest.mark.parametrize("kernel", KERNELS)
def test_routing_by_coherence_keeps_every_record(soup, kernel, route, monkeypatch):
    """Round 6: closest-hit launches the device can classify are launched as BOTH bodies and the batch word decides which one works; any-hit
    launches run the per-ray body under every name.  Whatever the name, whatever the batch (camera rays: coherent; box rays: scattered
    origins; a fan of long rays from one point: divergent), with routing on and with the named body forced -- the oracle's records.  The
    estimate's size thresholds are lowered so that a test-sized batch is routed; launches are repeated so that the batch's automatic hint
    forms and its kept batch word, not only a fresh prediction, steers the launch."""
    from gpu_util import assert_parity, gpu_trace
    dbvh, cam = soup
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_PREDICT_MIN_NODES", "1")
    monkeypatch.setenv("NTR_TRACE_AUTO_HINT_MIN_RAYS", "1")
    monkeypatch.setenv("NTR_TRACE_ROUTE", route)
    nt.set_tunables()
    try:
        plan = nt.trace_plan(kernel, 100000, False, dbvh.host.nodes.nbytes, dbvh.host.woop.nbytes)
        assert plan.coherentRoute == (1 if route == "1" else 0)
        prim = scenes.primary_rays(cam, 320, 240)[0]
        rnd = scenes.random_rays(70000, seed=5)
        fan = rnd.copy()
        for k in ("ox", "oy", "oz"):
            fan[k] = prim[k][0]
        for name, rays in (("camera", prim), ("box", rnd), ("fan", fan), ("mixed", np.concatenate([prim, rnd, fan]))):
            for any_hit in (False, True):
                ref, _ = oracle.trace(dbvh.host.nodes, dbvh.host.woop, dbvh.host.tri_index, rays, any_hit=any_hit, threads=8)
                for rep in range(4):
                    got, _ = gpu_trace(kernel, dbvh, rays, any_hit)
                    assert_parity(got, ref, "%s %s any_hit=%d route=%s launch %d" % (kernel, name, any_hit, route, rep))
    finally:
        for k in ("NTR_TRACE_PREDICT_MIN_RAYS", "NTR_TRACE_PREDICT_MIN_NODES", "NTR_TRACE_AUTO_HINT_MIN_RAYS", "NTR_TRACE_ROUTE"):
            monkeypatch.delenv(k, raising=False)
        nt.set_tunables()
