"""Dispatch hints that start from a prediction (ntr_bvh_leaf_depths, ntr_secondary_block_costs, ntr_sched_hint_predict): the leaf depths
equal a host walk of the same buffers (SAH and device-LBVH trees), the block costs equal their numpy restatement, and a launch with a
predicted hint -- any prediction, an adversarial one included -- returns the records of the plain launch."""
import numpy as np
import pytest
import torch

import ntrace_amd as nt
from gpu_util import DeviceBvh, assert_parity, gpu_trace, up
from ntrace_amd import scenes
from oracle import oracle

pytestmark = pytest.mark.gpu
K = "fermi_speculative_while_while"


def host_leaf_depths(nodes_u8, woop_u8, tri_index, num_tris):
    nodes = nodes_u8.view(np.int32).reshape(-1, 16)
    woop = woop_u8.view(np.uint32).reshape(-1, 4)
    depth = np.zeros(num_tris, np.int32)
    todo = [(0, 0)]
    deepest = 0
    while todo:
        ofs, d = todo.pop()
        rec = nodes[ofs // 64]
        for ch in (int(rec[12]), int(rec[13])):
            if ch >= 0:
                todo.append((ch, d + 1))
            else:
                a = ~ch
                deepest = max(deepest, d + 1)
                while woop[a][0] != 0x80000000:
                    depth[tri_index[a]] = d + 1
                    a += 3
    return depth, deepest


def device_leaf_depths(dbvh, num_tris):
    d_depth = torch.full((num_tris,), -7, dtype=torch.int32, device="cuda:0")
    levels = nt.bvh_leaf_depths(dbvh.view.d_nodes, dbvh.view.nodes_bytes, dbvh.view.d_woop, dbvh.view.woop_bytes, dbvh.view.d_tri_index, num_tris,
                                d_depth.data_ptr())
    return d_depth, levels


@pytest.mark.parametrize("leaf", [1, 4])
def test_leaf_depths_equal_a_host_walk_sah(leaf):
    tri, pos = scenes.random_soup(6000, seed=5)[:2]
    bvh = nt.sah_build(tri, pos, 1, leaf)
    want, deepest = host_leaf_depths(bvh.nodes, bvh.woop, bvh.tri_index.view(np.int32), tri.shape[0])
    dbvh = DeviceBvh(bvh)
    d_depth, levels = device_leaf_depths(dbvh, tri.shape[0])
    assert np.array_equal(d_depth.cpu().numpy(), want)
    assert want.min() >= 1                       # every triangle sits in a leaf
    assert deepest <= levels <= deepest + 4      # levels walked: the tree's depth, checked every fourth level


def test_leaf_depths_equal_a_host_walk_device_lbvh():
    tri, pos = scenes.random_soup(20000, seed=9)[:2]
    ref = oracle.lbvh_build(tri, pos, 8, 0.001)
    bvh = nt.HostBvh(ref["nodes"], ref["woop"], ref["tri_index"])
    want, deepest = host_leaf_depths(np.ascontiguousarray(ref["nodes"]).view(np.uint8).reshape(-1), np.ascontiguousarray(ref["woop"]).view(np.uint8).reshape(-1),
                                     np.ascontiguousarray(ref["tri_index"]).view(np.int32).reshape(-1), tri.shape[0])
    dbvh = DeviceBvh(bvh)
    d_depth, levels = device_leaf_depths(dbvh, tri.shape[0])
    assert np.array_equal(d_depth.cpu().numpy(), want)
    assert deepest <= levels <= deepest + 4


def test_block_costs_and_predicted_hints_leave_the_records_alone():
    tri, pos, cam = scenes.atrium()
    bvh = nt.sah_build(tri, pos, 1, 1)
    dbvh = DeviceBvh(bvh)
    rays, _ = scenes.primary_rays(cam, 320, 200)
    res, _ = gpu_trace(K, dbvh, rays)
    want_depth, _ = host_leaf_depths(bvh.nodes, bvh.woop, bvh.tri_index.view(np.int32), tri.shape[0])
    d_depth, _ = device_leaf_depths(dbvh, tri.shape[0])
    first, count, ns = 1000, 30011, 8           # a ragged batch: the last block is partial, samples straddle no block (256 % 8 == 0)
    nblocks = (count * ns + 255) // 256
    d_res = up(res)
    d_cost = torch.full((nblocks,), 12345, dtype=torch.int32, device="cuda:0")
    nt.secondary_block_costs(d_res.data_ptr(), first, count, ns, d_depth.data_ptr(), tri.shape[0], d_cost.data_ptr())
    ids = res["id"][first:first + count]
    dep = np.where(ids >= 0, want_depth[np.maximum(ids, 0)], 0)
    want_cost = np.zeros(nblocks, np.int64)
    np.maximum.at(want_cost, (np.arange(count) * ns) // 256, dep)
    assert np.array_equal(d_cost.cpu().numpy().astype(np.int64), want_cost)
    # samples that straddle blocks (256 % 3 != 0): an input ray counts for both
    d_cost3 = torch.zeros(((count * 3 + 255) // 256,), dtype=torch.int32, device="cuda:0")
    nt.secondary_block_costs(d_res.data_ptr(), first, count, 3, d_depth.data_ptr(), tri.shape[0], d_cost3.data_ptr())
    want3 = np.zeros(d_cost3.numel(), np.int64)
    for k in range(3):
        np.maximum.at(want3, (np.arange(count) * 3 + k) // 256, dep)
    assert np.array_equal(d_cost3.cpu().numpy().astype(np.int64), want3)

    # AO rays of that batch; plain launch = the reference records
    d_rays = up(rays)
    d_nrm = up(scenes.tri_normals(tri, pos))
    n = count * ns
    b_rays = torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0")
    b_a = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    nt.raygen_ao(b_rays.data_ptr(), b_a.data_ptr(), b_a.data_ptr(), d_rays.data_ptr(), d_res.data_ptr(), d_nrm.data_ptr(), first, count, ns, 40.0, 0xFFF2D5E4)
    nt.set_tunables(NTR_TRACE_AUTO_HINT=0)
    plain = torch.full((n * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
    dbvh.view.trace(K, n, True, b_rays.data_ptr(), plain.data_ptr())
    for name, cost in (("leaf depth", d_cost), ("reversed", torch.flip(d_cost, [0]).contiguous()),
                       ("random", torch.randint(0, 1000, (nblocks,), dtype=torch.int32, device="cuda:0")), ("all equal", torch.zeros_like(d_cost))):
        hint = nt.SchedHint()
        hint.predict(cost.data_ptr(), nblocks)
        for rep in range(3):   # the predicted order, then the orders refined by measurement
            got = torch.full((n * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
            dbvh.view.trace(K, n, True, b_rays.data_ptr(), got.data_ptr(), hint=hint)
            torch.cuda.synchronize()
            assert torch.equal(got, plain), (name, rep)
        hint.close()
    # a hint predicted for another block count rebinds itself at the launch
    hint = nt.SchedHint()
    hint.predict(d_cost.data_ptr(), nblocks // 2)
    got = torch.full((n * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
    dbvh.view.trace(K, n, True, b_rays.data_ptr(), got.data_ptr(), hint=hint)
    torch.cuda.synchronize()
    assert torch.equal(got, plain)
    assert nt.trace_status() == 0
