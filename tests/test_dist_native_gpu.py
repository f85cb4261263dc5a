"""The native multi-GPU part of the C-ABI (ntr_dist_*, RCCL bound at run time) on the one GPU of the test box: a world-size-1 group
runs every call through RCCL (unique id, communicator, broadcast, the grouped gather of records and of pixels)."""
import numpy as np
import pytest

import ntrace_amd as nt

pytestmark = pytest.mark.gpu


def test_native_group_of_one_rank_broadcasts_and_gathers():
    import torch
    dev = torch.device("cuda:0")
    uid = nt.DistGroup.unique_id()
    assert len(uid) == 128
    g = nt.DistGroup(uid, 0, 1)
    try:
        s = torch.cuda.current_stream().cuda_stream
        buf = torch.arange(100000, dtype=torch.int32, device=dev)
        g.broadcast(buf.data_ptr(), buf.numel() * 4, 0, s)
        torch.cuda.synchronize()
        assert torch.equal(buf.cpu(), torch.arange(100000, dtype=torch.int32))
        w, h = 200, 120
        n = w * h
        rec = torch.randint(0, 2 ** 31 - 1, (n * 4,), dtype=torch.int32, device=dev)
        full = torch.zeros_like(rec)
        g.gather_records(rec.data_ptr(), n, full.data_ptr(), 0, s)
        torch.cuda.synchronize()
        assert torch.equal(full, rec)
        # pixels: packed in slot order through the PixelTable's index-to-pixel map, scattered back on the root
        i2p = torch.zeros(n, dtype=torch.int32, device=dev)
        nt.pixel_table(w, h, i2p.data_ptr(), 0, s)
        px = torch.randint(1, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev)
        out = torch.zeros_like(px)
        scratch = torch.zeros_like(px)
        g.gather_pixels(px.data_ptr(), i2p.data_ptr(), n, out.data_ptr(), scratch.data_ptr(), 0, s)
        torch.cuda.synchronize()
        assert torch.equal(out, px)
        assert torch.equal(scratch.cpu(), px.cpu()[i2p.cpu().long()])   # what travelled: the pixels in slot order
    finally:
        g.close()


def _gpu_count():
    import torch
    return torch.cuda.device_count()   # (counts devices without initialising the GPU)


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: the non-root receive offsets of ntr_dist_gather_records* execute only at world size >= 2 "
                                              "(on a one-GPU box the N-rank flow runs host-staged instead: test_bench_obj_gpu.py)")
def test_native_gather_two_devices_two_threads_unequal_shards():
    """SURVEY 8(e) on two devices of one process, one host thread per GPU (ntr_dist_init_all): BVH replicated by ntr_dist_broadcast_bvh, the
    frame cut into UNEQUAL ranges (a quarter / three quarters, then ntr_frame_shard's equal ranges), every thread traces its own range on its
    own device, the root gathers through RCCL -- and its assembled frame equals a single-GPU trace of the whole frame bit for bit."""
    import threading
    import torch
    from ntrace_amd import scenes
    from gpu_util import up
    tri, pos, cam = scenes.random_soup(20000, seed=5, walls=True)
    bvh = nt.sah_build(tri, pos, 1, 1)
    w, h = 320, 200
    rays, _ = scenes.primary_rays(cam, w, h)
    n = rays.shape[0]
    K = "fermi_speculative_while_while"
    # the whole frame on device 0 alone
    d0 = torch.device("cuda:0")
    with torch.cuda.device(0):
        d_n, d_w, d_i, d_r = up(bvh.nodes), up(bvh.woop), up(bvh.tri_index), up(rays)
        view = nt.BvhView(d_n.data_ptr(), bvh.nodes.nbytes, d_w.data_ptr(), bvh.woop.nbytes, d_i.data_ptr())
        view.validate()
        ref = torch.zeros(n * 16, dtype=torch.uint8, device=d0)
        view.trace(K, n, False, d_r.data_ptr(), ref.data_ptr())
        torch.cuda.synchronize()
    groups = nt.DistGroup.init_all([0, 1])
    cuts_list = [[0, (n // 4) // 64 * 64, n], None]     # unequal host cuts, then the library's own equal ranges
    full = {}
    errors = []
    bar = threading.Barrier(2)

    def rank_main(r):
        try:
            nt.set_device(r)
            dev = torch.device("cuda", r)
            with torch.cuda.device(r):
                g = groups[r]
                s = torch.cuda.current_stream().cuda_stream
                # BVH replication: the root's buffers to every rank (sizes are known to all: same host build)
                if r == 0:
                    bn, bw, bi = d_n, d_w, d_i
                else:
                    bn = torch.zeros(bvh.nodes.nbytes, dtype=torch.uint8, device=dev)
                    bw = torch.zeros(bvh.woop.nbytes, dtype=torch.uint8, device=dev)
                    bi = torch.zeros(bvh.tri_index.nbytes, dtype=torch.uint8, device=dev)
                g.broadcast_bvh(bn.data_ptr(), bn.numel(), bw.data_ptr(), bw.numel(), bi.data_ptr(), bi.numel(), 0, s)
                torch.cuda.synchronize()
                v = nt.BvhView(bn.data_ptr(), bn.numel(), bw.data_ptr(), bw.numel(), bi.data_ptr())
                v.validate()
                rr = torch.from_numpy(rays.view(np.uint8).reshape(-1).copy()).to(dev)
                for ci, cuts in enumerate(cuts_list):
                    lo, hi = (cuts[r], cuts[r + 1]) if cuts else nt.frame_shard(n, r, 2, 64)
                    own = torch.zeros(max(hi - lo, 1) * 16, dtype=torch.uint8, device=dev)
                    if hi > lo:
                        v.trace(K, hi - lo, False, rr.data_ptr() + lo * 32, own.data_ptr())
                    out = torch.zeros(n * 16, dtype=torch.uint8, device=dev) if r == 0 else None
                    bar.wait()
                    if cuts:
                        g.gather_records_cuts(own.data_ptr(), cuts, out.data_ptr() if r == 0 else 0, 0, s)
                    else:
                        g.gather_records(own.data_ptr(), n, out.data_ptr() if r == 0 else 0, 0, s)
                    torch.cuda.synchronize()
                    bar.wait()
                    if r == 0:
                        full[ci] = out.cpu()
        except Exception as e:      # a thread that dies must not leave its peer waiting in a collective forever
            errors.append((r, repr(e)))
            bar.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    for g in groups:
        g.close()
    assert not errors, errors
    for ci in range(len(cuts_list)):
        assert torch.equal(full[ci], ref.cpu()), "cuts %r: the gathered frame differs from the single-GPU frame" % (cuts_list[ci],)


def test_gather_records_cuts_world_one_and_bad_tables():
    """The cut-table gather at world size 1 (the root's own copy) and its argument checks (uniform across ranks by construction)."""
    import torch
    dev = torch.device("cuda:0")
    g = nt.DistGroup(nt.DistGroup.unique_id(), 0, 1)
    try:
        s = torch.cuda.current_stream().cuda_stream
        n = 1000
        rec = torch.randint(0, 2 ** 31 - 1, (n * 4,), dtype=torch.int32, device=dev)
        full = torch.zeros_like(rec)
        g.gather_records_cuts(rec.data_ptr(), [0, n], full.data_ptr(), 0, s)
        torch.cuda.synchronize()
        assert torch.equal(full, rec)
        with pytest.raises(nt.NtrError):
            g.gather_records_cuts(rec.data_ptr(), [64, n], full.data_ptr(), 0, s)
        with pytest.raises(ValueError):
            g.gather_records_cuts(rec.data_ptr(), [0, 64, n], full.data_ptr(), 0, s)
    finally:
        g.close()
