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
