"""Device-buffer plumbing for the GPU parity tests (torch is only the allocator here)."""
import numpy as np
import torch

import ntrace_amd as nt


def up(a, dev="cuda:0"):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


class DeviceBvh:
    def __init__(self, bvh, dev="cuda:0"):
        self.host = bvh
        self.nodes, self.woop, self.idx = up(bvh.nodes, dev), up(bvh.woop, dev), up(bvh.tri_index, dev)
        self.view = nt.BvhView(self.nodes.data_ptr(), bvh.nodes.nbytes, self.woop.data_ptr(), bvh.woop.nbytes,
                               self.idx.data_ptr())
        self.flags = self.view.validate()


def gpu_trace(kernel, dbvh, rays, any_hit=False, flags=None, prefill=0xCD):
    n = rays.shape[0]
    d_rays = up(rays) if n else torch.zeros(32, dtype=torch.uint8, device="cuda:0")
    d_res = torch.full((max(n, 1) * 16,), prefill, dtype=torch.uint8, device="cuda:0")
    sec = dbvh.view.trace(kernel, n, any_hit, d_rays.data_ptr(), d_res.data_ptr(),
                          torch.cuda.current_stream().cuda_stream, True, flags)
    torch.cuda.synchronize()
    return d_res.cpu().numpy().view(nt.RESULT_DTYPE)[:n], sec


def assert_parity(got, ref, what=""):
    bad_id = np.nonzero(got["id"] != ref["id"])[0]
    bad_t = np.nonzero(got["t"].view(np.uint32) != ref["t"].view(np.uint32))[0]
    assert bad_id.size == 0 and bad_t.size == 0, (
        "%s: %d id / %d t mismatches of %d rays; first id mismatch %s first t mismatch %s"
        % (what, bad_id.size, bad_t.size, got.shape[0],
           (int(bad_id[0]), int(got["id"][bad_id[0]]), int(ref["id"][bad_id[0]])) if bad_id.size else None,
           (int(bad_t[0]), float(got["t"][bad_t[0]]), float(ref["t"][bad_t[0]])) if bad_t.size else None))
