"""Every shipped kernel name -- including the two the reference implements speculatively -- returns the CPU tracer's record on
the committed counter-example to speculative traversal (tests/test_speculative_order_cpu.py), in whatever company the
victim ray shares its wave with."""
import os

import numpy as np
import pytest

import ntrace_amd as nt

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "speculative_counterexample.npz"))
RAYS = np.ascontiguousarray(G["rays"]).view(nt.RAY_DTYPE).reshape(-1)


@pytest.mark.parametrize("kernel", nt.KERNELS)
@pytest.mark.parametrize("company", ["two rays", "victim + 63 searching lanes", "victim in every fourth lane"])
def test_shipped_kernels_return_the_cpu_record(kernel, company):
    import torch
    from gpu_util import up
    if company == "two rays":
        rays = RAYS
    elif company.startswith("victim + 63"):
        rays = np.concatenate([RAYS[:1], np.repeat(RAYS[1:2], 63)])
    else:
        rays = np.tile(np.concatenate([RAYS[:1], np.repeat(RAYS[1:2], 3)]), 64)
    d_n, d_w, d_i, d_r = up(G["nodes"]), up(G["woop"]), up(G["tri_index"]), up(rays)
    view = nt.BvhView(d_n.data_ptr(), G["nodes"].nbytes, d_w.data_ptr(), G["woop"].nbytes, d_i.data_ptr())
    flags = view.validate()
    n = rays.shape[0]
    victim = (rays["ox"] == RAYS["ox"][0]) & (rays["oy"] == RAYS["oy"][0])
    for use_flags in (flags, 0):
        d_res = torch.full((n * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
        view.trace(kernel, n, False, d_r.data_ptr(), d_res.data_ptr(), 0, True, use_flags)
        got = d_res.cpu().numpy().view(nt.RESULT_DTYPE)
        assert (got["id"][victim] == G["cpu_id"][0]).all() and (got["t"].view(np.uint32)[victim] == G["cpu_t_bits"][0]).all(), \
            (kernel, company, use_flags, got["id"][victim][:4], got["t"][victim][:4])
        assert (got["id"][~victim] == G["cpu_id"][1]).all() and (got["t"].view(np.uint32)[~victim] == G["cpu_t_bits"][1]).all()
