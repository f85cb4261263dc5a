"""Hand-derived known-answer vectors (tests/kat_vectors.py) on the HIP kernels, through the C-ABI: every kernel name,
closest hit and any hit, the generic and the fast slab path, plus the traversal counters that decide the accept-rule cases."""
import numpy as np
import pytest

import kat_vectors as kat
import ntrace_amd as nt

pytestmark = pytest.mark.gpu
CASES = kat.cases()


@pytest.mark.parametrize("kernel", nt.KERNELS)
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_known_answers_hip(case, kernel):
    import torch
    from gpu_util import up
    name, nodes, woop, idx, rays, exp = case
    d_n, d_w, d_i, d_r = up(nodes), up(woop), up(idx), up(rays)
    view = nt.BvhView(d_n.data_ptr(), nodes.nbytes, d_w.data_ptr(), woop.nbytes, d_i.data_ptr())
    flags = view.validate()
    n = rays.shape[0]
    for use_flags in (flags, 0):  # fast exact-divide path where the rays allow it, and the generic path
        for any_hit, key in ((False, "closest"), (True, "any")):
            d_res = torch.full((n * 16,), 0xCD, dtype=torch.uint8, device="cuda:0")
            view.trace(kernel, n, any_hit, d_r.data_ptr(), d_res.data_ptr(), 0, True, use_flags)
            got = d_res.cpu().numpy().view(nt.RESULT_DTYPE)
            for i, e in enumerate(exp):
                want_id, want_bits = e[key]
                assert int(got["id"][i]) == want_id and int(got["t"][i:i + 1].view(np.uint32)[0]) == want_bits, \
                    "%s %s ray %d (%s, flags %d): got (%d, 0x%08X), derived (%d, 0x%08X)" % (
                        name, kernel, i, key, use_flags, int(got["id"][i]), int(got["t"][i:i + 1].view(np.uint32)[0]), want_id, want_bits)
    for i, e in enumerate(exp):
        if e["inner"] is None:
            continue
        d_res = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
        st = view.trace_stats(kernel, 1, False, d_r.data_ptr() + 32 * i, d_res.data_ptr())
        assert (st.numInnerVisits, st.numTriTests) == (e["inner"], e["tris"]), (name, i, st.as_dict())
