"""Short run of the randomised differential test (tests/fuzz_parity.py): random scenes, SAH and
on-device LBVH trees, ray mixes with zero / tiny / non-finite components, every kernel name, closest-hit
and any-hit; hit records, traversal counters and LBVH trees must match the CPU oracle exactly."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_parity_short(seed):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_parity
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = fuzz_parity.main(["--seconds", "12", "--seed", str(seed), "--rays", "60000"])
    out = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert rc == 0, out["failures"]
    assert out["rounds"] >= 3 and out["rays_compared"] > 100000   # (time-boxed: the round count depends on the draws)
    assert out["record_mismatches"] == 0 and out["counter_mismatches"] == 0 and out["lbvh_tree_mismatch"] == 0
