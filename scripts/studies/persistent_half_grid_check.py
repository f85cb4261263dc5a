import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scripts'))
import numpy as np, torch
import ntrace_amd as nt
from ntrace_amd import scenes
from workloads import lbvh, scene_of, up
dev = torch.device("cuda:0")
for scene in ("hairball", "courtyard"):
    tri, pos, cam = scene_of(scene)
    best, keep = lbvh(tri, pos, 1)
    view = nt.BvhView(keep[0].data_ptr(), best.nodesBytes, keep[1].data_ptr(), best.triWoopBytes, keep[2].data_ptr())
    view.validate()
    n = 1 << 21
    d_rays = up(scenes.box_rays(pos, n, seed=21)); d_res = torch.zeros(n*16, dtype=torch.uint8, device=dev)
    prim = scenes.primary_rays(cam, 1920, 1080)[0]; d_prim = up(prim); d_pres = torch.zeros(prim.shape[0]*16, dtype=torch.uint8, device=dev)
    ref = None
    for kernel, env in (("fermi_speculative_while_while", {}), ("kepler_dynamic_fetch", {}), ("kepler_dynamic_fetch", {"NTR_TRACE_BLOCKS_PER_CU_INCOHERENT": "0"}), ("tesla_persistent_while_while", {}), ("tesla_persistent_while_while", {"NTR_TRACE_BLOCKS_PER_CU_INCOHERENT": "0"})):
        nt.set_tunables(NTR_TRACE_BLOCKS_PER_CU_INCOHERENT=None); nt.set_tunables(**env)
        ts = [view.trace(kernel, n, False, d_rays.data_ptr(), d_res.data_ptr())*1e3 for _ in range(6)]
        got = d_res.cpu().numpy().view(nt.RESULT_DTYPE).copy()
        if ref is None: ref = got
        eq = bool((got["id"] == ref["id"]).all() and (got["t"].view(np.uint32) == ref["t"].view(np.uint32)).all())
        tp = [view.trace(kernel, prim.shape[0], False, d_prim.data_ptr(), d_pres.data_ptr())*1e3 for _ in range(5)]
        print(json.dumps(dict(scene=scene, kernel=kernel, env=env, incoherent_ms=round(min(ts[2:]),4), primary_ms=round(min(tp[2:]),4), eq=eq)), flush=True)
