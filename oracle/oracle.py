"""ctypes binding of the CPU oracle (oracle/libntr_oracle.so).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never by the product package (ntrace_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libntr_oracle.so")

RAY_DTYPE = np.dtype([("ox", "<f4"), ("oy", "<f4"), ("oz", "<f4"), ("tmin", "<f4"),
                      ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"), ("tmax", "<f4")])
RESULT_DTYPE = np.dtype([("id", "<i4"), ("t", "<f4"), ("padA", "<i4"), ("padB", "<i4")])


class TraceStats(C.Structure):
    _fields_ = [("numRays", C.c_int64), ("numInnerVisits", C.c_int64), ("numTriTests", C.c_int64),
                ("numLeafVisits", C.c_int64), ("numHits", C.c_int64), ("maxStackDepth", C.c_int64)]

    def algorithmic_bytes(self):
        """32 (ray) + 16 (result) + 64*I + 48*T + 16*L + 4*H, summed."""
        return (48 * self.numRays + 64 * self.numInnerVisits + 48 * self.numTriTests
                + 16 * self.numLeafVisits + 4 * self.numHits)

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Lbvh(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("nodesBytes", C.c_int64), ("woop", C.c_void_p), ("woopBytes", C.c_int64),
                ("triIndex", C.c_void_p), ("triIndexBytes", C.c_int64), ("mortonSorted", C.c_void_p),
                ("triSorted", C.c_void_p), ("numInner", C.c_int32), ("numLeaves", C.c_int32), ("numLevels", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.orc_trace_compact.argtypes = [vp, vp, vp, vp, vp, i32, i32, C.POINTER(TraceStats)]
        L.orc_trace_compact.restype = C.c_int
        L.orc_trace_compact_mt.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(TraceStats)]
        L.orc_trace_compact_mt.restype = C.c_int
        L.orc_bruteforce_closest.argtypes = [vp, vp, i32, vp, vp, i32]
        L.orc_bruteforce_closest.restype = None
        L.orc_lbvh_build.argtypes = [i32, vp, i32, vp, vp, vp, i32, C.c_float, C.POINTER(Lbvh)]
        L.orc_lbvh_build.restype = C.c_int
        L.orc_lbvh_free.argtypes = [C.POINTER(Lbvh)]
        L.orc_lbvh_free.restype = None
        L.orc_bvh_canonical_hash.argtypes = [vp, i64, vp, vp, i32]
        L.orc_bvh_canonical_hash.restype = C.c_uint64
        L.orc_lbvh_morton.argtypes = [i32, vp, vp, vp, vp, vp, vp]
        L.orc_lbvh_morton.restype = None
        L.orc_lbvh_woop.argtypes = [i32, vp, vp, vp]
        L.orc_lbvh_woop.restype = None
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _as_rays(rays):
    rays = np.ascontiguousarray(rays)
    if rays.dtype != RAY_DTYPE:
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8).view(RAY_DTYPE).reshape(-1)
    return rays


def trace(nodes, woop, tri_index, rays, any_hit=False, threads=1, results=None):
    """CudaBVH::trace on Compact buffers.  Returns (results[RESULT_DTYPE], TraceStats)."""
    nodes = np.ascontiguousarray(nodes)
    woop = np.ascontiguousarray(woop)
    tri_index = np.ascontiguousarray(tri_index, dtype=np.int32)
    rays = _as_rays(rays)
    n = rays.shape[0]
    if results is None:
        results = np.zeros(n, dtype=RESULT_DTYPE)
    st = TraceStats()
    if threads <= 1:
        rc = lib().orc_trace_compact(_ptr(nodes), _ptr(woop), _ptr(tri_index), _ptr(rays), _ptr(results),
                                     n, int(bool(any_hit)), C.byref(st))
    else:
        rc = lib().orc_trace_compact_mt(_ptr(nodes), _ptr(woop), _ptr(tri_index), _ptr(rays), _ptr(results),
                                        n, int(bool(any_hit)), int(threads), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle: traversal stack overflow (reference stack is 100 entries)")
    return results, st


def bruteforce_closest(woop, tri_index, rays):
    woop = np.ascontiguousarray(woop)
    tri_index = np.ascontiguousarray(tri_index, dtype=np.int32)
    rays = _as_rays(rays)
    res = np.zeros(rays.shape[0], dtype=RESULT_DTYPE)
    lib().orc_bruteforce_closest(_ptr(woop), _ptr(tri_index), woop.nbytes // 16, _ptr(rays), _ptr(res), rays.shape[0])
    return res


def scene_bbox(pos):
    """Scene::getBBox (src/rt/Scene.cpp:112-126): min/max over the vertex positions."""
    pos = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
    return pos.min(axis=0), pos.max(axis=0)


def lbvh_build(tri, pos, leaf_size=8, epsilon=0.001, bbox=None):
    """HLBVHBuilder::buildLBVH restated on the CPU.  Returns dict(nodes, woop, tri_index, morton_sorted,
    tri_sorted, num_inner, num_leaves, num_levels) with numpy copies of the Compact buffers."""
    tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    mn, mx = bbox if bbox is not None else scene_bbox(pos)
    mn = np.ascontiguousarray(mn, dtype=np.float32)
    mx = np.ascontiguousarray(mx, dtype=np.float32)
    b = Lbvh()
    rc = lib().orc_lbvh_build(tri.shape[0], _ptr(tri), pos.shape[0], _ptr(pos), _ptr(mn), _ptr(mx), int(leaf_size),
                              float(epsilon), C.byref(b))
    if rc != 0:
        raise RuntimeError("oracle: orc_lbvh_build failed")
    try:
        def arr(p, nbytes, dt):
            return np.frombuffer(C.string_at(p, nbytes), dtype=dt).copy()
        out = dict(nodes=arr(b.nodes, b.nodesBytes, np.uint8), woop=arr(b.woop, b.woopBytes, np.uint8),
                   tri_index=arr(b.triIndex, b.triIndexBytes, np.int32),
                   morton_sorted=arr(b.mortonSorted, tri.shape[0] * 4, np.uint32),
                   tri_sorted=arr(b.triSorted, tri.shape[0] * 4, np.int32),
                   num_inner=int(b.numInner), num_leaves=int(b.numLeaves), num_levels=int(b.numLevels))
    finally:
        lib().orc_lbvh_free(C.byref(b))
    return out


def bvh_canonical_hash(nodes, woop, tri_index, hash_woop=True):
    nodes = np.ascontiguousarray(nodes)
    woop = np.ascontiguousarray(woop)
    tri_index = np.ascontiguousarray(tri_index, dtype=np.int32)
    return int(lib().orc_bvh_canonical_hash(_ptr(nodes), nodes.nbytes, _ptr(woop), _ptr(tri_index), int(bool(hash_woop))))
