"""ctypes binding of include/ntrace_amd.h (libntrace_amd.so)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

RAY_DTYPE = np.dtype([("ox", "<f4"), ("oy", "<f4"), ("oz", "<f4"), ("tmin", "<f4"),
                      ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"), ("tmax", "<f4")])
RESULT_DTYPE = np.dtype([("id", "<i4"), ("t", "<f4"), ("padA", "<i4"), ("padB", "<i4")])


class NtrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ntrace_amd error %d: %s" % (code, msg))
        self.code = code


class KernelConfig(C.Structure):
    _fields_ = [("bvhLayout", C.c_int32), ("blockWidth", C.c_int32), ("blockHeight", C.c_int32),
                ("usePersistentThreads", C.c_int32)]


class TraceStats(C.Structure):
    _fields_ = [("numRays", C.c_int64), ("numInnerVisits", C.c_int64), ("numTriTests", C.c_int64),
                ("numLeafVisits", C.c_int64), ("numHits", C.c_int64)]

    def algorithmic_bytes(self):
        """DESIGN.md / SURVEY.md section 8(d): 32 B ray + 16 B result + 64 B per inner node visited +
        48 B per triangle tested + 16 B per leaf terminator + 4 B index remap per hit."""
        return (48 * self.numRays + 64 * self.numInnerVisits + 48 * self.numTriTests
                + 16 * self.numLeafVisits + 4 * self.numHits)

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class LbvhResult(C.Structure):
    _fields_ = [("numNodes", C.c_int32), ("numLeaves", C.c_int32), ("numLevels", C.c_int32), ("pad", C.c_int32),
                ("nodesBytes", C.c_int64), ("triWoopBytes", C.c_int64), ("triIndexBytes", C.c_int64),
                ("seconds", C.c_float), ("mortonMs", C.c_float), ("sortMs", C.c_float), ("woopMs", C.c_float),
                ("emitMs", C.c_float), ("refitMs", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


class _HostBvhInfo(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("nodesBytes", C.c_int64), ("triWoop", C.c_void_p),
                ("triWoopBytes", C.c_int64), ("triIndex", C.c_void_p), ("triIndexBytes", C.c_int64),
                ("layout", C.c_int32), ("numInnerNodes", C.c_int32), ("numLeafNodes", C.c_int32),
                ("maxDepth", C.c_int32), ("buildSeconds", C.c_float)]


def lib_path():
    # NTR_LIB_OVERRIDE: another build of the library (scripts/ only: a patched build for an A/B run, scripts/studies/rejected_patches/)
    return os.environ.get("NTR_LIB_OVERRIDE") or os.path.join(_HERE, "libntrace_amd.so")


_lib = None
_libs = {}   # path -> loaded CDLL (use_library switches between two builds of the library inside one process: scripts/studies/lib_ab.py)

# every symbol include/ntrace_amd.h declares: (name, restype, argtypes)
_vp, _i32, _i64, _u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
SYMBOLS = [
    ("ntr_last_error", C.c_char_p, []),
    ("ntr_version", C.c_int, []),
    ("ntr_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("ntr_set_device", C.c_int, [C.c_int]),
    ("ntr_malloc", C.c_int, [C.POINTER(_vp), C.c_size_t]),
    ("ntr_free", C.c_int, [_vp]),
    ("ntr_memcpy_h2d", C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    ("ntr_memcpy_d2h", C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    ("ntr_memcpy_d2d", C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    ("ntr_memset", C.c_int, [_vp, C.c_int, C.c_size_t, _vp]),
    ("ntr_stream_synchronize", C.c_int, [_vp]),
    ("ntr_query_config", C.c_int, [C.c_char_p, C.POINTER(KernelConfig)]),
    ("ntr_trace_bvh", C.c_int, [C.c_char_p, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _u32, _vp,
                                C.POINTER(C.c_float)]),
    ("ntr_trace_bvh_hinted", C.c_int, [C.c_char_p, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _u32, _vp,
                                       C.POINTER(C.c_float), _vp]),
    ("ntr_trace_status", C.c_int, [_vp, C.POINTER(_u32)]),
    ("ntr_trace_plan", C.c_int, [C.c_char_p, _i32, _i32, C.c_uint64, _i64, C.c_uint64, _i64, _u32, _i32, _i32, _vp]),
    ("ntr_trace_plan_hint_step", C.c_int, [_i32, _i32, _i32, C.POINTER(_i32 * 3)]),
    ("ntr_selftest_gather_rate", C.c_int, [_i64, _i32, _i32, _i32, _vp, C.POINTER(C.c_float)]),
    ("ntr_frame_shard", C.c_int, [_i32, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    ("ntr_frame_ao_batches", C.c_int, [_i32, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), _i32, C.POINTER(_i32)]),
    ("ntr_dist_unique_id", C.c_int, [C.c_char_p]),
    ("ntr_dist_init", C.c_int, [C.c_char_p, _i32, _i32, C.POINTER(_vp)]),
    ("ntr_dist_init_all", C.c_int, [_i32, C.POINTER(_i32), C.POINTER(_vp)]),
    ("ntr_dist_info", C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    ("ntr_dist_destroy", C.c_int, [_vp]),
    ("ntr_dist_broadcast", C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    ("ntr_dist_broadcast_bvh", C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp]),
    ("ntr_dist_gather_records", C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    ("ntr_dist_gather_records_cuts", C.c_int, [_vp, _vp, C.POINTER(_i32), _vp, _i32, _vp]),
    ("ntr_dist_gather_pixels", C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    ("ntr_tunables_reload", C.c_int, []),
    ("ntr_predict_block_costs", C.c_int, [_i32, _vp, _vp, _i64, _vp, _vp]),
    ("ntr_predict_batch_coherence", C.c_int, [_i32, _vp, _vp, _i64, _vp, _vp]),
    ("ntr_trace_graph_reserve", C.c_int, [_i32, _i32]),
    ("ntr_trace_graph_release_all", C.c_int, []),
    ("ntr_stream_release", C.c_int, [_vp]),
    ("ntr_selftest_auto_hint_table", C.c_int, [_i32, _i32, _i32, C.POINTER(_i32)]),
    ("ntr_lbvh_release_workspace", C.c_int, []),
    ("ntr_sched_hint_create", C.c_int, [C.POINTER(_vp)]),
    ("ntr_sched_hint_destroy", C.c_int, [_vp]),
    ("ntr_sched_hint_reset", C.c_int, [_vp]),
    ("ntr_sched_hint_predict", C.c_int, [_vp, _vp, _i32, _vp]),
    ("ntr_secondary_block_costs", C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    ("ntr_bvh_leaf_depths", C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, C.POINTER(_i32), _vp]),
    ("ntr_selftest_division", C.c_int, [_vp, _i32, _vp, _i32, C.POINTER(_u32), _vp]),
    ("ntr_selftest_division_hard", C.c_int, [_i32, _i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _vp]),
    ("ntr_trace_bvh_stats", C.c_int, [C.c_char_p, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _u32, _vp,
                                      C.POINTER(TraceStats)]),
    ("ntr_bvh_validate", C.c_int, [_vp, _i64, C.POINTER(_u32), _vp]),
    ("ntr_pixel_table", C.c_int, [_i32, _i32, _vp, _vp, _vp]),
    ("ntr_raygen_primary", C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _i32, _i32,
                                     C.c_float, _u32, _vp]),
    ("ntr_raygen_ao", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.c_float, _u32, _vp]),
    ("ntr_raygen_shadow", C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.POINTER(C.c_float * 3), C.c_float, _u32, _vp]),
    ("ntr_count_hits", C.c_int, [_vp, _i32, C.POINTER(_i32), _vp]),
    ("ntr_lbvh_capacity", C.c_int, [_i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    ("ntr_lbvh_build", C.c_int, [_i32, _vp, _i32, _vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _i32, C.c_float,
                                 _vp, _i64, _vp, _i64, _vp, _i64, C.POINTER(LbvhResult), _vp]),
    ("ntr_reconstruct", C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("ntr_ray_morton_sort", C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_float)]),
    ("ntr_camera_decode", C.c_int, [C.c_char_p, C.POINTER(C.c_float)]),
    ("ntr_camera_reencode", C.c_int, [C.c_char_p, C.c_char_p, _i32]),
    ("ntr_camera_nscreen_to_world", C.c_int, [C.c_char_p, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                              C.POINTER(C.c_float)]),
    ("ntr_obj_load", C.c_int, [C.c_char_p, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    ("ntr_obj_get", C.c_int, [_vp, _vp, _vp]),
    ("ntr_obj_free", None, [_vp]),
    ("ntr_sah_build", C.c_int, [_i32, _vp, _i32, _vp, _i32, _i32, C.POINTER(_vp)]),
    ("ntr_host_bvh_info", C.c_int, [_vp, C.POINTER(_HostBvhInfo)]),
    ("ntr_host_bvh_free", None, [_vp]),
    ("ntr_host_bvh_wrap", C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, C.POINTER(_vp)]),
    ("ntr_host_bvh_trace", C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, C.POINTER(TraceStats)]),
]


def _load(path):
    if path not in _libs:
        if not os.path.exists(path):
            raise ImportError("ntrace_amd: %s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP extension is mandatory; there is no CPU fallback)" % path)
        # When torch is in the process it must load its bundled libamdhip64 first: our library
        # then binds to that same HIP runtime (two HIP runtimes in one process do not share the
        # device).  Without torch (C++ hosts) the system /opt/rocm runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _libs[path] = L
    return _libs[path]


def lib():
    """Load libntrace_amd.so; raises if it was not built (no fallback)."""
    global _lib
    if _lib is None:
        _lib = _load(lib_path())
    return _lib


def use_library(path=None):
    """Scripts only: make `path` (default: the product library, or NTR_LIB_OVERRIDE) the library every wrapper of this module calls.
    Two builds of the library can be loaded in one process (an A/B run of a patched build); each has its own tunables and workspaces."""
    global _lib
    _lib = _load(path or lib_path())
    return _lib


def _check(rc):
    if rc != 0:
        raise NtrError(rc, (lib().ntr_last_error() or b"").decode("utf-8", "replace"))


def query_config(kernel):
    cfg = KernelConfig()
    _check(lib().ntr_query_config(kernel.encode(), C.byref(cfg)))
    return cfg


class SchedHint:
    """NtrSchedHint: block-order feedback for repeated traces of one logical batch (include/ntrace_amd.h)."""

    def __init__(self):
        self._h = _vp()
        _check(lib().ntr_sched_hint_create(C.byref(self._h)))

    def reset(self):
        _check(lib().ntr_sched_hint_reset(self._h))

    def predict(self, d_block_cost, num_blocks, stream=0):
        """ntr_sched_hint_predict: dispatch the batch's next launch by these per-block cost estimates (device pointer, uint32 per block)"""
        _check(lib().ntr_sched_hint_predict(self._h, _vp(d_block_cost), int(num_blocks), _vp(stream)))

    def close(self):
        if self._h:
            lib().ntr_sched_hint_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def trace_bvh(kernel, num_rays, any_hit, d_rays, d_results, d_nodes, nodes_bytes, d_woop, woop_bytes, d_tri_index,
              layout=4, bvh_flags=0, stream=0, timed=True, hint=None):
    """ntr_trace_bvh (ntr_trace_bvh_hinted with a SchedHint) on raw device pointers (ints).
    Returns GPU seconds if timed else None."""
    sec = C.c_float(0.0)
    args = (kernel.encode(), int(num_rays), int(bool(any_hit)), _vp(d_rays), _vp(d_results), _vp(d_nodes), int(nodes_bytes),
            _vp(d_woop), int(woop_bytes), _vp(d_tri_index), int(layout), int(bvh_flags), _vp(stream),
            C.byref(sec) if timed else None)
    if hint is None:
        _check(lib().ntr_trace_bvh(*args))
    else:
        _check(lib().ntr_trace_bvh_hinted(*args, hint._h))
    return float(sec.value) if timed else None


class TracePlan(C.Structure):
    """NtrTracePlan (include/ntrace_amd.h): what ntr_trace_bvh decides before it touches the device."""
    _fields_ = [(n, C.c_int32) for n in (
        "variant", "launchVariant", "launchBlocks", "numBlocks", "orderBlocks", "chunk", "fetchThreshold", "leafSwitchBelow", "octant",
        "flatFetch", "uniformPrologue", "splitSlice", "numHeads", "shardRays", "numBlocksIncoherent", "numBlocksDivergent", "wholeWave", "prefetchAfter", "unified", "minipool", "poolKConst",
        "poolKFromDevice", "minipoolWide", "hintable", "useAutoHint", "predictable", "persistentOrder", "probeOnRefresh", "coherentRoute",
        "persistentVariant", "persistentBlocks", "persistentFetchThreshold", "perrayBlocks", "perrayFetchThreshold")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


PLAN_FLAG_STATS, PLAN_FLAG_CAPTURING, PLAN_FLAG_CALLER_HINT = 1, 2, 4


def trace_plan(kernel, num_rays, any_hit, nodes_bytes, woop_bytes, nodes_addr=1 << 32, woop_addr=None, bvh_flags=0, num_cus=256, flags=0):
    """ntr_trace_plan: the launch plan of such a batch (no device needed)."""
    if woop_addr is None:
        woop_addr = nodes_addr + nodes_bytes
    pl = TracePlan()
    _check(lib().ntr_trace_plan(kernel.encode(), int(num_rays), int(bool(any_hit)), int(nodes_addr), int(nodes_bytes), int(woop_addr),
                                int(woop_bytes), int(bvh_flags), int(num_cus), int(flags), C.byref(pl)))
    return pl


def trace_plan_hint_step(valid, predicted, uses):
    out = (_i32 * 3)()
    _check(lib().ntr_trace_plan_hint_step(int(bool(valid)), int(bool(predicted)), int(uses), C.byref(out)))
    return dict(zeroK=bool(out[0]), refresh=bool(out[1]), useOrder=bool(out[2]))


def trace_status(stream=0):
    """ntr_trace_status: waits for `stream`, raises NtrError(NTR_ERR_OVERFLOW) if a launch since the last check
    overflowed its traversal stack; returns the status bits otherwise."""
    bits = _u32(0)
    _check(lib().ntr_trace_status(_vp(stream), C.byref(bits)))
    return int(bits.value)


def selftest_gather_rate(table_bytes, waves, lanes_per_wave=64, steps=256, stream=0):
    """ntr_selftest_gather_rate: seconds of the best of three launches of waves x lanes dependent chains of `steps` random 64-byte records."""
    sec = C.c_float(0.0)
    _check(lib().ntr_selftest_gather_rate(int(table_bytes), int(waves), int(lanes_per_wave), int(steps), _vp(stream), C.byref(sec)))
    return float(sec.value)


def frame_shard(num_primary, rank, world, align=64):
    """ntr_frame_shard: the rank-th of `world` contiguous align-aligned ranges of [0, num_primary)."""
    lo, hi = _i32(0), _i32(0)
    _check(lib().ntr_frame_shard(int(num_primary), int(rank), int(world), int(align), C.byref(lo), C.byref(hi)))
    return int(lo.value), int(hi.value)


def frame_ao_batches(lo, hi, samples, max_batch_rays):
    """ntr_frame_ao_batches: [(first input slot, inputs)] of the AO batches of the input range [lo, hi)."""
    n = _i32(0)
    _check(lib().ntr_frame_ao_batches(int(lo), int(hi), int(samples), int(max_batch_rays), None, None, 0, C.byref(n)))
    first, count = (_i32 * max(n.value, 1))(), (_i32 * max(n.value, 1))()
    _check(lib().ntr_frame_ao_batches(int(lo), int(hi), int(samples), int(max_batch_rays), first, count, n.value, C.byref(n)))
    return [(int(first[i]), int(count[i])) for i in range(n.value)]


class DistGroup:
    """ntr_dist_*: the native RCCL group of this process' GPU (one process per GPU).  `uid` = DistGroup.unique_id() of the root, handed to
    every rank by the caller."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().ntr_dist_unique_id(buf))
        return buf.raw

    def __init__(self, uid, rank, world, _handle=None):
        if _handle is not None:
            self._h, self.rank, self.world = _handle, int(rank), int(world)
            return
        h = _vp()
        _check(lib().ntr_dist_init(C.c_char_p(uid), int(rank), int(world), C.byref(h)))
        self._h, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def init_all(devices):
        """ntr_dist_init_all: one group object per device of THIS process, for one host thread per GPU (thread i: ntr_set_device(devices[i]),
        then groups[i])."""
        n = len(devices)
        devs = (_i32 * n)(*[int(x) for x in devices])
        hs = (_vp * n)()
        _check(lib().ntr_dist_init_all(n, devs, hs))
        return [DistGroup(None, i, n, _handle=_vp(hs[i])) for i in range(n)]

    def broadcast(self, d_buf, nbytes, root=0, stream=0):
        _check(lib().ntr_dist_broadcast(self._h, _vp(d_buf), int(nbytes), int(root), _vp(stream)))

    def broadcast_bvh(self, d_nodes, nodes_bytes, d_woop, woop_bytes, d_tri_index, tri_index_bytes, root=0, stream=0):
        _check(lib().ntr_dist_broadcast_bvh(self._h, _vp(d_nodes), int(nodes_bytes), _vp(d_woop), int(woop_bytes), _vp(d_tri_index), int(tri_index_bytes),
                                            int(root), _vp(stream)))

    def gather_records(self, d_own, num_primary, d_full, root=0, stream=0, align=64):
        _check(lib().ntr_dist_gather_records(self._h, _vp(d_own), int(num_primary), int(align), _vp(d_full), int(root), _vp(stream)))

    def gather_records_cuts(self, d_own, cuts, d_full, root=0, stream=0):
        """ntr_dist_gather_records_cuts: ranges cut by the host (world + 1 slot indices, the same table on every rank)."""
        if len(cuts) != self.world + 1:
            raise ValueError("gather_records_cuts: %d cut points for %d ranks" % (len(cuts), self.world))
        c = (_i32 * (self.world + 1))(*[int(x) for x in cuts])
        _check(lib().ntr_dist_gather_records_cuts(self._h, _vp(d_own), c, _vp(d_full), int(root), _vp(stream)))

    def gather_pixels(self, d_own_pixels, d_slot_to_pixel, num_primary, d_full_pixels, d_scratch, root=0, stream=0, align=64):
        _check(lib().ntr_dist_gather_pixels(self._h, _vp(d_own_pixels), _vp(d_slot_to_pixel), int(num_primary), int(align), _vp(d_full_pixels), _vp(d_scratch),
                                            int(root), _vp(stream)))

    def close(self):
        if self._h:
            lib().ntr_dist_destroy(self._h)
            self._h = None


def predict_batch_coherence(num_rays, d_rays, d_nodes, nodes_bytes, d_out, stream=0):
    """ntr_predict_batch_coherence: d_out[0] = 256-ray blocks whose sample rays start far apart, d_out[1] = the pool K derived from it."""
    _check(lib().ntr_predict_batch_coherence(int(num_rays), _vp(d_rays), _vp(d_nodes), int(nodes_bytes), _vp(d_out), _vp(stream)))


def predict_block_costs(num_rays, d_rays, d_nodes, nodes_bytes, d_block_cost, stream=0):
    """ntr_predict_block_costs: predicted cost (top-of-tree boxes hit by the sample ray) of every 256-ray block, into d_block_cost."""
    _check(lib().ntr_predict_block_costs(int(num_rays), _vp(d_rays), _vp(d_nodes), int(nodes_bytes), _vp(d_block_cost), _vp(stream)))


def trace_graph_reserve(launches, num_rays):
    """ntr_trace_graph_reserve: provision scratch for `launches` captured launches of num_rays rays."""
    _check(lib().ntr_trace_graph_reserve(int(launches), int(num_rays)))


def trace_graph_release_all():
    """ntr_trace_graph_release_all: return every resource pinned by captured launches (call when their graphs are destroyed)."""
    _check(lib().ntr_trace_graph_release_all())


def stream_release(stream=0):
    """ntr_stream_release: return the scheduling state (automatic hints, prediction scratch) `stream` owns on the current device; call
    before destroying the stream."""
    _check(lib().ntr_stream_release(_vp(stream)))


def selftest_auto_hint_table(devices, keys_per_device, rounds=3):
    """ntr_selftest_auto_hint_table: per simulated device, the batches that found their automatic hint in the last round (CPU only)."""
    out = (_i32 * devices)()
    _check(lib().ntr_selftest_auto_hint_table(devices, keys_per_device, rounds, out))
    return [int(x) for x in out]


def lbvh_release_workspace():
    _check(lib().ntr_lbvh_release_workspace())


def set_tunables(**kv):
    """Sweep helper: set NTR_* environment tunables (None removes one) and make the library re-read them."""
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    _check(lib().ntr_tunables_reload())


def bvh_leaf_depths(d_nodes, nodes_bytes, d_woop, woop_bytes, d_tri_index, num_tris, d_depth_by_tri, stream=0):
    """ntr_bvh_leaf_depths: depth of every triangle's leaf into d_depth_by_tri (int32 per triangle); returns the number of levels walked"""
    m = _i32(0)
    _check(lib().ntr_bvh_leaf_depths(_vp(d_nodes), int(nodes_bytes), _vp(d_woop), int(woop_bytes), _vp(d_tri_index), int(num_tris),
                                     _vp(d_depth_by_tri), C.byref(m), _vp(stream)))
    return int(m.value)


def secondary_block_costs(d_in_results, first, count, num_samples, d_depth_by_tri, num_tris, d_block_cost, stream=0):
    """ntr_secondary_block_costs: predicted cost of the 256-ray blocks of a secondary batch (deepest leaf among a block's input rays)"""
    _check(lib().ntr_secondary_block_costs(_vp(d_in_results), int(first), int(count), int(num_samples), _vp(d_depth_by_tri), int(num_tris),
                                           _vp(d_block_cost), _vp(stream)))


def selftest_division(d_x, nx, d_d, nd, stream=0):
    m = _u32(0)
    _check(lib().ntr_selftest_division(_vp(d_x), int(nx), _vp(d_d), int(nd), C.byref(m), _vp(stream)))
    return int(m.value)


def selftest_division_hard(x_exp, d_exp, stream=0):
    """(pairs tested, mismatches) of the FAST divide on the enumerated near-midpoint quotients (ntr_selftest_division_hard)"""
    n, m = C.c_uint64(0), C.c_uint64(0)
    _check(lib().ntr_selftest_division_hard(int(x_exp), int(d_exp), C.byref(n), C.byref(m), _vp(stream)))
    return int(n.value), int(m.value)


def trace_bvh_stats(kernel, num_rays, any_hit, d_rays, d_results, d_nodes, nodes_bytes, d_woop, woop_bytes,
                    d_tri_index, layout=4, bvh_flags=0, stream=0):
    st = TraceStats()
    _check(lib().ntr_trace_bvh_stats(kernel.encode(), int(num_rays), int(bool(any_hit)), _vp(d_rays), _vp(d_results),
                                     _vp(d_nodes), int(nodes_bytes), _vp(d_woop), int(woop_bytes), _vp(d_tri_index),
                                     int(layout), int(bvh_flags), _vp(stream), C.byref(st)))
    return st


def bvh_validate(d_nodes, nodes_bytes, stream=0):
    flags = _u32(0)
    _check(lib().ntr_bvh_validate(_vp(d_nodes), int(nodes_bytes), C.byref(flags), _vp(stream)))
    return int(flags.value)


def pixel_table(w, h, d_index_to_pixel, d_pixel_to_index=0, stream=0):
    _check(lib().ntr_pixel_table(int(w), int(h), _vp(d_index_to_pixel), _vp(d_pixel_to_index), _vp(stream)))


def raygen_primary(d_rays, d_id_to_slot, d_slot_to_id, d_index_to_pixel, origin, nscreen_to_world, w, h, max_dist,
                   kernel_seed=0, stream=0):
    o = (C.c_float * 3)(*[float(x) for x in origin])
    m = (C.c_float * 16)(*[float(x) for x in np.asarray(nscreen_to_world, dtype=np.float32).reshape(-1)])
    _check(lib().ntr_raygen_primary(_vp(d_rays), _vp(d_id_to_slot), _vp(d_slot_to_id), _vp(d_index_to_pixel), o, m,
                                    int(w), int(h), float(max_dist), int(kernel_seed), _vp(stream)))


def raygen_ao(d_out_rays, d_out_id_to_slot, d_out_slot_to_id, d_in_rays, d_in_results, d_tri_normals, first_input_slot,
              num_input_rays, num_samples, max_dist, kernel_seed=0, stream=0):
    _check(lib().ntr_raygen_ao(_vp(d_out_rays), _vp(d_out_id_to_slot), _vp(d_out_slot_to_id), _vp(d_in_rays),
                               _vp(d_in_results), _vp(d_tri_normals), int(first_input_slot), int(num_input_rays),
                               int(num_samples), float(max_dist), int(kernel_seed), _vp(stream)))


def raygen_shadow(d_out_rays, d_out_id_to_slot, d_out_slot_to_id, d_in_rays, d_in_results, first_input_slot, num_input_rays, num_samples,
                  light_pos, light_radius, kernel_seed=0, stream=0):
    lp = (C.c_float * 3)(*[float(x) for x in light_pos])
    _check(lib().ntr_raygen_shadow(_vp(d_out_rays), _vp(d_out_id_to_slot), _vp(d_out_slot_to_id), _vp(d_in_rays), _vp(d_in_results),
                                   int(first_input_slot), int(num_input_rays), int(num_samples), C.byref(lp), float(light_radius),
                                   int(kernel_seed), _vp(stream)))


def count_hits(d_results, num_rays, stream=0):
    cnt = _i32(0)
    _check(lib().ntr_count_hits(_vp(d_results), int(num_rays), C.byref(cnt), _vp(stream)))
    return int(cnt.value)


def reconstruct(ray_type, rays_per_primary, first_primary, num_primary, d_primary_slot_to_id, d_primary_results,
                d_batch_id_to_slot, d_batch_results, d_tri_material_color, d_tri_shaded_color, d_pixels, stream=0):
    _check(lib().ntr_reconstruct(int(ray_type), int(rays_per_primary), int(first_primary), int(num_primary),
                                 _vp(d_primary_slot_to_id), _vp(d_primary_results), _vp(d_batch_id_to_slot),
                                 _vp(d_batch_results), _vp(d_tri_material_color), _vp(d_tri_shaded_color), _vp(d_pixels),
                                 _vp(stream)))


def ray_morton_sort(num_rays, d_in_rays, d_in_slot_to_id, d_out_rays, d_out_id_to_slot, d_out_slot_to_id, stream=0):
    sec = C.c_float(0.0)
    _check(lib().ntr_ray_morton_sort(int(num_rays), _vp(d_in_rays), _vp(d_in_slot_to_id), _vp(d_out_rays),
                                     _vp(d_out_id_to_slot), _vp(d_out_slot_to_id), _vp(stream), C.byref(sec)))
    return float(sec.value)


def lbvh_capacity(num_tris):
    a, b, c = _i64(0), _i64(0), _i64(0)
    _check(lib().ntr_lbvh_capacity(int(num_tris), C.byref(a), C.byref(b), C.byref(c)))
    return int(a.value), int(b.value), int(c.value)


def lbvh_build(num_tris, d_tri, num_verts, d_pos, scene_min, scene_max, leaf_size, epsilon, d_nodes, nodes_cap,
               d_woop, woop_cap, d_idx, idx_cap, stream=0):
    res = LbvhResult()
    mn = (C.c_float * 3)(*[float(x) for x in scene_min])
    mx = (C.c_float * 3)(*[float(x) for x in scene_max])
    _check(lib().ntr_lbvh_build(int(num_tris), _vp(d_tri), int(num_verts), _vp(d_pos), mn, mx, int(leaf_size),
                                float(epsilon), _vp(d_nodes), int(nodes_cap), _vp(d_woop), int(woop_cap), _vp(d_idx),
                                int(idx_cap), C.byref(res), _vp(stream)))
    return res


def camera_decode(signature):
    out = (C.c_float * 16)()
    _check(lib().ntr_camera_decode(signature.encode(), out))
    v = list(out)
    return dict(position=v[0:3], forward=v[3:6], up=v[6:9], speed=v[9], fov=v[10], near=v[11], far=v[12], keep_aligned=bool(v[13]))


def camera_reencode(signature):
    buf = C.create_string_buffer(256)
    _check(lib().ntr_camera_reencode(signature.encode(), buf, 256))
    return buf.value.decode()


def camera_nscreen_to_world(signature, w, h):
    m, p, f = (C.c_float * 16)(), (C.c_float * 3)(), C.c_float(0)
    _check(lib().ntr_camera_nscreen_to_world(signature.encode(), int(w), int(h), m, p, C.byref(f)))
    return np.array(list(m), dtype=np.float32).reshape(4, 4), np.array(list(p), dtype=np.float32), float(f.value)


def obj_load(path):
    """Returns (tri[int32 n,3], pos[float32 m,3], numSubmeshes) with the reference's numbering."""
    h, nt_, nv, ns = _vp(), _i32(0), _i32(0), _i32(0)
    _check(lib().ntr_obj_load(path.encode(), C.byref(h), C.byref(nt_), C.byref(nv), C.byref(ns)))
    try:
        tri = np.zeros((nt_.value, 3), dtype=np.int32)
        pos = np.zeros((nv.value, 3), dtype=np.float32)
        _check(lib().ntr_obj_get(h, tri.ctypes.data_as(_vp), pos.ctypes.data_as(_vp)))
    finally:
        lib().ntr_obj_free(h)
    return tri, pos, int(ns.value)


class BvhView:
    """Device-side Compact BVH as raw pointers + extents (what CudaBVHTracer::traceBatch passes)."""

    def __init__(self, d_nodes, nodes_bytes, d_woop, woop_bytes, d_tri_index, flags=0, layout=4):
        self.d_nodes, self.nodes_bytes = int(d_nodes), int(nodes_bytes)
        self.d_woop, self.woop_bytes = int(d_woop), int(woop_bytes)
        self.d_tri_index, self.flags, self.layout = int(d_tri_index), int(flags), int(layout)

    def validate(self, stream=0):
        self.flags = bvh_validate(self.d_nodes, self.nodes_bytes, stream)
        return self.flags

    def trace(self, kernel, num_rays, any_hit, d_rays, d_results, stream=0, timed=True, flags=None, hint=None):
        return trace_bvh(kernel, num_rays, any_hit, d_rays, d_results, self.d_nodes, self.nodes_bytes, self.d_woop,
                         self.woop_bytes, self.d_tri_index, self.layout, self.flags if flags is None else flags,
                         stream, timed, hint)

    def trace_stats(self, kernel, num_rays, any_hit, d_rays, d_results, stream=0):
        return trace_bvh_stats(kernel, num_rays, any_hit, d_rays, d_results, self.d_nodes, self.nodes_bytes,
                               self.d_woop, self.woop_bytes, self.d_tri_index, self.layout, self.flags, stream)


class HostBvh:
    """Host Compact BVH (numpy copies of nodes / triWoop / triIndex) from ntr_sah_build."""

    def __init__(self, nodes, woop, tri_index, info=None):
        self.nodes = nodes          # uint8[nodesBytes]
        self.woop = woop            # uint8[woopBytes]
        self.tri_index = tri_index  # int32[]
        self.info = info or {}
        self.layout = 4
        self._h = None

    def host_trace(self, rays, any_hit=False, num_visibility=0, want_stats=False):
        """CudaAS::trace (the reference's host tracer, ntr_host_bvh_trace) on numpy rays; needs sah_build(keep_handle=True).
        Returns (results, visibility or None, TraceStats or None)."""
        if self._h is None:
            raise NtrError(-1, "host_trace needs sah_build(..., keep_handle=True)")
        rays = np.ascontiguousarray(rays)
        n = rays.shape[0]
        res = np.zeros(n, dtype=RESULT_DTYPE)
        vis = np.zeros(num_visibility, dtype=np.int32) if num_visibility else None
        st = TraceStats() if want_stats else None
        _check(lib().ntr_host_bvh_trace(self._h, n, int(bool(any_hit)), rays.ctypes.data_as(_vp), res.ctypes.data_as(_vp),
                                        vis.ctypes.data_as(_vp) if vis is not None else None, int(num_visibility),
                                        C.byref(st) if st is not None else None))
        return res, vis, st

    def close(self):
        if self._h is not None:
            lib().ntr_host_bvh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_bvh_wrap(nodes, woop, tri_index):
    """HostBvh (with a live handle, for host_trace) over copies of existing Compact buffers."""
    nodes = np.ascontiguousarray(nodes).view(np.uint8).reshape(-1)
    woop = np.ascontiguousarray(woop).view(np.uint8).reshape(-1)
    tri_index = np.ascontiguousarray(tri_index, dtype=np.int32).reshape(-1)
    h = _vp()
    _check(lib().ntr_host_bvh_wrap(nodes.ctypes.data_as(_vp), nodes.nbytes, woop.ctypes.data_as(_vp), woop.nbytes,
                                   tri_index.ctypes.data_as(_vp), tri_index.nbytes, C.byref(h)))
    out = HostBvh(nodes.copy(), woop.copy(), tri_index.copy())
    out._h = h
    return out


def sah_build(tri_vtx_index, vtx_pos, min_leaf=1, max_leaf=1, keep_handle=False):
    tri = np.ascontiguousarray(tri_vtx_index, dtype=np.int32).reshape(-1, 3)
    pos = np.ascontiguousarray(vtx_pos, dtype=np.float32).reshape(-1, 3)
    h = _vp()
    _check(lib().ntr_sah_build(tri.shape[0], tri.ctypes.data_as(_vp), pos.shape[0], pos.ctypes.data_as(_vp),
                               int(min_leaf), int(max_leaf), C.byref(h)))
    try:
        info = _HostBvhInfo()
        _check(lib().ntr_host_bvh_info(h, C.byref(info)))
        nodes = np.ctypeslib.as_array(C.cast(info.nodes, C.POINTER(C.c_uint8)), (info.nodesBytes,)).copy()
        woop = np.ctypeslib.as_array(C.cast(info.triWoop, C.POINTER(C.c_uint8)), (info.triWoopBytes,)).copy()
        tidx = np.ctypeslib.as_array(C.cast(info.triIndex, C.POINTER(C.c_int32)), (info.triIndexBytes // 4,)).copy()
        meta = dict(numInnerNodes=info.numInnerNodes, numLeafNodes=info.numLeafNodes, maxDepth=info.maxDepth,
                    buildSeconds=float(info.buildSeconds))
    except Exception:
        lib().ntr_host_bvh_free(h)
        raise
    out = HostBvh(nodes, woop, tidx, meta)
    if keep_handle:
        out._h = h
    else:
        lib().ntr_host_bvh_free(h)
    return out
