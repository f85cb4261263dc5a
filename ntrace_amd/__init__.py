"""ntrace_amd -- MI355X-native tracer backend for NTrace (BVH trace + LBVH build).

The product is libntrace_amd.so (hand-written HIP kernels for gfx950 behind the C-ABI
of include/ntrace_amd.h) plus the C++ host mirror of the reference's Renderer / CudaBVH /
RayBuffer classes in ntrace_amd/host.  This Python module is plumbing for tests and
bench.py: it loads the library with ctypes and passes raw device pointers (e.g. torch
tensors' data_ptr()).  There is no CPU or PyTorch fallback: if the library is missing the
import of `ntrace_amd.lib()` fails loudly.
"""
from ._capi import (BvhView, RAY_DTYPE, RESULT_DTYPE, HostBvh, KernelConfig, NtrError, TraceStats, bvh_validate, lib,
                    lib_path, query_config, sah_build, trace_bvh, trace_bvh_stats, pixel_table,
                    raygen_primary, raygen_ao, raygen_shadow, count_hits, selftest_division, selftest_division_hard, bvh_leaf_depths, secondary_block_costs, lbvh_capacity,
                    lbvh_build, LbvhResult, reconstruct, ray_morton_sort, camera_decode, camera_reencode,
                    camera_nscreen_to_world, obj_load, SchedHint, trace_status, trace_plan, trace_plan_hint_step, TracePlan, set_tunables, host_bvh_wrap, use_library, trace_graph_reserve, trace_graph_release_all, stream_release, selftest_auto_hint_table, selftest_gather_rate, frame_shard, frame_ao_batches, DistGroup,
                    lbvh_release_workspace, predict_block_costs, predict_batch_coherence)

BVHLayout_Compact = 4
BVH_FINITE, BVH_FASTDIV, BVH_NOTINY, BVH_ORDERED, BVH_WIDE_LEAVES = 1, 2, 4, 8, 16
KERNELS = ("fermi_speculative_while_while", "tesla_persistent_while_while",
           "tesla_persistent_speculative_while_while", "kepler_dynamic_fetch")

__all__ = ["BvhView", "RAY_DTYPE", "RESULT_DTYPE", "HostBvh", "KernelConfig", "NtrError", "bvh_validate", "lib",
           "lib_path", "query_config", "sah_build", "trace_bvh", "trace_bvh_stats", "TraceStats", "pixel_table", "raygen_primary", "raygen_ao", "count_hits", "selftest_division", "selftest_division_hard", "bvh_leaf_depths", "secondary_block_costs", "lbvh_capacity", "lbvh_build", "LbvhResult", "reconstruct", "ray_morton_sort", "camera_decode", "camera_reencode", "camera_nscreen_to_world", "obj_load", "SchedHint", "trace_status", "set_tunables", "host_bvh_wrap", "BVHLayout_Compact", "KERNELS"]
