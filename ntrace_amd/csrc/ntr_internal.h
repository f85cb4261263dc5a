// ntr_internal.h -- helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "ntrace_amd.h"

namespace ntr {
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t e, const char* what);

// Tunables (environment overrides for sweeps), read once at first use -- see ntr_api.cpp.
struct Tunables {
    int chunk, fetchThreshold, leafSwitchBelow, blocksPerCU, blocksPerCUIncoherent, blocksPerCUDivergent, poolHeads, octant, anyHitWaves, closestWaves, unified, perrayUnified, flatFetch, uniformPrologue, splitSlice, wholeWave, prefetchAfter, minipool, minipoolThreshold, minipoolWide;
    int autoHint, autoHintMinRays, persistentHints, route, predict, predictPersistent, predictDepth, predictMinRays, predictMinNodes;
    int schedRefreshEvery, schedClasses;
    int lbvhSplit, lbvhSubThreads, lbvhAggLds, lbvhAggStaged, lbvhSortItems, lbvhMortonKeys, lbvhMortonThreads, lbvhMarkThreads;
};
Tunables tunables();

// true when `s` is being captured into a HIP graph: scratch handed to such a launch must stay valid for the
// lifetime of the graph, so it is taken from never-recycled storage.
bool stream_is_capturing(hipStream_t s);

// rayops_kernels.hip: returns the per-device scratch of ntr_ray_morton_sort (ntr_lbvh_release_workspace calls it)
int raysort_scratch_release();
}  // namespace ntr

// ntr_api.cpp: (re)build the top-of-tree box table cached for this node buffer (dispatch-order prediction)
extern "C" int ntr_top_table_refresh(const void* d_nodes, int64_t nodesBytes, void* stream);

#define NTR_HIP(call)                                            \
    do {                                                         \
        hipError_t _e = (call);                                  \
        if (_e != hipSuccess) return ::ntr::hip_fail(_e, #call); \
    } while (0)
