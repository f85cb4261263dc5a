// trace_kernels.h -- launch contract between ntr_api.cpp and trace_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntrace_amd.h"

#ifndef NTR_TRACE_WAVES_PER_BLOCK
#define NTR_TRACE_WAVES_PER_BLOCK 4  // 256-thread workgroups
#endif
#ifndef NTR_TRACE_MIN_WAVES_PER_SIMD
#define NTR_TRACE_MIN_WAVES_PER_SIMD 1
#endif
// The persistent kernels take the registers the compiler gives them: 69 (unified-step loop) / 62 (while-while), i.e. seven 256-thread
// workgroups per CU.  Held to 64 (-DNTR_TRACE_PERSISTENT_MIN_WAVES_PER_SIMD=8: no spill, eight workgroups per CU) they are no faster on
// coherent batches (not occupancy-bound, EXPERIMENTS.md round 6) and 2-4 % slower on the incoherent batches they exist for under routing
// (courtyard box rays 3.55-3.73 against 3.48-3.59 ms).
#ifndef NTR_TRACE_PERSISTENT_MIN_WAVES_PER_SIMD
#define NTR_TRACE_PERSISTENT_MIN_WAVES_PER_SIMD 1
#endif

// kernel variants (selected by the reference's kernel file names, see ntr_query_config)
#define NTR_VARIANT_PERRAY 0      // one ray per lane, while-while
#define NTR_VARIANT_PERSISTENT 1  // persistent waves, ballot/mbcnt dynamic fetch
#define NTR_VARIANT_PERRAY_STATS 2 // per-ray kernel + traversal counters
#define NTR_VARIANT_PERRAY_W2 3    // per-ray kernel in 128-thread workgroups (any-hit launches)
#define NTR_VARIANT_PERRAY_W1 4    // per-ray kernel in 64-thread workgroups
#define NTR_VARIANT_PERSISTENT_UNIFIED 5  // persistent waves, unified-step loop (every live lane advances each iteration)
#define NTR_VARIANT_PERRAY_UNIFIED_W1 6   // per-ray kernel, 64-thread workgroups, unified-step loop
#define NTR_MINIPOOL_MAX_K 16             // a wave's private pool: at most this many 64-ray chunks
#define NTR_VARIANT_PERRAY_UNIFIED_MINI 7 // the same launch, which runs as the wave-private mini-pool instead when the batch's pool K (TraceParams::poolK,
                                          // decided on the device) is 2 ... 16: a wave owns K x 64 rays and refills its finished lanes from them

#define NTR_ROUTE_SKIP_COHERENT 1
#define NTR_ROUTE_SKIP_INCOHERENT 2
// the batch word (sched_kernels.hip pool_k): bits 0-15 the mini-pool depth K (> 1: origins scattered), bit 16 NTR_BATCH_DIVERGENT
#define NTR_BATCH_WORD_K(w) ((w) & 0xFFFFu)
#define NTR_BATCH_WORD_INCOHERENT(w) (NTR_BATCH_WORD_K(w) > 1u || ((w) & NTR_BATCH_DIVERGENT) != 0u)

// bits of the device status word
#define NTR_STATUS_STACK_OVERFLOW 1u


namespace ntr {

struct TraceParams {
    int32_t numRays;
    int32_t anyHit;
    const NtrRay* rays;
    NtrRayResult* results;
    const void* nodes;
    const void* woop;
    uint32_t nodesBytes;  // buffer-descriptor ranges (out-of-range loads return 0)
    uint32_t woopBytes;
    const int32_t* triIndex;
    int32_t* counter;        // persistent: numHeads pool heads 64 B apart (zeroed on the stream before launch)
    unsigned int* status;    // sticky error bits
    int32_t chunk;           // persistent: rays per pool grab
    int32_t shardRays;       // persistent: rays per pool shard (numHeads shards, a multiple of chunk)
    int32_t numHeads;        // persistent: pool heads, a multiple of 8 (one group per XCD), <= 1024
    int32_t numBlocks;       // persistent: grid size (the statically assigned first chunks are counted from it)
    int32_t numBlocksIncoherent;   // persistent: the grid that works on a batch whose pool word (poolK) says incoherent (0 = the whole grid)
    int32_t numBlocksDivergent;    // ... on a batch whose word only carries NTR_BATCH_DIVERGENT (rays that start together and wander apart)
    int32_t orderBlocks;     // persistent, with `order`: number of 256-ray blocks in order[]
    int32_t fetchThreshold;  // persistent: refill when fewer lanes are live
    int32_t wholeWave;       // persistent, dynamic fetch: 1 = single-lane refills only on batches the device found incoherent (poolK > 1), whole-wave
                             // refills otherwise; 0 = dynamic fetch always (as until round 5)
    int32_t prefetchAfter;   // persistent: iterations into a chunk after which the wave posts the dequeue of its next one (< 0: no dequeue-ahead)
    uint32_t bvhFlags;
    int32_t leafSwitchBelow; // serve waiting leaves when fewer lanes than this still hold an inner node
    int32_t octant;          // per-ray kernel: specialise the slab test for waves whose rays share their direction signs
    int32_t flatFetch;       // unified-step loop: one group of global loads for nodes and triangles (needs both extents >= 64 bytes)
    int32_t uniformPrologue; // per-ray kernels, unified-step loop: scalar node fetches while every live lane of the wave holds the same inner node
    int32_t splitSlice;      // persistent kernels, unified-step loop: once the pool is dry, idle lanes take over stack entries of the wave's
                             // live rays; the lanes are looked at every splitSlice steps (trace_split.h); 0 = off
    const unsigned int* order;     // per-ray kernel: workgroup i traces ray block order[i] (null = identity); persistent kernels: the pool
                                   // hands the 256-ray blocks out in this order
    unsigned int* cost;            // per-ray kernel: cost[block] = max wave lifetime in 10 ns ticks (null = off)
    unsigned long long* stats;  // STATS variant: {innerVisits, triTests, leafVisits, hits}
    const unsigned int* poolK;  // mini-pool kernel: device word holding the rays a wave owns / 64 (1 .. NTR_MINIPOOL_MAX_K; anything else reads as 1),
                                // written by the dispatch-order prediction of this launch or kept in the launch's hint; null = poolKConst
    int32_t poolKConst;
    int32_t routeSkip;          // batch routing by the device's coherence word (*poolK; ntr_api.cpp launches BOTH bodies for such a batch): 0 = none,
                                // NTR_ROUTE_SKIP_COHERENT = this launch leaves at once when the word says coherent (the persistent bodies),
                                // NTR_ROUTE_SKIP_INCOHERENT = ... when it says incoherent (the per-ray body)
};

}  // namespace ntr

extern "C" hipError_t ntr_launch_trace(int variant, const ntr::TraceParams* p, int numBlocks, hipStream_t stream);
// dispatch-order prediction (sched_kernels.hip)
#define NTR_TOP_DEPTH_MAX 10          // top-of-tree table: child boxes of the nodes above this depth
#define NTR_SCHED_PRED_CLASSES 64     // cost classes of the predictor (one per lane in the flatten step)
#define NTR_SCHED_PRED_WORDS (NTR_SCHED_PRED_CLASSES + 3)   // + blocks whose two sample rays start far apart, blocks whose rays point apart, pool K
extern "C" hipError_t ntr_launch_top_table(const void* d_nodes, unsigned int nodesBytes, int depth, void* d_table,
                                           unsigned int* d_tableCount, hipStream_t stream);
extern "C" hipError_t ntr_launch_predict(const void* d_rays, int numRays, int numBlocks, const void* d_table,
                                         const unsigned int* d_tableCount, unsigned int* d_classCount, unsigned int* d_classList,
                                         unsigned int* d_order, unsigned int* d_poolKCopy, int poolKWide, hipStream_t stream);
// cost query: d_blockCost[b] = boxes of the top-of-tree table the sample ray of 256-ray block b intersects
extern "C" hipError_t ntr_launch_predict_costs(const void* d_rays, int numRays, int numBlocks, const void* d_table,
                                               const unsigned int* d_tableCount, unsigned int* d_blockCost, hipStream_t stream);
// coherence query: d_out[0] = 256-ray blocks whose two sample rays start further apart than 1/8 of the scene extent, d_out[1] = blocks
// whose sample rays start together and point more than 60 degrees apart, d_out[2] = pool K
extern "C" hipError_t ntr_launch_coherence(const void* d_rays, int numRays, int numBlocks, const void* d_table, const unsigned int* d_tableCount,
                                           unsigned int* d_out, int poolKWide, hipStream_t stream);
// clears 32-bit words with a kernel (graph-replay safe, unlike a memset node)
extern "C" hipError_t ntr_launch_leaf_depth_level(const void* d_nodes, unsigned int nodesBytes, const void* d_woop, unsigned int woopVec4, const int* d_triIndex,
                                                  int numTris, const unsigned int* d_qin, const unsigned int* d_nIn, unsigned int* d_qout, unsigned int* d_nOut,
                                                  unsigned int capacity, unsigned int gridThreads, int depth, int* d_depthByTri, hipStream_t stream);
extern "C" hipError_t ntr_launch_secondary_block_costs(const void* d_inResults, int first, int count, int numSamples, const int* d_depthByTri, int numTris,
                                                       unsigned int* d_blockCost, hipStream_t stream);
extern "C" hipError_t ntr_launch_zero_words(void* d_ptr, int words, hipStream_t stream);
// out[0] = atomicExch(status, 0): fetch-and-clear of the sticky status word in one device-side step
extern "C" hipError_t ntr_launch_status_exchange(unsigned int* d_status, unsigned int* d_out, hipStream_t stream);
extern "C" hipError_t ntr_launch_sched_order(const unsigned int* d_cost, int numBlocks, int classes, unsigned int* d_order,
                                             hipStream_t stream);
extern "C" hipError_t ntr_launch_selftest_division(const float* d_x, const float* d_d, int nx, int nd,
                                                   unsigned int* d_mismatches, hipStream_t stream);
extern "C" hipError_t ntr_launch_selftest_division_hard(int xe0, int de0, unsigned long long* d_counts, hipStream_t stream);
