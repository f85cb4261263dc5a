// sched_kernels.hip -- dispatch-order prediction for the per-ray trace kernel (gfx950).
//
// No counterpart in the reference (its kernels take rays in buffer order).  The launch time of
// trace_bvh_perray is set by where the long-lived blocks start: started late, they
// are the tail of the launch.  Before a closest-hit launch the cost of every 256-ray block is
// therefore PREDICTED and the blocks are dispatched heaviest class first.  The predictor is the
// number of BVH nodes of depth <= 9 (NTR_TRACE_PREDICT_DEPTH) whose boxes one sample ray of the block
// intersects -- exactly what a traversal truncated at that depth would visit, evaluated as a dense
// loop over the top-of-tree box table (no dependent loads).  Prediction only reorders blocks: every
// ray is traced exactly as without it, so hit records do not depend on anything in this file, and
// approximate arithmetic (rcp, no division) is fine here.
//
//   top_table_kernel   one workgroup, once per BVH: breadth-first walk to depth D, every child box
//                      of the nodes above D -> table of <= 2^(D+1)-2 boxes (32 B each)
//   predict_kernel     one lane per block, table boxes as wave-uniform loads: count the boxes hit by the
//                      block's sample ray, class = min(count / 2, CLASSES-1); blocks are appended to
//                      their class's list with one global atomic per (workgroup, class), in block
//                      order inside the workgroup (neighbouring blocks stay neighbours: they share
//                      BVH nodes, and a finer order by cost was measured slower)
//   flatten_kernel     class lists -> one block order, heaviest class first (the order[] the per-ray
//                      kernel already understands); clears the class counters for the next prediction
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trace_kernels.h"

namespace ntr {

// ---- top-of-tree box table -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void top_table_kernel(const float4* __restrict__ nodes, unsigned int nodesBytes, int depth,
                                                        float4* __restrict__ table, unsigned int* __restrict__ tableCount)
{
    // level queues in LDS: a level above `depth` <= NTR_TOP_DEPTH_MAX holds at most 2^(depth-1) inner nodes
    __shared__ int s_q[2][1 << (NTR_TOP_DEPTH_MAX - 1)];
    __shared__ unsigned int s_cnt[2], s_out;
    const int tid = threadIdx.x;
    if (tid == 0) { s_q[0][0] = 0; s_cnt[0] = 1; s_cnt[1] = 0; s_out = 0; }
    __syncthreads();
    for (int d = 0; d < depth; d++) {
        const int cur = d & 1, nxt = cur ^ 1;
        const unsigned int n = s_cnt[cur];
        for (unsigned int e = tid; e < n; e += 256) {
            const unsigned int ofs = (unsigned int)s_q[cur][e];
            if (ofs + 64u > nodesBytes) continue;  // malformed child pointer: ignore (prediction only)
            const float4* nd = nodes + (ofs >> 4);
            const float4 a = nd[0], b = nd[1], c = nd[2];
            const int4 ch = *reinterpret_cast<const int4*>(nd + 3);
            const unsigned int o = atomicAdd(&s_out, 2u);
            table[2 * o + 0] = a;                                  // child 0: lo.x hi.x lo.y hi.y
            table[2 * o + 1] = make_float4(c.x, c.y, 0.0f, 0.0f);  //          lo.z hi.z
            table[2 * o + 2] = b;                                  // child 1
            table[2 * o + 3] = make_float4(c.z, c.w, 0.0f, 0.0f);
            if (d + 1 < depth) {
                if (ch.x >= 0) s_q[nxt][atomicAdd(&s_cnt[nxt], 1u)] = ch.x;
                if (ch.y >= 0) s_q[nxt][atomicAdd(&s_cnt[nxt], 1u)] = ch.y;
            }
        }
        __syncthreads();
        if (tid == 0) s_cnt[cur] = 0;
        __syncthreads();
    }
    if (tid == 0) *tableCount = s_out;
}

// ---- per-block cost class + class lists --------------------------------------------------------------
// One lane per block (its sample ray in registers); the sixteen waves of a workgroup share 64 blocks and split
// the table between them: a wave stages its slice in LDS and reads every box back at a wave-uniform address
// (one LDS broadcast for all 64 lanes).
constexpr int PRED_WAVES = 16;
constexpr int PRED_SAMPLE = 100;        // sample ray inside the block (any fixed lane)
constexpr int PRED_SAMPLE2 = 227;       // second sample, for the coherence estimate

// Coherence of a 256-ray block, from its sample ray and a second ray of the block (in another wave of it).  Bit 0: the two start
// further apart than 1/8 of the scene's extent -- such rays share no deep nodes and their step counts are uncorrelated; bit 1: they start
// together but their directions are more than 60 degrees apart (bounce rays off neighbouring surface points).  The mini-pool cases
// (trace_kernels.hip).  o, d = origin and direction of the block's sample ray; table[0..3] = the root's two child boxes.
__device__ __forceinline__ unsigned int block_incoherence(const float4* __restrict__ rays, int numRays, int block, const float4 o, const float4 d,
                                                          const float4* __restrict__ table)
{
    if (!(o.w < d.w)) return 8u;   // a degenerate sample ray (a missed pixel's secondary ray, Util.hpp:65): the block has no say
    const int r2 = min(block * 256 + PRED_SAMPLE2, numRays - 1);
    const float4 o2 = rays[2 * r2], d2 = rays[2 * r2 + 1];
    const float4 a0 = table[0], a1 = table[1], b0 = table[2], b1 = table[3];
    const float ext = fmaxf(fmaxf(fmaxf(a0.y, b0.y) - fminf(a0.x, b0.x), fmaxf(a0.w, b0.w) - fminf(a0.z, b0.z)),
                            fmaxf(a1.y, b1.y) - fminf(a1.x, b1.x));
    const float dist = fmaxf(fmaxf(fabsf(o2.x - o.x), fabsf(o2.y - o.y)), fabsf(o2.z - o.z));
    if (dist > 0.125f * ext) return 1u;   // (comparisons are false for NaN: coherent)
    const float dot = d.x * d2.x + d.y * d2.y + d.z * d2.z;
    const float l1 = d.x * d.x + d.y * d.y + d.z * d.z, l2 = d2.x * d2.x + d2.y * d2.y + d2.z * d2.z;
    if (!(dot < 0.0f || 4.0f * dot * dot < l1 * l2)) return 0u;   // cos >= 1/2, directions of any length
    // ... and the sample ray is LONG (more than 1/8 of the scene's extent): short rays that point apart -- an AO batch's -- stay in the
    // leaves they start in and end together; long ones -- a diffuse batch's -- wander through different parts of the tree (bit 2)
    const float reach = (d.w - o.w) * (d.w - o.w) * l1;
    return reach > (0.125f * ext) * (0.125f * ext) ? 6u : 2u;
}

__global__ __launch_bounds__(PRED_WAVES * 64) void predict_kernel(const float4* __restrict__ rays, int numRays, int numBlocks,
                                                                  const float4* __restrict__ table,
                                                                  const unsigned int* __restrict__ tableCount,
                                                                  unsigned int* __restrict__ classCount, unsigned int* __restrict__ classList,
                                                                  unsigned int* __restrict__ blockCost /* or null: counts only, no class lists */)
{
    constexpr int PER_MAX = ((2 << NTR_TOP_DEPTH_MAX) + PRED_WAVES - 1) / PRED_WAVES;  // boxes per wave, at most
    __shared__ float2 s_tab[PRED_WAVES][PER_MAX * 3];  // per wave: its slice of the table, (lo, hi) per axis
    __shared__ unsigned int s_cnt[64];
    __shared__ unsigned int s_hist[NTR_SCHED_PRED_CLASSES], s_base[NTR_SCHED_PRED_CLASSES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int block = blockIdx.x * 64 + lane;
    if (tid < 64) s_cnt[tid] = 0;
    if (tid < NTR_SCHED_PRED_CLASSES) s_hist[tid] = 0;
    const int nBoxes = (int)*tableCount;
    const int per = min((nBoxes + PRED_WAVES - 1) / PRED_WAVES, PER_MAX);
    const int b0 = wave * per, n = max(min(b0 + per, nBoxes) - b0, 0);
    for (int i = lane; i < n; i += 64) {  // stage this wave's slice (coalesced), then read it back with uniform addresses
        const float4 xy = table[2 * (b0 + i)], z = table[2 * (b0 + i) + 1];
        s_tab[wave][3 * i + 0] = make_float2(xy.x, xy.y);
        s_tab[wave][3 * i + 1] = make_float2(xy.z, xy.w);
        s_tab[wave][3 * i + 2] = make_float2(z.x, z.y);
    }
    const int r = min(min(block, numBlocks - 1) * 256 + PRED_SAMPLE, numRays - 1);
    const float4 o = rays[2 * r], d = rays[2 * r + 1];  // (origin, tmin), (direction, tmax)
    const float ix = __frcp_rn(d.x), iy = __frcp_rn(d.y), iz = __frcp_rn(d.z);
    const float ax = -o.x * ix, ay = -o.y * iy, az = -o.z * iz;  // (plane - o) * inv = fma(plane, inv, a)
    __syncthreads();
    unsigned int cnt = 0;
    const float2* tb = &s_tab[wave][0];
#pragma unroll 8
    for (int i = 0; i < n; i++) {
        const float2 bx = tb[3 * i], by = tb[3 * i + 1], bz = tb[3 * i + 2];  // same address in every lane: LDS broadcast
        const float x0 = fmaf(bx.x, ix, ax), x1 = fmaf(bx.y, ix, ax);
        const float y0 = fmaf(by.x, iy, ay), y1 = fmaf(by.y, iy, ay);
        const float z0 = fmaf(bz.x, iz, az), z1 = fmaf(bz.y, iz, az);
        const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), o.w));
        const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), d.w));
        cnt += (tn <= tf) ? 1u : 0u;
    }
    if (cnt) atomicAdd(&s_cnt[lane], cnt);
    __syncthreads();
    if (blockCost) {   // cost query (ntr_predict_block_costs): the raw box counts, nothing else
        if (tid < 64 && block < numBlocks) blockCost[block] = s_cnt[lane];
        return;
    }
    if (wave == 0) {
        const unsigned int inc = (block < numBlocks && nBoxes >= 2) ? block_incoherence(rays, numRays, block, o, d, table) : 0u;
        const unsigned long long m1 = __ballot(inc == 1u), m2 = __ballot(inc == 6u), m3 = __ballot(inc == 8u);
        if (lane == 0 && m1) atomicAdd(&classCount[NTR_SCHED_PRED_CLASSES], (unsigned int)__popcll(m1));
        if (lane == 0 && (m2 | m3)) atomicAdd(&classCount[NTR_SCHED_PRED_CLASSES + 1], 4u * (unsigned int)__popcll(m2) + (unsigned int)__popcll(m3));   // the divergence score (pool_k)
    }
    unsigned int cls = 0;
    if (tid < 64 && block < numBlocks) {
        cls = min(s_cnt[lane] >> 1, (unsigned int)(NTR_SCHED_PRED_CLASSES - 1));
        atomicAdd(&s_hist[cls], 1u);
        s_cnt[lane] = cls;
    }
    __syncthreads();
    if (tid < NTR_SCHED_PRED_CLASSES && s_hist[tid]) s_base[tid] = atomicAdd(&classCount[tid], s_hist[tid]);
    __syncthreads();
    if (tid < 64 && block < numBlocks) {
        // rank among the workgroup's earlier blocks of the same class: neighbours stay in buffer order
        unsigned int rank = 0;
        for (int j = 0; j < lane; j++) rank += (s_cnt[j] == cls) ? 1u : 0u;
        classList[(size_t)cls * numBlocks + s_base[cls] + rank] = (unsigned int)block;
    }
}

// order[e] = e-th block when the class lists are concatenated from the heaviest class down.  ONE workgroup, which
// therefore can also clear the class counters once every thread has read them: the counters are zero whenever no
// prediction is in flight, without a memset in the stream (hipMemsetAsync nodes were observed not to re-execute
// when a captured HIP graph is replayed; kernels do).
constexpr int FLATTEN_THREADS = 1024;

// Rays a wave of the mini-pool kernel owns / 64: scattered origins in at least half of the batch's blocks -> poolKWide (the host's choice
// by tree and batch size, ntr_api.cpp minipool_wide), else 1.  Measured (profiles/r03_minipool_matrix.txt): 2^21 box rays +29-33 % on the
// 2.8 M / 10 M-triangle LBVHs with K = 4, +9-12 % on the 262 k / 331 k SAH trees with K = 2; camera rays lose with any K > 1 (-15 % to
// -50 %).  Blocks whose rays start together but point apart (bounce rays: 44 % of the blocks of a diffuse batch) are counted and reported,
// not acted on: K = 2 for such batches measured between 0 and +4 %, inside the run-to-run noise.
// Round 6: the word also carries NTR_BATCH_DIVERGENT (bit 16) when a quarter of the blocks hold LONG rays that start together and point
// apart (a diffuse batch: 44 %).  The mini-pool depth ignores it, as before; the choice between the per-ray body and the persistent
// dynamic-fetch body (trace_kernels.hip, TraceParams::routeSkip) does not: those batches are 20-35 % faster with single-lane refills and
// ray splitting in the drain phase.
// The divergence score counts 4 for a block of such rays and 1 for a block whose sample ray is degenerate (a missed pixel's); blocks whose
// rays start apart weigh 4 as well.  Reaching the number of blocks means: a quarter of the blocks that hold LIVE rays are incoherent one
// way or the other (a hairball frame hits in 30 % of its pixels, and its first and last batches hold a few thousand live rays whose
// chains are as long as any: 0.88 ms by the per-ray body, 0.53 ms with single-lane refills and ray splitting).
__device__ __forceinline__ unsigned int pool_k(unsigned int originApart, unsigned int divergenceScore, int numBlocks, int poolKWide)
{
    const unsigned int k = (originApart > 0u && 2u * originApart >= (unsigned int)numBlocks) ? (unsigned int)poolKWide : 1u;
    return k | ((numBlocks > 0 && 4u * originApart + divergenceScore >= (unsigned int)numBlocks) ? (unsigned int)NTR_BATCH_DIVERGENT : 0u);
}
__global__ __launch_bounds__(FLATTEN_THREADS) void flatten_kernel(unsigned int* __restrict__ classCount, const unsigned int* __restrict__ classList,
                                                                  int numBlocks, unsigned int* __restrict__ order,
                                                                  unsigned int* __restrict__ poolKCopy, int poolKWide)
{
    __shared__ unsigned int s_end[NTR_SCHED_PRED_CLASSES];  // s_end[k] = entries in the k+1 heaviest classes
    const int tid = threadIdx.x;
    if (tid == 64) {
        // most blocks incoherent -> the mini-pool kernel's waves own poolKWide x 64 rays; else one ray per lane (the word outlives this
        // prediction: the trace launch behind it reads it, the next prediction on this scratch is ordered after that launch)
        const unsigned int k = pool_k(classCount[NTR_SCHED_PRED_CLASSES], classCount[NTR_SCHED_PRED_CLASSES + 1], numBlocks, poolKWide);
        classCount[NTR_SCHED_PRED_CLASSES] = 0;
        classCount[NTR_SCHED_PRED_CLASSES + 1] = 0;
        classCount[NTR_SCHED_PRED_CLASSES + 2] = k;
        if (poolKCopy) *poolKCopy = k;
    }
    if (tid < 64) {
        static_assert(NTR_SCHED_PRED_CLASSES == 64, "one class per lane");
        unsigned int incl = classCount[NTR_SCHED_PRED_CLASSES - 1 - tid];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int u = (unsigned int)__shfl_up((int)incl, off);
            if (tid >= off) incl += u;
        }
        s_end[tid] = incl;
        classCount[NTR_SCHED_PRED_CLASSES - 1 - tid] = 0;  // consumed: ready for the next prediction
    }
    __syncthreads();
    for (int e = tid; e < numBlocks; e += FLATTEN_THREADS) {
        int k = 0;  // first k with e < s_end[k]
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (k + step <= NTR_SCHED_PRED_CLASSES - 1 && s_end[k + step - 1] <= (unsigned int)e) k += step;
        const unsigned int begin = k ? s_end[k - 1] : 0u;
        const int cls = NTR_SCHED_PRED_CLASSES - 1 - k;
        order[e] = classList[(size_t)cls * numBlocks + ((unsigned int)e - begin)];
    }
}

// Coherence query (ntr_predict_batch_coherence): out[0] / out[1] += blocks whose sample rays start apart / start together and point apart;
// the finish step turns the counts into the pool K (out[2]) exactly as flatten_kernel does.
__global__ __launch_bounds__(256) void coherence_kernel(const float4* __restrict__ rays, int numRays, int numBlocks, const float4* __restrict__ table,
                                                        const unsigned int* __restrict__ tableCount, unsigned int* __restrict__ out)
{
    const int block = blockIdx.x * 256 + threadIdx.x;
    unsigned int inc = 0;
    if (block < numBlocks && *tableCount >= 2u) {
        const int r = min(block * 256 + PRED_SAMPLE, numRays - 1);
        inc = block_incoherence(rays, numRays, block, rays[2 * r], rays[2 * r + 1], table);
    }
    const unsigned long long m1 = __ballot(inc == 1u), m2 = __ballot(inc == 6u), m3 = __ballot(inc == 8u);
    if ((threadIdx.x & 63) == 0 && m1) atomicAdd(&out[0], (unsigned int)__popcll(m1));
    if ((threadIdx.x & 63) == 0 && (m2 | m3)) atomicAdd(&out[1], 4u * (unsigned int)__popcll(m2) + (unsigned int)__popcll(m3));
}
__global__ void coherence_finish_kernel(unsigned int* __restrict__ out, int numBlocks, int poolKWide)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = pool_k(out[0], out[1], numBlocks, poolKWide);
}

// Clears `words` 32-bit words (a kernel, not hipMemsetAsync: see flatten_kernel).
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned int* __restrict__ p, int words)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) p[i] = 0;
}

// Fetches and clears the sticky status word in ONE device-side step (an overflow bit set by a launch on another stream between a
// copy and a separate clear would be lost): out[0] = atomicExch(status, 0).
__global__ void status_exchange_kernel(unsigned int* __restrict__ status, unsigned int* __restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = atomicExch(status, 0u);
}

// ---- leaf depths: a cost predictor for secondary batches without history ----------------------------------------------------------
// Short secondary rays (AO) mostly pay for descending from the root to where they start, so the depth in the tree of the leaf a pixel's
// primary ray hit predicts what the pixel's secondary rays cost: Spearman 0.59-0.91 per 256-ray block on the bench frame's AO batches,
// and dispatching the blocks deepest class first recovers what the order LEARNED from a previous launch gives (-9.5 % on cold AO batches,
// scripts/studies/static_order_study.py) -- without a previous launch.
// One launch per tree level: the frontier of inner nodes at depth d -> their inner children (next frontier) and, for leaf children, depth
// d + 1 recorded for every triangle of the leaf.  Malformed references are skipped (this is a hint, not a validator).
__global__ __launch_bounds__(256) void leaf_depth_level_kernel(const int4* __restrict__ nodes, unsigned int nodesBytes, const uint4* __restrict__ woop,
                                                               unsigned int woopVec4, const int* __restrict__ triIndex, int numTris,
                                                               const unsigned int* __restrict__ qin, const unsigned int* __restrict__ nIn,
                                                               unsigned int* __restrict__ qout, unsigned int* __restrict__ nOut, unsigned int capacity,
                                                               int depth, int* __restrict__ depthByTri)
{
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i >= *nIn) return;
    const unsigned int ofs = qin[i];
    if ((unsigned long long)ofs + 64ull > nodesBytes) return;
    const int4 ch = nodes[(ofs >> 4) + 3];
    const int c[2] = {ch.x, ch.y};
    for (int k = 0; k < 2; k++) {
        if (c[k] >= 0) {
            if ((c[k] & 63) == 0 && (unsigned int)c[k] != 0x76543210u) {
                const unsigned int slot = atomicAdd(nOut, 1u);
                if (slot < capacity) qout[slot] = (unsigned int)c[k];
            }
        } else {
            for (unsigned int a = (unsigned int)~c[k], guard = 0; a < woopVec4 && guard < 4096u; a += 3, guard++) {
                if (woop[a].x == 0x80000000u) break;
                const int t = triIndex[a];
                if (t >= 0 && t < numTris) depthByTri[t] = depth + 1;
            }
        }
    }
}

// Predicted cost of the 256-ray blocks of a secondary batch made of numSamples rays per input ray (primary hit): the deepest leaf among the
// block's input rays.  blockCost must be zero on entry.
__global__ __launch_bounds__(256) void secondary_block_cost_kernel(const int4* __restrict__ inResults, int first, int count, int numSamples,
                                                                   const int* __restrict__ depthByTri, int numTris, unsigned int* __restrict__ blockCost)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int id = inResults[first + i].x;
    const unsigned int d = (id >= 0 && id < numTris) ? (unsigned int)depthByTri[id] : 0u;
    if (d == 0u) return;
    const long long o0 = (long long)i * numSamples, o1 = o0 + numSamples - 1;
    for (long long b = o0 >> 8; b <= (o1 >> 8); b++) atomicMax(&blockCost[b], d);
}

}  // namespace ntr

extern "C" hipError_t ntr_launch_leaf_depth_level(const void* d_nodes, unsigned int nodesBytes, const void* d_woop, unsigned int woopVec4, const int* d_triIndex,
                                                  int numTris, const unsigned int* d_qin, const unsigned int* d_nIn, unsigned int* d_qout, unsigned int* d_nOut,
                                                  unsigned int capacity, unsigned int gridThreads, int depth, int* d_depthByTri, hipStream_t stream)
{
    if (gridThreads == 0) return hipSuccess;
    hipLaunchKernelGGL(ntr::leaf_depth_level_kernel, dim3((gridThreads + 255u) / 256u), dim3(256), 0, stream, (const int4*)d_nodes, nodesBytes,
                       (const uint4*)d_woop, woopVec4, d_triIndex, numTris, d_qin, d_nIn, d_qout, d_nOut, capacity, depth, d_depthByTri);
    return hipGetLastError();
}

extern "C" hipError_t ntr_launch_secondary_block_costs(const void* d_inResults, int first, int count, int numSamples, const int* d_depthByTri, int numTris,
                                                       unsigned int* d_blockCost, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(ntr::secondary_block_cost_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, (const int4*)d_inResults, first, count, numSamples,
                       d_depthByTri, numTris, d_blockCost);
    return hipGetLastError();
}

extern "C" hipError_t ntr_launch_status_exchange(unsigned int* d_status, unsigned int* d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(ntr::status_exchange_kernel, dim3(1), dim3(64), 0, stream, d_status, d_out);
    return hipGetLastError();
}

extern "C" hipError_t ntr_launch_top_table(const void* d_nodes, unsigned int nodesBytes, int depth, void* d_table,
                                           unsigned int* d_tableCount, hipStream_t stream)
{
    if (depth < 1) depth = 1;
    if (depth > NTR_TOP_DEPTH_MAX) depth = NTR_TOP_DEPTH_MAX;
    hipLaunchKernelGGL(ntr::top_table_kernel, dim3(1), dim3(256), 0, stream, (const float4*)d_nodes, nodesBytes, depth, (float4*)d_table,
                       d_tableCount);
    return hipGetLastError();
}

// d_classCount: NTR_SCHED_PRED_WORDS words; the first NTR_SCHED_PRED_CLASSES + 2 must be zero on entry and are zero again when the
// launches have run, the last one receives the pool K of the mini-pool kernel (also stored to d_poolKCopy when given).
extern "C" hipError_t ntr_launch_predict(const void* d_rays, int numRays, int numBlocks, const void* d_table,
                                         const unsigned int* d_tableCount, unsigned int* d_classCount, unsigned int* d_classList,
                                         unsigned int* d_order, unsigned int* d_poolKCopy, int poolKWide, hipStream_t stream)
{
    const int grid = (numBlocks + 63) / 64;
    hipLaunchKernelGGL(ntr::predict_kernel, dim3(grid), dim3(ntr::PRED_WAVES * 64), 0, stream, (const float4*)d_rays, numRays, numBlocks,
                       (const float4*)d_table, d_tableCount, d_classCount, d_classList, (unsigned int*)nullptr);
    hipLaunchKernelGGL(ntr::flatten_kernel, dim3(1), dim3(ntr::FLATTEN_THREADS), 0, stream, d_classCount, (const unsigned int*)d_classList,
                       numBlocks, d_order, d_poolKCopy, poolKWide);
    return hipGetLastError();
}

extern "C" hipError_t ntr_launch_predict_costs(const void* d_rays, int numRays, int numBlocks, const void* d_table,
                                               const unsigned int* d_tableCount, unsigned int* d_blockCost, hipStream_t stream)
{
    const int grid = (numBlocks + 63) / 64;
    hipLaunchKernelGGL(ntr::predict_kernel, dim3(grid), dim3(ntr::PRED_WAVES * 64), 0, stream, (const float4*)d_rays, numRays, numBlocks,
                       (const float4*)d_table, d_tableCount, (unsigned int*)nullptr, (unsigned int*)nullptr, d_blockCost);
    return hipGetLastError();
}

extern "C" hipError_t ntr_launch_coherence(const void* d_rays, int numRays, int numBlocks, const void* d_table, const unsigned int* d_tableCount,
                                           unsigned int* d_out, int poolKWide, hipStream_t stream)
{
    hipLaunchKernelGGL(ntr::zero_words_kernel, dim3(1), dim3(256), 0, stream, d_out, 3);
    if (numBlocks > 0)
        hipLaunchKernelGGL(ntr::coherence_kernel, dim3((numBlocks + 255) / 256), dim3(256), 0, stream, (const float4*)d_rays, numRays, numBlocks,
                           (const float4*)d_table, d_tableCount, d_out);
    hipLaunchKernelGGL(ntr::coherence_finish_kernel, dim3(1), dim3(64), 0, stream, d_out, numBlocks, poolKWide);
    return hipGetLastError();
}

extern "C" hipError_t ntr_launch_zero_words(void* d_ptr, int words, hipStream_t stream)
{
    if (words <= 0) return hipSuccess;
    int grid = (words + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(ntr::zero_words_kernel, dim3(grid), dim3(256), 0, stream, (unsigned int*)d_ptr, words);
    return hipGetLastError();
}

namespace ntr {

// ---------------------------------------------------------------------------------
// Scheduling feedback (ntr_trace_bvh_hinted): turns the per-block costs one launch recorded into
// the block order of the next launch of the same logical batch -- heaviest cost class first, so
// that the long-lived waves start early instead of forming the tail of the launch.  Blocks are
// only CLASSIFIED (NTR_SCHED_CLASSES linear classes of the maximum cost) and keep their original
// order inside a class: neighbouring blocks trace neighbouring rays, and a full sort by cost was
// measured slower than the coarse one because it gives that locality up (scripts/studies/order_experiment.py).
// One workgroup; stable counting sort with a per-thread segment of the block range.
// ---------------------------------------------------------------------------------
constexpr int SCHED_THREADS = 256;   // 64 classes x 256 threads x 4 B = 64 KB of static LDS (+ 1 KB of wave totals)
constexpr int SCHED_MAX_CLASSES = 64;

// MAXC: the classes the unrolled scans are written for (32: the default NTR_SCHED_CLASSES -- half the shuffles and registers of 64)
template <int MAXC>
__global__ __launch_bounds__(SCHED_THREADS) void sched_order_kernel(const unsigned int* __restrict__ cost, int numBlocks, int classes,
                                                                    unsigned int* __restrict__ order)
{
    __shared__ unsigned int s_cnt[MAXC][SCHED_THREADS];
    __shared__ unsigned int s_tot[MAXC][SCHED_THREADS / 64];
    __shared__ unsigned int s_red[SCHED_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = (numBlocks + SCHED_THREADS - 1) / SCHED_THREADS;
    const int b0 = min(tid * seg, numBlocks), b1 = min(b0 + seg, numBlocks);

    unsigned int mx = 0;
    for (int i = tid; i < numBlocks; i += SCHED_THREADS) mx = max(mx, cost[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, off));
    if (lane == 0) s_red[wave] = mx;
    for (int c = 0; c < classes; c++) s_cnt[c][tid] = 0;
    __syncthreads();
    mx = 0;
    for (int w = 0; w < SCHED_THREADS / 64; w++) mx = max(mx, s_red[w]);
    // class 0 = heaviest.  Any monotone map of the cost onto [0, classes) will do -- the order only has to be a permutation, and both
    // passes below use the same map -- so a float multiply stands in for the 64-bit division (a hundred instructions per block).
    const float toClass = (float)classes / ((float)mx + 1.0f);
    auto cls = [&](unsigned int c) { return (classes - 1) - min((int)((float)c * toClass), classes - 1); };

    for (int i = b0; i < b1; i++) s_cnt[cls(cost[i])][tid]++;
    __syncthreads();
    // Exclusive scan over (class major, thread minor).  Every lane scans all its classes' counts across the wave at once (independent
    // shuffle chains), the waves exchange their totals once: three barriers in all instead of two per class.
    unsigned int v[MAXC], incl[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        v[c] = c < classes ? s_cnt[c][tid] : 0u;
        incl[c] = v[c];
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            const unsigned int u = (unsigned int)__shfl_up((int)incl[c], off);
            if (lane >= off) incl[c] += u;
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < MAXC; c++) s_tot[c][wave] = incl[c];
    }
    __syncthreads();
    unsigned int running = 0;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        if (c < classes) {
            unsigned int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < SCHED_THREADS / 64; w++) {
                const unsigned int t = s_tot[c][w];
                if (w < wave) before += t;
                total += t;
            }
            s_cnt[c][tid] = running + before + incl[c] - v[c];
            running += total;
        }
    }
    // (every thread reads back only its own column of s_cnt: no barrier needed)
    for (int i = b0; i < b1; i++) {
        const int c = cls(cost[i]);
        order[s_cnt[c][tid]++] = (unsigned int)i;
    }
}

}  // namespace ntr

extern "C" hipError_t ntr_launch_sched_order(const unsigned int* d_cost, int numBlocks, int classes, unsigned int* d_order,
                                             hipStream_t stream)
{
    if (classes < 1) classes = 1;
    if (classes > ntr::SCHED_MAX_CLASSES) classes = ntr::SCHED_MAX_CLASSES;
    if (classes <= 32)
        hipLaunchKernelGGL(ntr::sched_order_kernel<32>, dim3(1), dim3(ntr::SCHED_THREADS), 0, stream, d_cost, numBlocks, classes, d_order);
    else
        hipLaunchKernelGGL(ntr::sched_order_kernel<ntr::SCHED_MAX_CLASSES>, dim3(1), dim3(ntr::SCHED_THREADS), 0, stream, d_cost, numBlocks, classes, d_order);
    return hipGetLastError();
}
