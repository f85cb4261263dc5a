// lbvh_kernels.hip -- on-device LBVH builder for gfx950.
//
// Rebuilds the pipeline of HLBVHBuilder::buildLBVH (src/rt/bvh/HLBVH/HLBVHBuilder.cpp:451-593):
//   calcMorton      emitTreeKernel.cu:655-691   -> lbvh_morton_kernel
//   radixSortCuda   radixSort.cu:22-50 (Thrust) -> hand-written LSD radix sort, 4 x 8 bits:
//                                                  per-tile LDS histograms, one scan, and a stable
//                                                  scatter ranked with wave64 ballots (match-any)
//   calcWoopKernel  emitTreeKernel.cu:574-645   -> lbvh_woop_kernel
//   emitTreeKernel  emitTreeKernel.cu:233-381   -> lbvh_emit_kernel, one launch per level with the
//   + createLeaf    :170-231                       queue counts kept on the device (the reference
//                                                  reads g_outQueuePtr back to the host every level,
//                                                  HLBVHBuilder.cpp:347)
//   calcAABB        emitTreeKernel.cu:417-562   -> lbvh_refit_kernel, deepest level first
//
// The tree is the reference's tree: same split rule (highest differing Morton bit at or below the
// level's bit, median when none), same leaf rule (count <= leafSize, or the level's bit is 0), same
// Woop rows and boxes (strict IEEE evaluation of the reference expressions; the reference builds
// these kernels with -use_fast_math so its own bits are toolchain dependent).  Node numbering and
// leaf placement depend on atomic order, as in the reference (emitTreeKernel.cu:176,303); parity is
// checked on the canonical (numbering-independent) form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <string.h>

#include "ntr_internal.h"
#include "radix_sort.h"

namespace ntr {

struct LbvhState {
    unsigned int lvlNodes[34];   // nodes per level (lvlNodes[0] = 1)
    unsigned int lvlStart[34];   // first node index of each level
    unsigned long long leafPtr;  // (triCount << 32) | leafCount, like g_leafsPtr
    unsigned int overflow;
    unsigned int pad;
};

// ---- Morton codes ------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int spread10(unsigned int n)  // emitTreeKernel.cu:647-653
{
    n &= 0x3ff;
    n = (n ^ (n << 16)) & 0xff0000ff;
    n = (n ^ (n << 8)) & 0x0300f00f;
    n = (n ^ (n << 4)) & 0x030c30c3;
    return (n ^ (n << 2)) & 0x09249249;
}

struct F3 { float x, y, z; };

__global__ __launch_bounds__(256) void lbvh_morton_kernel(int n, const int* __restrict__ tri, const float* __restrict__ pos,
                                                          F3 lo, F3 step, unsigned int* __restrict__ keys,
                                                          int* __restrict__ idx)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    const float l[3] = {lo.x, lo.y, lo.z}, s[3] = {step.x, step.y, step.z};
    int cell[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float a = pos[3 * i0 + k], b = pos[3 * i1 + k], c = pos[3 * i2 + k];
        const float mn = fminf(a, fminf(b, c)), mx = fmaxf(a, fmaxf(b, c));
        const float mid = mn + (mx - mn) / 2.0f;
        const int v = (int)floorf((mid - l[k]) / s[k]);
        cell[k] = min(max(v, 0), 1023);
    }
    keys[t] = spread10(cell[0]) | (spread10(cell[1]) << 1) | (spread10(cell[2]) << 2);
    idx[t] = t;
}

// ---- Woop rows (emitTreeKernel.cu:574-635) ---------------------------------------------------------
__global__ __launch_bounds__(256) void lbvh_woop_kernel(int n, const int* __restrict__ tri, const float* __restrict__ pos,
                                                        float4* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    const float v0x = pos[3 * i0], v0y = pos[3 * i0 + 1], v0z = pos[3 * i0 + 2];
    const float v1x = pos[3 * i1], v1y = pos[3 * i1 + 1], v1z = pos[3 * i1 + 2];
    const float v2x = pos[3 * i2], v2y = pos[3 * i2 + 1], v2z = pos[3 * i2 + 2];
    const float c0x = v0x - v2x, c0y = v0y - v2y, c0z = v0z - v2z;
    const float c1x = v1x - v2x, c1y = v1y - v2y, c1z = v1z - v2z;
    const float c2x = c0y * c1z - c0z * c1y, c2y = c0z * c1x - c0x * c1z, c2z = c0x * c1y - c0y * c1x;
    const float den = c0x * (c2z * c1y - c1z * c2y) - c0y * (c2z * c1x - c1z * c2x) + c0z * (c2y * c1x - c1y * c2x);
    const float det = (float)(1.0 / (double)den);  // `1.0/(float)` is a binary64 divide in the reference (:589)

    const float i0x = (c2z * c1y - c1z * c2y) * det, i0y = -(c2z * c1x - c1z * c2x) * det, i0z = (c2y * c1x - c1y * c2x) * det;
    const float i1x = -(c2z * c0y - c0z * c2y) * det, i1y = (c2z * c0x - c0z * c2x) * det, i1z = -(c2y * c0x - c0y * c2x) * det;
    const float i2x = (c1z * c0y - c0z * c1y) * det, i2y = -(c1z * c0x - c0z * c1x) * det, i2z = (c1y * c0x - c0y * c1x) * det;
    const float o0w = -((-i2x) * v2x + (-i2y) * v2y + (-i2z) * v2z);
    const float o1w = (-i0x) * v2x + (-i0y) * v2y + (-i0z) * v2z;
    const float o2w = (-i1x) * v2x + (-i1y) * v2y + (-i1z) * v2z;
    float o0x = i2x;
    if (o0x == 0.0f) o0x = 0.0f;  // -0 would alias the leaf terminator
    out[3 * t + 0] = make_float4(o0x, i2y, i2z, o0w);
    out[3 * t + 1] = make_float4(i0x, i0y, i0z, o1w);
    out[3 * t + 2] = make_float4(i1x, i1y, i1z, o2w);
}

// ---- tree emission, one level per launch (emitTreeKernel.cu:233-381) ------------------------------
__device__ __forceinline__ int create_leaf(LbvhState* st, const float4* __restrict__ inWoop, const int* __restrict__ triSorted,
                                           float4* __restrict__ outWoop, int* __restrict__ outIdx, int start, int end)
{
    const unsigned int numTris = end - start;
    const unsigned long long add = ((unsigned long long)numTris << 32) + 1ull;
    const unsigned long long p = atomicAdd(&st->leafPtr, add);
    const unsigned int numLeafs = (unsigned int)(p & 0xFFFFFFFFull), allTris = (unsigned int)(p >> 32);
    const int out = allTris * 3 + numLeafs;  // float4 index; one extra float4 per leaf for the terminator
    for (unsigned int i = 0; i < numTris; i++) {
        const int t = triSorted[start + i];
        outWoop[out + 3 * i + 0] = inWoop[3 * t + 0];
        outWoop[out + 3 * i + 1] = inWoop[3 * t + 1];
        outWoop[out + 3 * i + 2] = inWoop[3 * t + 2];
        outIdx[out + 3 * i + 0] = t;
        outIdx[out + 3 * i + 1] = 0;
        outIdx[out + 3 * i + 2] = 0;
    }
    const float nz = __uint_as_float(0x80000000u);
    outWoop[out + 3 * numTris] = make_float4(nz, nz, nz, nz);
    outIdx[out + 3 * numTris] = 0;
    return ~out;
}

__global__ __launch_bounds__(256) void lbvh_emit_kernel(int lvl, int levelBit, int leafSize, LbvhState* __restrict__ st,
                                                        const unsigned int* __restrict__ keys, const int* __restrict__ triSorted,
                                                        const float4* __restrict__ inWoop, const int* __restrict__ qIn,
                                                        int* __restrict__ qOut, int* __restrict__ nodes, unsigned int nodeCapacity,
                                                        float4* __restrict__ outWoop, int* __restrict__ outIdx)
{
    const unsigned int inCount = st->lvlNodes[lvl];
    const unsigned int inOfs = st->lvlStart[lvl] + inCount;  // index of the first node of the next level
    if (blockIdx.x == 0 && threadIdx.x == 0) st->lvlStart[lvl + 1] = inOfs;
    const int lane = threadIdx.x & 63;
    const unsigned int stride = gridDim.x * blockDim.x;
    // all lanes of a wave run the same number of iterations (the wave-level scan needs them)
    const unsigned int rounds = (inCount + stride - 1) / stride;
    for (unsigned int it = 0; it < rounds; it++) {
        const unsigned int e = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = e < inCount;
        int nIdx = 0, nStart = 0, nEnd = 0, split = 0, level = levelBit;
        bool leaf0 = false, leaf1 = false;
        if (valid) {
            nIdx = qIn[3 * e]; nStart = qIn[3 * e + 1]; nEnd = qIn[3 * e + 2];
            const unsigned int kFirst = keys[nStart], kLast = keys[nEnd - 1];
            while (level >= 0 && (((kFirst >> level) & 1) == ((kLast >> level) & 1))) level--;
            if (level >= 0) {  // split where the bit flips (binary search, :263-280)
                const unsigned int startBit = (kFirst >> level) & 1;
                int a = nStart, b = nEnd;
                for (;;) {
                    split = (a + b) >> 1;
                    const unsigned int splitBit = (keys[split] >> level) & 1;
                    if (((keys[split - 1] >> level) & 1) != splitBit) break;
                    if (splitBit == startBit) a = split; else b = split;
                }
            } else {
                split = (nStart + nEnd) >> 1;  // identical keys: median (:282)
            }
            leaf0 = (split - nStart) <= leafSize || levelBit == 0;
            leaf1 = (nEnd - split) <= leafSize || levelBit == 0;
        }
        // queue slots for the inner children: wave prefix sum + one atomic per wave (:296-303)
        const int mine = valid ? ((leaf0 ? 0 : 1) + (leaf1 ? 0 : 1)) : 0;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const int total = __shfl(incl, 63);
        unsigned int waveBase = 0;
        if (lane == 63 && total > 0) waveBase = atomicAdd(&st->lvlNodes[lvl + 1], (unsigned int)total);
        waveBase = __shfl(waveBase, 63);
        if (!valid) continue;
        unsigned int outOff = waveBase + (incl - mine);
        unsigned int outIdxNode = inOfs + outOff;
        if (outIdxNode + 2 > nodeCapacity) { atomicOr(&st->overflow, 1u); continue; }

        int c0, c1;
        int* nd = nodes + (size_t)nIdx * 16;
        if (leaf0) {
            c0 = create_leaf(st, inWoop, triSorted, outWoop, outIdx, nStart, split);
            nd[0] = nStart; nd[1] = split;  // consumed by the refit pass
        } else {
            qOut[3 * outOff] = outIdxNode; qOut[3 * outOff + 1] = nStart; qOut[3 * outOff + 2] = split;
            c0 = outIdxNode * 64;
            outOff++; outIdxNode++;
        }
        if (leaf1) {
            c1 = create_leaf(st, inWoop, triSorted, outWoop, outIdx, split, nEnd);
            nd[4] = split; nd[5] = nEnd;
        } else {
            qOut[3 * outOff] = outIdxNode; qOut[3 * outOff + 1] = split; qOut[3 * outOff + 2] = nEnd;
            c1 = outIdxNode * 64;
        }
        nd[12] = c0; nd[13] = c1; nd[14] = level % 3; nd[15] = 0;
    }
}

// ---- bottom-up refit, one level per launch (emitTreeKernel.cu:417-562) ---------------------------
__device__ __forceinline__ void calc_leaf(const int* __restrict__ tri, const float* __restrict__ pos,
                                          const int* __restrict__ triSorted, int start, int end, float eps, float (&lo)[3], float (&hi)[3])
{
    for (int i = start; i < end; i++) {
        const int t = triSorted[i];
        const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float a = pos[3 * i0 + k], b = pos[3 * i1 + k], c = pos[3 * i2 + k];
            lo[k] = fminf(lo[k], fminf(a, fminf(b, c)) - eps);
            hi[k] = fmaxf(hi[k], fmaxf(a, fmaxf(b, c)) + eps);
        }
    }
}

__global__ __launch_bounds__(256) void lbvh_refit_kernel(int lvl, float eps, const LbvhState* __restrict__ st,
                                                         const int* __restrict__ tri, const float* __restrict__ pos,
                                                         const int* __restrict__ triSorted, int* __restrict__ nodes)
{
    const unsigned int cnt = st->lvlNodes[lvl], start = st->lvlStart[lvl];
    for (unsigned int q = blockIdx.x * blockDim.x + threadIdx.x; q < cnt; q += gridDim.x * blockDim.x) {
        int* ni = nodes + (size_t)(start + q) * 16;
        float* nf = reinterpret_cast<float*>(ni);
        const int ch[2] = {ni[12], ni[13]};
        float box[2][6];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (ch[k] < 0) {
                float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                calc_leaf(tri, pos, triSorted, ni[4 * k], ni[4 * k + 1], eps, lo, hi);
                box[k][0] = lo[0]; box[k][1] = hi[0]; box[k][2] = lo[1]; box[k][3] = hi[1]; box[k][4] = lo[2]; box[k][5] = hi[2];
            } else {
                const float4* cn = reinterpret_cast<const float4*>(nodes + (size_t)(ch[k] >> 6) * 16);
                const float4 a = cn[0], b = cn[1], c = cn[2];
                box[k][0] = fminf(a.x, b.x); box[k][1] = fmaxf(a.y, b.y);
                box[k][2] = fminf(a.z, b.z); box[k][3] = fmaxf(a.w, b.w);
                box[k][4] = fminf(c.x, c.z); box[k][5] = fmaxf(c.y, c.w);
            }
        }
        reinterpret_cast<float4*>(nf)[0] = make_float4(box[0][0], box[0][1], box[0][2], box[0][3]);
        reinterpret_cast<float4*>(nf)[1] = make_float4(box[1][0], box[1][1], box[1][2], box[1][3]);
        reinterpret_cast<float4*>(nf)[2] = make_float4(box[0][4], box[0][5], box[1][4], box[1][5]);
    }
}

}  // namespace ntr

using namespace ntr;

namespace {
struct Timer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t s;
    explicit Timer(hipStream_t st) : s(st) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
    ~Timer() { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
    void start() { (void)hipEventRecord(e0, s); }
    float stop_ms() { (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); return ms; }
};
struct DevMem {  // frees on scope exit
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
};
}  // namespace

extern "C" {

int ntr_lbvh_capacity(int32_t numTris, int64_t* nodesBytes, int64_t* triWoopBytes, int64_t* triIndexBytes)
{
    if (numTris < 1) return set_error(NTR_ERR_INVALID, "ntr_lbvh_capacity: numTris < 1");
    // HLBVHBuilder::initMemory(q0, q1, min(2, leafSize)) sizes the node array for n nodes
    // (HLBVHBuilder.cpp:561, 772-784); Woop / index take (3+1) entries per triangle (:532-538).
    if (nodesBytes) *nodesBytes = ((int64_t)numTris + 2) * 64;
    if (triWoopBytes) *triWoopBytes = ((int64_t)numTris * 4 + 4) * 16;
    if (triIndexBytes) *triIndexBytes = ((int64_t)numTris * 4 + 4) * 4;
    return NTR_OK;
}

int ntr_lbvh_build(int32_t numTris, const int32_t* d_triVtxIndex, int32_t numVerts, const float* d_vtxPos,
                   const float sceneMin[3], const float sceneMax[3], int32_t leafSize, float epsilon,
                   void* d_nodes, int64_t nodesCapacity, void* d_triWoop, int64_t triWoopCapacity,
                   int32_t* d_triIndex, int64_t triIndexCapacity, NtrLbvhResult* result, void* stream)
{
    if (!result) return set_error(NTR_ERR_INVALID, "ntr_lbvh_build: null result");
    memset(result, 0, sizeof(*result));
    if (numTris < 1 || numVerts < 1 || leafSize < 1 || !d_triVtxIndex || !d_vtxPos || !sceneMin || !sceneMax)
        return set_error(NTR_ERR_INVALID, "ntr_lbvh_build: bad geometry arguments");
    int64_t needN, needW, needI;
    ntr_lbvh_capacity(numTris, &needN, &needW, &needI);
    if (!d_nodes || !d_triWoop || !d_triIndex || nodesCapacity < needN || triWoopCapacity < needW || triIndexCapacity < needI)
        return set_error(NTR_ERR_INVALID, "ntr_lbvh_build: output buffers smaller than ntr_lbvh_capacity()");
    hipStream_t s = (hipStream_t)stream;
    const int n = numTris;
    const int nb = (n + SORT_TILE - 1) / SORT_TILE;

    DevMem keysA, keysB, idxA, idxB, inWoop, q0, q1, hist, state;
    NTR_HIP(hipMalloc(&keysA.p, (size_t)n * 4));
    NTR_HIP(hipMalloc(&keysB.p, (size_t)n * 4));
    NTR_HIP(hipMalloc(&idxA.p, (size_t)n * 4));
    NTR_HIP(hipMalloc(&idxB.p, (size_t)n * 4));
    NTR_HIP(hipMalloc(&inWoop.p, (size_t)n * 48));
    NTR_HIP(hipMalloc(&q0.p, ((size_t)n + 2) * 12));
    NTR_HIP(hipMalloc(&q1.p, ((size_t)n + 2) * 12));
    NTR_HIP(hipMalloc(&hist.p, ((size_t)nb * 256 + 256) * 4));
    NTR_HIP(hipMalloc(&state.p, sizeof(LbvhState)));

    Timer tAll(s), tPhase(s);
    tAll.start();

    // L1: Morton codes (step = (max - min) / 1024 on the host, HLBVHBuilder.cpp:76-81)
    F3 lo = {sceneMin[0], sceneMin[1], sceneMin[2]};
    F3 step = {(sceneMax[0] - sceneMin[0]) / 1024.0f, (sceneMax[1] - sceneMin[1]) / 1024.0f, (sceneMax[2] - sceneMin[2]) / 1024.0f};
    tPhase.start();
    hipLaunchKernelGGL(lbvh_morton_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_triVtxIndex, d_vtxPos, lo, step,
                       (unsigned int*)keysA.p, (int*)idxA.p);
    NTR_HIP(hipGetLastError());
    result->mortonMs = tPhase.stop_ms();

    // L2: stable radix sort by key, 4 passes of 8 bits (the 30-bit code fits)
    tPhase.start();
    unsigned int *kIn = (unsigned int*)keysA.p, *kOut = (unsigned int*)keysB.p;
    int *vIn = (int*)idxA.p, *vOut = (int*)idxB.p;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = pass * 8;
        hipLaunchKernelGGL(sort_hist_kernel<false>, dim3(nb), dim3(SORT_THREADS), 0, s, n, kIn, (const int*)vIn, 1, shift,
                           (unsigned int*)hist.p, nb);
        hipLaunchKernelGGL(sort_scan_rows_kernel, dim3(256), dim3(256), 0, s, (unsigned int*)hist.p, nb, (unsigned int*)hist.p + (size_t)nb * 256);
        hipLaunchKernelGGL(sort_scan_totals_kernel, dim3(1), dim3(256), 0, s, (unsigned int*)hist.p + (size_t)nb * 256);
        hipLaunchKernelGGL(sort_scatter_kernel<false>, dim3(nb), dim3(SORT_THREADS), 0, s, n, kIn, (const int*)vIn, kOut, vOut, 1,
                           shift, (const unsigned int*)hist.p, (const unsigned int*)hist.p + (size_t)nb * 256, nb);
        unsigned int* tk = kIn; kIn = kOut; kOut = tk;
        int* tv = vIn; vIn = vOut; vOut = tv;
    }
    NTR_HIP(hipGetLastError());
    result->sortMs = tPhase.stop_ms();
    const unsigned int* keys = kIn;  // after 4 passes the sorted data is back in the A buffers
    const int* triSorted = vIn;

    // L4: Woop rows in original triangle order
    tPhase.start();
    hipLaunchKernelGGL(lbvh_woop_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_triVtxIndex, d_vtxPos, (float4*)inWoop.p);
    NTR_HIP(hipGetLastError());
    result->woopMs = tPhase.stop_ms();

    // L3: emit, one launch per level, counts stay on the device
    tPhase.start();
    NTR_HIP(hipMemsetAsync(state.p, 0, sizeof(LbvhState), s));
    {
        const unsigned int one = 1;
        NTR_HIP(hipMemcpyAsync(&((LbvhState*)state.p)->lvlNodes[0], &one, 4, hipMemcpyHostToDevice, s));
        const int root[3] = {0, 0, n};
        NTR_HIP(hipMemcpyAsync(q0.p, root, 12, hipMemcpyHostToDevice, s));
    }
    const unsigned int nodeCap = (unsigned int)(nodesCapacity / 64);
    int emitBlocks = (n / 2 + 255) / 256;
    if (emitBlocks < 1) emitBlocks = 1;
    if (emitBlocks > 2048) emitBlocks = 2048;
    int* qIn = (int*)q0.p;
    int* qOut = (int*)q1.p;
    for (int lvl = 0; lvl < 30; lvl++) {  // kernel bit = 29 - lvl (HLBVHBuilder.cpp:344)
        hipLaunchKernelGGL(lbvh_emit_kernel, dim3(emitBlocks), dim3(256), 0, s, lvl, 29 - lvl, leafSize, (LbvhState*)state.p,
                           keys, triSorted, (const float4*)inWoop.p, qIn, qOut, (int*)d_nodes, nodeCap, (float4*)d_triWoop,
                           d_triIndex);
        int* t = qIn; qIn = qOut; qOut = t;
    }
    NTR_HIP(hipGetLastError());
    LbvhState h;
    NTR_HIP(hipMemcpyAsync(&h, state.p, sizeof(h), hipMemcpyDeviceToHost, s));
    result->emitMs = tPhase.stop_ms();
    if (h.overflow) return set_error(NTR_ERR_OVERFLOW, "ntr_lbvh_build: node buffer overflow");
    int numLevels = 0;
    unsigned int numNodes = 0;
    while (numLevels < 31 && h.lvlNodes[numLevels] > 0) { numNodes += h.lvlNodes[numLevels]; numLevels++; }

    // L5: refit, deepest level first
    tPhase.start();
    for (int lvl = numLevels - 1; lvl >= 0; lvl--) {
        int blocks = (int)((h.lvlNodes[lvl] + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(lbvh_refit_kernel, dim3(blocks), dim3(256), 0, s, lvl, epsilon, (const LbvhState*)state.p,
                           d_triVtxIndex, d_vtxPos, triSorted, (int*)d_nodes);
    }
    NTR_HIP(hipGetLastError());
    result->refitMs = tPhase.stop_ms();
    result->seconds = tAll.stop_ms() * 1e-3f;

    const unsigned int leafs = (unsigned int)(h.leafPtr & 0xFFFFFFFFull);
    result->numNodes = (int32_t)numNodes;
    result->numLeaves = (int32_t)leafs;
    result->numLevels = numLevels;
    result->nodesBytes = (int64_t)numNodes * 64;                // HLBVHBuilder.cpp:382-386
    result->triWoopBytes = ((int64_t)n * 3 + leafs) * 16;
    result->triIndexBytes = ((int64_t)n * 3 + leafs) * 4;
    return NTR_OK;
}

}  // extern "C"
