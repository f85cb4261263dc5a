// lbvh_kernels.hip -- on-device LBVH builder for gfx950.
//
// Rebuilds the pipeline of HLBVHBuilder::buildLBVH (src/rt/bvh/HLBVH/HLBVHBuilder.cpp:451-593).  Default path, ten launches and no
// host read-back before the end:
//   calcMorton      emitTreeKernel.cu:655-691   -> lbvh_morton_hist_kernel: codes, the digit histograms of all four sort passes and a
//                                                  packed 36-B vertex record per triangle in ONE read of the mesh
//   radixSortCuda   radixSort.cu:22-50 (Thrust) -> onesweep_pass_kernel x 4 (radix_sort.h): one launch per 8-bit digit, chained scan
//                                                  with decoupled look-back, match-any ballot ranking
//   emitTreeKernel  emitTreeKernel.cu:233-381   -> BOTTOM-UP: lbvh_leafmark_kernel + lbvh_markscan_kernel find every leaf start from the
//   + createLeaf    :170-231                       sorted keys alone and rank them (node index = rank of the split position, leaf storage =
//   calcWoopKernel  :574-645                       3 * start + leaves before); lbvh_agglomerate_kernel (+ lbvh_agglomerate_top_kernel from
//   calcAABB        :417-562                       2^20 triangles) grows the radix tree from the leaves -- two siblings meet at their split
//   + calcLeaf      :383-408                       position, the second to arrive forms the parent WITH its boxes -- and writes every node
//                                                  word, Woop row, index and terminator once, to its final place; lbvh_runs_kernel adds the
//                                                  reference's median subtrees for runs of equal codes.  (Section "Bottom-up emit" below.)
// Fallback, top-down (scenes of at most NTR_LBVH_SPLIT = 3072 triangles, n <= leafSize, leaves of more than 32 triangles):
//   lbvh_gather_box_kernel (box terms in sorted order), lbvh_top_kernel (one workgroup splits ranges larger than the split size level
//   by level), lbvh_subtree_kernel (one workgroup per smaller range: topology in an LDS entry list, one pair of global atomics, bottom-up
//   refit), lbvh_top_refit_kernel, lbvh_place_kernel (Woop rows straight into the slots the leaves reserved).
// The round-1 / round-2 A/B paths (one launch per level, three-kernel sort passes, cell-table top pass) are compiled only with
// -DNTR_EXPERIMENTS into libntrace_amd_exp.so (`make exp`); tests/test_lbvh_gpu.py runs them against that library.
//
// The tree is the reference's tree: same split rule (highest differing Morton bit at or below the
// level's bit, median when none), same leaf rule (count <= leafSize, or the level's bit is 0), same
// Woop rows and boxes (strict IEEE evaluation of the reference expressions; the reference builds
// these kernels with -use_fast_math so its own bits are toolchain dependent).  Node numbering and
// leaf placement depend on atomic order in the reference (emitTreeKernel.cu:176,303) and are position ranks here; parity is
// checked on the canonical (numbering-independent) form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>

#include <mutex>

#include "ntr_internal.h"
#include "radix_sort.h"

namespace ntr {

struct LbvhState {
    unsigned int lvlNodes[34];   // nodes per level (lvlNodes[0] = 1)
    unsigned int lvlStart[34];   // first node index of each level
    unsigned long long leafPtr;  // (triCount << 32) | leafCount, like g_leafsPtr
    unsigned int overflow;
    unsigned int nodeCount;      // subtree path: next free node index (the root is node 0)
    unsigned int numSub;         // subtree roots emitted by the top pass
    unsigned int subNext;        // work counter of the subtree pass
    unsigned int maxLevel;       // deepest level holding an inner node, plus one
    unsigned int topLevels;      // levels the top pass processed
    unsigned int topLvlOfs[34];  // per-level offsets into the top pass's node list
    unsigned int topTrieLevels;  // cell-table top: deepest trie level holding a top node, plus one
    unsigned int rootSplit;      // bottom-up path: split position of the root (its node gets index 0)
    unsigned int holes;          // bottom-up path: node indices (= leaf terminators) set aside inside leaves the depth rule made larger
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
// a triangle's vertex record: three packed float3 (36 B, 4-byte aligned) moved as three 12-byte accesses
struct __attribute__((packed, aligned(4))) V3 { float x, y, z; };
struct __attribute__((packed, aligned(4))) TriVerts { V3 v[3]; };
static_assert(sizeof(TriVerts) == 36, "TriVerts must be 36 bytes");

#ifdef NTR_EXPERIMENTS // round-1 per-level path (A/B scaffolding: libntrace_amd_exp.so only)
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

#endif  // NTR_EXPERIMENTS
// Morton codes as lbvh_morton_kernel, fused with everything else that one pass over the mesh can produce:
//   * the digit histograms of all four radix passes (LDS, then one global add per non-empty bin and workgroup), so
//     that the sort is four one-sweep launches and nothing else;
//   * every triangle's term of its leaf's box (calcLeaf, emitTreeKernel.cu:383-408: min/max over the three
//     vertices, -/+ epsilon) in MESH order -- the vertices are in registers anyway; after the sort one 24-byte
//     gather per triangle replaces the index -> vertex double gather;
//   * clearing the one-sweep tile state.
// Grid-stride over a bounded number of workgroups, so that the histogram flush stays at <= 2048 x 1024 atomics.
constexpr int MORTON_THREADS = 256;
constexpr int MORTON_MAX_BLOCKS = 2048;
#ifdef NTR_EXPERIMENTS
constexpr int TOP_CELL_BITS = 14;                      // cell-table top pass: the top of the tree is derived from the keys' upper 14 bits
constexpr int TOP_CELLS = 1 << TOP_CELL_BITS;
#endif

__global__ __launch_bounds__(MORTON_THREADS) void lbvh_morton_hist_kernel(int n, const int* __restrict__ tri, const float* __restrict__ pos,
                                                                          F3 lo, F3 step, float eps, unsigned int* __restrict__ keys,
                                                                          int* __restrict__ idx, float2* __restrict__ boxMesh /* or null */,
                                                                          TriVerts* __restrict__ triVerts /* mesh order, or null */,
                                                                          unsigned int* __restrict__ hist /* [4][256] */,
                                                                          unsigned int* __restrict__ tileState, int tileStateWords)
{
    __shared__ unsigned int s_hist[4][256];
    for (int i = threadIdx.x; i < 1024; i += MORTON_THREADS) (&s_hist[0][0])[i] = 0;
    const int gtid = blockIdx.x * MORTON_THREADS + threadIdx.x, gstride = gridDim.x * MORTON_THREADS;
    for (int i = gtid; i < tileStateWords; i += gstride) tileState[i] = 0;
    __syncthreads();
    const float l[3] = {lo.x, lo.y, lo.z}, s[3] = {step.x, step.y, step.z};
    for (int t = gtid; t < n; t += gstride) {
        const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        int cell[3];
        const V3 p0 = *reinterpret_cast<const V3*>(pos + 3 * (size_t)i0), p1 = *reinterpret_cast<const V3*>(pos + 3 * (size_t)i1),
                 p2 = *reinterpret_cast<const V3*>(pos + 3 * (size_t)i2);
        if (triVerts) { triVerts[t].v[0] = p0; triVerts[t].v[1] = p1; triVerts[t].v[2] = p2; }
        const float va[3] = {p0.x, p0.y, p0.z}, vb[3] = {p1.x, p1.y, p1.z}, vc[3] = {p2.x, p2.y, p2.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float a = va[k], b = vb[k], c = vc[k];
            const float mn = fminf(a, fminf(b, c)), mx = fmaxf(a, fmaxf(b, c));
            if (boxMesh) boxMesh[3 * (size_t)t + k] = make_float2(mn - eps, mx + eps);
            const float mid = mn + (mx - mn) / 2.0f;
            const int v = (int)floorf((mid - l[k]) / s[k]);
            cell[k] = min(max(v, 0), 1023);
        }
        const unsigned int key = spread10(cell[0]) | (spread10(cell[1]) << 1) | (spread10(cell[2]) << 2);
        keys[t] = key;
        idx[t] = t;
        atomicAdd(&s_hist[0][key & 255], 1u);
        atomicAdd(&s_hist[1][(key >> 8) & 255], 1u);
        atomicAdd(&s_hist[2][(key >> 16) & 255], 1u);
        atomicAdd(&s_hist[3][(key >> 24) & 255], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += MORTON_THREADS) {
        const unsigned int v = (&s_hist[0][0])[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// After the sort: box terms in sorted order (one 24-byte gather per triangle) and the cell table of the top pass:
// cellStart[c] = first sorted position whose key's upper TOP_CELL_BITS bits are >= c (cellStart[TOP_CELLS] = n).
__global__ __launch_bounds__(256) void lbvh_gather_box_kernel(int n, const unsigned int* __restrict__ keys, const int* __restrict__ triSorted,
                                                              const float2* __restrict__ boxMesh, float2* __restrict__ triBox,
                                                              unsigned int* __restrict__ cellStart)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int t = triSorted[j];
    const float2 a = boxMesh[3 * (size_t)t], b = boxMesh[3 * (size_t)t + 1], c = boxMesh[3 * (size_t)t + 2];
    triBox[3 * (size_t)j] = a; triBox[3 * (size_t)j + 1] = b; triBox[3 * (size_t)j + 2] = c;
#ifdef NTR_EXPERIMENTS
    if (cellStart) {
        const int c1 = (int)(keys[j] >> (30 - TOP_CELL_BITS));
        const int c0 = j ? (int)(keys[j - 1] >> (30 - TOP_CELL_BITS)) : -1;
        for (int cc = c0 + 1; cc <= c1; cc++) cellStart[cc] = (unsigned int)j;
        if (j == n - 1)
            for (int cc = c1 + 1; cc <= TOP_CELLS; cc++) cellStart[cc] = (unsigned int)n;
    }
#endif
}

// ---- Woop rows (emitTreeKernel.cu:574-635) ---------------------------------------------------------
__device__ __forceinline__ void woop_rows_verts(float v0x, float v0y, float v0z, float v1x, float v1y, float v1z, float v2x, float v2y,
                                                float v2z, float4& r0, float4& r1, float4& r2)
{
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
    r0 = make_float4(o0x, i2y, i2z, o0w);
    r1 = make_float4(i0x, i0y, i0z, o1w);
    r2 = make_float4(i1x, i1y, i1z, o2w);
}

__device__ __forceinline__ void woop_rows(const int* __restrict__ tri, const float* __restrict__ pos, int t, float4& r0, float4& r1,
                                          float4& r2)
{
    const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    woop_rows_verts(pos[3 * i0], pos[3 * i0 + 1], pos[3 * i0 + 2], pos[3 * i1], pos[3 * i1 + 1], pos[3 * i1 + 2], pos[3 * i2],
                    pos[3 * i2 + 1], pos[3 * i2 + 2], r0, r1, r2);
}

#ifdef NTR_EXPERIMENTS // round-1 per-level path
__global__ __launch_bounds__(256) void lbvh_woop_kernel(int n, const int* __restrict__ tri, const float* __restrict__ pos,
                                                        float4* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float4 r0, r1, r2;
    woop_rows(tri, pos, t, r0, r1, r2);
    out[3 * t + 0] = r0;
    out[3 * t + 1] = r1;
    out[3 * t + 2] = r2;
}

#endif  // NTR_EXPERIMENTS

// Subtree path, after the emit: Woop rows and original index of every triangle, written straight to the
// slot its leaf reserved (triOut[j] = float4 index of sorted triangle j).
__global__ __launch_bounds__(256) void lbvh_place_kernel(int n, const int* __restrict__ tri, const float* __restrict__ pos,
                                                         const int* __restrict__ triSorted, const int* __restrict__ triOut,
                                                         float4* __restrict__ outWoop, int* __restrict__ outIdx)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int t = triSorted[j], o = triOut[j];
    float4 r0, r1, r2;
    woop_rows(tri, pos, t, r0, r1, r2);
    outWoop[o + 0] = r0;
    outWoop[o + 1] = r1;
    outWoop[o + 2] = r2;
    outIdx[o + 0] = t;
    outIdx[o + 1] = 0;
    outIdx[o + 2] = 0;
}

#ifdef NTR_EXPERIMENTS // round-1 per-level emit / refit kernels
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


#endif  // NTR_EXPERIMENTS
// ---- subtree path: emit + refit with workgroup barriers only ------------------------------------------
// Position where bit `level` of the sorted keys flips inside [nStart, nEnd) (emitTreeKernel.cu:263-280).
// keys[nStart] and keys[nEnd-1] differ in that bit and agree above it, so the flip is unique; K-1
// independent probes per step shorten the dependent-load chain of the plain binary search.
// `keys` is indexed relative to `base` (a subtree's keys live in LDS).
template <int K>
__device__ __forceinline__ int find_split(const unsigned int* keys, int base, int nStart, int nEnd, int level, unsigned int startBit)
{
    int a = nStart, b = nEnd - 1;
    while (b - a > 1) {
        const int len = b - a;
        const int step = len / K;  // K is a power of two; any probes strictly inside (a, b) are valid
        int na = a, nb = b;
#pragma unroll
        for (int j = 1; j < K; j++) {
            const int p = step ? a + j * step : min(a + j, b - 1);
            const unsigned int bit = (keys[p - base] >> level) & 1;
            if (bit == startBit) na = max(na, p); else nb = min(nb, p);
        }
        a = na; b = nb;
    }
    return b;
}

struct EmitCtx {
    LbvhState* st;
    const unsigned int* keys;
    const float2* triBox;  // per sorted triangle: (lo, hi) per axis, epsilon applied (lbvh_tribox_kernel)
    int* triOut;           // per sorted triangle: float4 index of its Woop rows (for lbvh_place_kernel)
    int* nodes;
    unsigned int nodeCap;
    float4* outWoop;
    int* outIdx;
    int leafSize;
    int4* subList;       // (node, start, end, level) of the ranges handed to lbvh_subtree_kernel
    int spill;           // ranges of at most this many triangles are emitted by one workgroup each
};

struct EmitShared {      // LDS bookkeeping of one workgroup
    unsigned long long leafCtr;   // (triangles << 32) | leaves reserved so far, like g_leafsPtr
    unsigned long long leafBase;
    unsigned int nodeCtr, nodeBase, numSub, item, maxLevel;
    unsigned int cnt[3];          // queue lengths of three consecutive levels, rotating
    unsigned int lvlOfs[34];
};

__device__ __forceinline__ void lds_barrier()
{
    // workgroup barrier that orders LDS traffic only: global stores of the emit stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// createLeaf (emitTreeKernel.cu:170-231) without the copy: the leaf's triangles learn their slot, the
// terminator is stored, and the leaf's box (calcLeaf :383-408, folded in stored order from FLT_MAX) goes
// straight into child slot k of its parent.  lbvh_place_kernel fills the Woop rows afterwards.
__device__ __forceinline__ void emit_leaf(const EmitCtx& c, int out, int start, int end, int* nd, int k)
{
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const float2* __restrict__ tb = c.triBox;
    // eight triangles per round trip; indices past the end repeat the last triangle, which min/max ignore
    for (int j = start; j < end; j += 8) {
        float2 b[8][3];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = min(j + u, end - 1);
            b[u][0] = tb[3 * q]; b[u][1] = tb[3 * q + 1]; b[u][2] = tb[3 * q + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], b[u][k].x); hi[k] = fmaxf(hi[k], b[u][k].y); }
        }
    }
    for (int j = start; j < end; j++) c.triOut[j] = out + 3 * (j - start);
    const int tpos = out + 3 * (end - start);
    const float nz = __uint_as_float(0x80000000u);
    c.outWoop[tpos] = make_float4(nz, nz, nz, nz);
    c.outIdx[tpos] = 0;
    float* nf = reinterpret_cast<float*>(nd);
    reinterpret_cast<float4*>(nf)[k] = make_float4(lo[0], hi[0], lo[1], hi[1]);
    reinterpret_cast<float2*>(nf)[4 + k] = make_float2(lo[2], hi[2]);
}

// Level-by-level emit by all threads of ONE workgroup that owns the node / leaf counters of the whole tree in LDS
// (a level costs one barrier and no global atomic).  The queue holds (node, start, end, depth) entries -- `inCount` of
// them are in qA on entry -- and every entry is split exactly as lbvh_emit_kernel splits it (its level bit is
// 29 - depth); ranges of at most c.spill triangles are appended to c.subList (for lbvh_subtree_kernel) instead of the
// next round's queue.  Returns the number of rounds that held nodes; lst receives the node indices round by round
// (offsets in sh.lvlOfs) for the refit; sh.maxLevel = deepest depth that held a node, plus one.
template <int THREADS, int K>
__device__ __forceinline__ int emit_top(const EmitCtx& c, EmitShared& sh, int4* qA, int4* qB, int* lst, unsigned int firstCount)
{
    const int tid = threadIdx.x;
    if (tid == 0) { sh.cnt[0] = firstCount; sh.cnt[1] = 0; sh.cnt[2] = 0; }
    __syncthreads();
    unsigned int total = 0;
    int lv = 0;
    for (; lv < 31; lv++) {
        const unsigned int inCount = sh.cnt[lv % 3];
        if (inCount == 0) break;
        unsigned int* outCount = &sh.cnt[(lv + 1) % 3];
        if (tid == 0) {
            sh.cnt[(lv + 2) % 3] = 0;  // read one round ago, added to one round ahead
            sh.lvlOfs[lv] = total;
        }
        for (unsigned int e = tid; e < inCount; e += THREADS) {
            const int4 q = qA[e];
            const int nIdx = q.x, nStart = q.y, nEnd = q.z, lvl = q.w;
            const int levelBit = 29 - lvl;
            const unsigned int kFirst = c.keys[nStart], kLast = c.keys[nEnd - 1];
            const unsigned int diff = (kFirst ^ kLast) & ((2u << levelBit) - 1u);
            const int level = diff ? 31 - __clz((int)diff) : -1;  // highest differing bit at or below the level's bit
            const int split = level >= 0 ? find_split<K>(c.keys, 0, nStart, nEnd, level, (kFirst >> level) & 1)
                                         : (nStart + nEnd) >> 1;  // identical keys: median (:282)
            const int cs[2] = {nStart, split}, ce[2] = {split, nEnd};
            const bool isLeaf[2] = {(split - nStart) <= c.leafSize || levelBit == 0, (nEnd - split) <= c.leafSize || levelBit == 0};
            const unsigned int inner = (isLeaf[0] ? 0u : 1u) + (isLeaf[1] ? 0u : 1u);
            const unsigned long long lf = (isLeaf[0] ? (((unsigned long long)(split - nStart) << 32) + 1ull) : 0ull) +
                                          (isLeaf[1] ? (((unsigned long long)(nEnd - split) << 32) + 1ull) : 0ull);
            unsigned int childNode = inner ? atomicAdd(&sh.nodeCtr, inner) : 0u;
            unsigned long long lp = lf ? atomicAdd(&sh.leafCtr, lf) : 0ull;
            lst[total + e] = nIdx;
            atomicMax(&sh.maxLevel, (unsigned int)lvl + 1u);
            if (childNode + inner > c.nodeCap) {  // cannot happen with ntr_lbvh_capacity() buffers
                atomicOr(&c.st->overflow, 1u);
                continue;
            }
            int* nd = c.nodes + (size_t)nIdx * 16;
            int ch[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (isLeaf[k]) {
                    const int out = (int)(lp >> 32) * 3 + (int)(lp & 0xFFFFFFFFull);  // createLeaf (:176-181)
                    lp += ((unsigned long long)(ce[k] - cs[k]) << 32) + 1ull;
                    ch[k] = ~out;
                    emit_leaf(c, out, cs[k], ce[k], nd, k);
                } else {
                    if ((ce[k] - cs[k]) <= c.spill) {
                        const unsigned int si = atomicAdd(&sh.numSub, 1u);
                        c.subList[si] = make_int4((int)childNode, cs[k], ce[k], lvl + 1);
                    } else {
                        const unsigned int slot = atomicAdd(outCount, 1u);
                        qB[slot] = make_int4((int)childNode, cs[k], ce[k], lvl + 1);
                    }
                    ch[k] = (int)childNode * 64;
                    childNode++;
                }
            }
            nd[12] = ch[0]; nd[13] = ch[1]; nd[14] = level % 3; nd[15] = 0;
        }
        total += inCount;
        __syncthreads();
        int4* t = qA; qA = qB; qB = t;
    }
    if (tid == 0) sh.lvlOfs[lv] = total;
    return lv;
}

// calcAABB (emitTreeKernel.cu:417-562) for the inner children of one node: the child's box is the union of
// that child's two stored boxes.  Leaf children received their boxes when they were emitted.
__device__ __forceinline__ void refit_node(int* ni, const int* nodes)
{
    float* nf = reinterpret_cast<float*>(ni);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int ch = ni[12 + k];
        if (ch < 0) continue;
        const float4* cn = reinterpret_cast<const float4*>(nodes + (size_t)(ch >> 6) * 16);
        const float4 a = cn[0], b = cn[1], c = cn[2];
        reinterpret_cast<float4*>(nf)[k] = make_float4(fminf(a.x, b.x), fmaxf(a.y, b.y), fminf(a.z, b.z), fmaxf(a.w, b.w));
        reinterpret_cast<float2*>(nf)[4 + k] = make_float2(fminf(c.x, c.z), fmaxf(c.y, c.w));
    }
}

template <int THREADS>
__device__ __forceinline__ void refit_levels(const unsigned int* lvlOfs, int numLv, const int* lst, int* nodes)
{
    for (int lv = numLv - 1; lv >= 0; lv--) {
        const unsigned int b = lvlOfs[lv], e = lvlOfs[lv + 1];
        for (unsigned int q = b + threadIdx.x; q < e; q += THREADS)
            refit_node(nodes + (size_t)lst[q] * 16, nodes);
        __syncthreads();  // the level above reads these boxes (same workgroup, same CU)
    }
}

constexpr int TOP_THREADS = 1024;

__global__ __launch_bounds__(TOP_THREADS) void lbvh_top_kernel(EmitCtx c, int n, int4* qA, int4* qB, int* topLst)
{
    __shared__ EmitShared sh;
    if (threadIdx.x == 0) {
        sh.nodeCtr = 1; sh.nodeBase = 0;  // node 0 is the root
        sh.leafCtr = 0ull; sh.leafBase = 0ull; sh.numSub = 0; sh.maxLevel = 0;
        qA[0] = make_int4(0, 0, n, 0);  // the root: node 0 over all triangles, depth 0
    }
    __syncthreads();
    const int lv = emit_top<TOP_THREADS, 16>(c, sh, qA, qB, topLst, 1u);
    __syncthreads();
    if ((int)threadIdx.x <= lv) c.st->topLvlOfs[threadIdx.x] = sh.lvlOfs[threadIdx.x];
    if (threadIdx.x == 0) {
        c.st->topLevels = (unsigned int)lv;
        c.st->maxLevel = sh.maxLevel;
        c.st->nodeCount = sh.nodeCtr;
        c.st->leafPtr = sh.leafCtr;
        c.st->numSub = sh.numSub;
    }
}

#ifdef NTR_EXPERIMENTS // cell-table top pass (measured slower than the bottom-up emit; A/B only)
// ---- top of the tree from the cell table ---------------------------------------------------------------------------------------
// Above the cells (the keys' upper TOP_CELL_BITS bits) the tree is a function of the cell table alone: a tree node whose keys
// first differ in bit 29 - L is the trie node (L, prefix) whose two halves are both non-empty, its range is the trie node's range
// and its split is the boundary between the halves -- table look-ups, no key probes, no level-by-level dependency.  One workgroup
// keeps the table and a heap-indexed node-index map in LDS and
//   1. classifies all 2^(B+1) trie nodes in parallel: TOP NODE (both halves non-empty, more than `spill` triangles), HAND-OVER
//      ROOT (a child of a top node with at most `spill` triangles: one subtree workgroup each) or OVERSIZE CELL (a single cell
//      with more than `spill` triangles), and gives each a node index from an LDS counter;
//   2. writes every top node (children = leaves, or the node indices of step 1) and the hand-over list;
//   3. splits oversize cells level by level with key probes (emit_top) -- nothing to do for ordinary scenes.
// The depth of a node (needed for the reference's level-bit-0 leaf rule and its level count) is the number of its trie ancestors
// with two non-empty halves.  Heap index h = 2^L + prefix; the cells are the heap's last level.
constexpr int TOP_HEAP = 2 * TOP_CELLS;  // heap indices 1 .. TOP_HEAP-1

struct TopLds {
    unsigned int cell[TOP_CELLS + 1];
    unsigned short idx[TOP_HEAP];
};

__device__ __forceinline__ void trie_range(const unsigned int* cell, unsigned int h, int L, unsigned int& lo, unsigned int& hi)
{
    const unsigned int p = h - (1u << L);
    lo = cell[p << (TOP_CELL_BITS - L)];
    hi = cell[(p + 1) << (TOP_CELL_BITS - L)];
}
__device__ __forceinline__ bool trie_actual(const unsigned int* cell, unsigned int h, int L)  // both halves non-empty (L < TOP_CELL_BITS)
{
    const unsigned int p = h - (1u << L);
    const unsigned int lo = cell[p << (TOP_CELL_BITS - L)], mid = cell[(2 * p + 1) << (TOP_CELL_BITS - L - 1)], hi = cell[(p + 1) << (TOP_CELL_BITS - L)];
    return lo < mid && mid < hi;
}
__device__ __forceinline__ int trie_depth(const unsigned int* cell, unsigned int h, int L)  // trie ancestors with two non-empty halves
{
    int d = 0;
    for (int l = L - 1; l >= 0; l--) {
        h >>= 1;
        d += trie_actual(cell, h, l) ? 1 : 0;
    }
    return d;
}

__global__ __launch_bounds__(TOP_THREADS) void lbvh_top_cells_kernel(EmitCtx c, int n, const unsigned int* __restrict__ cellStart,
                                                                     int* __restrict__ topIdx, int4* qA, int4* qB, int* topLst)
{
    extern __shared__ int smem[];
    TopLds& t = *reinterpret_cast<TopLds*>(smem);
    __shared__ EmitShared sh;
    __shared__ unsigned int s_over, s_trieLevels;
    const int tid = threadIdx.x;
    for (int i = tid; i <= TOP_CELLS; i += TOP_THREADS) t.cell[i] = cellStart[i];
    for (int i = tid; i < TOP_HEAP; i += TOP_THREADS) t.idx[i] = 0xFFFFu;
    if (tid == 0) {
        sh.nodeCtr = 1; sh.nodeBase = 0;  // node 0 is the root
        sh.leafCtr = 0ull; sh.leafBase = 0ull; sh.numSub = 0; sh.maxLevel = 0;
        s_over = 0; s_trieLevels = 0;
    }
    __syncthreads();
    const unsigned int spill = (unsigned int)c.spill, leafSize = (unsigned int)c.leafSize;

    // ---- 1. classify, allocate node indices ------------------------------------------------------------------------------
    for (unsigned int h = 1 + tid; h < (unsigned int)TOP_HEAP; h += TOP_THREADS) {
        const int L = 31 - __clz((int)h);
        unsigned int lo, hi;
        trie_range(t.cell, h, L, lo, hi);
        const unsigned int cnt = hi - lo;
        if (cnt <= leafSize) continue;                                     // a leaf of its parent, or empty
        const bool isCell = L == TOP_CELL_BITS;
        if (!isCell && !trie_actual(t.cell, h, L)) continue;               // one empty half: no tree node here
        // the tree parent: nearest ancestor holding more triangles (its other half is non-empty)
        unsigned int pcnt = 0xFFFFFFFFu;                                   // none: this is the root
        {
            unsigned int a = h;
            for (int l = L - 1; l >= 0; l--) {
                a >>= 1;
                unsigned int alo, ahi;
                trie_range(t.cell, a, l, alo, ahi);
                if (ahi - alo != cnt) { pcnt = ahi - alo; break; }
            }
        }
        const bool top = !isCell && cnt > spill;
        if (!top && pcnt != 0xFFFFFFFFu && pcnt <= spill) continue;       // inside some hand-over root's subtree
        const unsigned int nIdx = pcnt == 0xFFFFFFFFu ? 0u : atomicAdd(&sh.nodeCtr, 1u);
        t.idx[h] = (unsigned short)nIdx;
        if (!top) {
            const int depth = trie_depth(t.cell, h, L);
            if (cnt <= spill) {                                            // hand-over root
                const unsigned int si = atomicAdd(&sh.numSub, 1u);
                c.subList[si] = make_int4((int)nIdx, (int)lo, (int)hi, depth);
            } else {                                                       // oversize cell
                const unsigned int qi = atomicAdd(&s_over, 1u);
                qA[qi] = make_int4((int)nIdx, (int)lo, (int)hi, depth);
            }
        }
    }
    __syncthreads();
    if (sh.nodeCtr > c.nodeCap) {  // cannot happen with ntr_lbvh_capacity() buffers
        if (tid == 0) atomicOr(&c.st->overflow, 1u);
        return;
    }

    // ---- 2. write the top nodes ------------------------------------------------------------------------------------------
    for (unsigned int h = 1 + tid; h < (unsigned int)TOP_CELLS; h += TOP_THREADS) {
        const unsigned int nIdx = t.idx[h];
        const int L = 31 - __clz((int)h);
        unsigned int lo, hi;
        trie_range(t.cell, h, L, lo, hi);
        const bool isTop = nIdx != 0xFFFFu && hi - lo > spill;            // else: nothing, or a hand-over root (its workgroup writes it)
        topIdx[h] = isTop ? (int)nIdx : -1;                                // every entry of the map is written: no clearing pass
        if (!isTop) continue;
        atomicMax(&sh.maxLevel, (unsigned int)trie_depth(t.cell, h, L) + 1u);
        atomicMax(&s_trieLevels, (unsigned int)L + 1u);
        int* nd = c.nodes + (size_t)nIdx * 16;
        int ch[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            unsigned int d = 2 * h + k;
            int dl = L + 1;
            unsigned int clo, chi;
            trie_range(t.cell, d, dl, clo, chi);
            if (chi - clo <= leafSize) {                                   // createLeaf (:170-231)
                const unsigned long long lp = atomicAdd(&sh.leafCtr, ((unsigned long long)(chi - clo) << 32) + 1ull);
                const int out = (int)(lp >> 32) * 3 + (int)(lp & 0xFFFFFFFFull);
                ch[k] = ~out;
                emit_leaf(c, out, (int)clo, (int)chi, nd, k);
                continue;
            }
            while (dl < TOP_CELL_BITS && !trie_actual(t.cell, d, dl)) {  // skip trie nodes with an empty half
                unsigned int llo, lhi;
                trie_range(t.cell, 2 * d, dl + 1, llo, lhi);
                d = 2 * d + (lhi > llo ? 0u : 1u);
                dl++;
            }
            ch[k] = (int)t.idx[d] * 64;
        }
        nd[12] = ch[0]; nd[13] = ch[1]; nd[14] = (29 - L) % 3; nd[15] = 0;
    }
    __syncthreads();

    // ---- 3. oversize cells: level by level with key probes -----------------------------------------------------------------
    const unsigned int over = s_over;
    int lv = 0;
    if (over) lv = emit_top<TOP_THREADS, 16>(c, sh, qA, qB, topLst, over);
    __syncthreads();
    if ((int)tid <= lv) c.st->topLvlOfs[tid] = over ? sh.lvlOfs[tid] : 0u;
    if (tid == 0) {
        c.st->topLevels = (unsigned int)lv;
        c.st->maxLevel = sh.maxLevel;
        c.st->nodeCount = sh.nodeCtr;
        c.st->leafPtr = sh.leafCtr;
        c.st->numSub = sh.numSub;
        c.st->topTrieLevels = s_trieLevels;
    }
}

#endif  // NTR_EXPERIMENTS

// Split position as find_split, for ranges of fewer than 2^16 keys held in LDS: 32-bit probe arithmetic.
template <int LOGK>
__device__ __forceinline__ int find_split_small(const unsigned int* keys, int nStart, int nEnd, int level, unsigned int startBit)
{
    int a = nStart, b = nEnd - 1;
    while (b - a > 1) {
        const int len = b - a;
        int na = a, nb = b;
#pragma unroll
        for (int j = 1; j < (1 << LOGK); j++) {
            const int p = a + ((len * j) >> LOGK);
            const unsigned int bit = (keys[p] >> level) & 1;
            if (bit == startBit) na = max(na, p); else nb = min(nb, p);
        }
        a = na; b = nb;
    }
    return b;
}

// One entry of a subtree's node list in LDS (positions are relative to the subtree's first triangle, which keeps
// every field below 2^16 for the subtree sizes a workgroup's LDS can hold).
struct SubEntry {
    unsigned int range;   // start | end << 16
    unsigned int split;   // split | (level + 1) << 16 | leaf0 << 24 | leaf1 << 25
    unsigned int child;   // entry position of inner child 0 | of inner child 1 << 16
    unsigned int leaf;    // triangles | leaves << 16 reserved by this subtree before this entry's leaves
};

// One workgroup per range of at most `cap` triangles.
//   1. topology, level by level, entirely in LDS (keys, the entry list that doubles as the queue): splits, leaf
//      decisions, positions of the children in the list, leaf storage offsets;
//   2. ONE pair of global atomics reserves the subtree's node indices and leaf storage;
//   3. every entry is written in parallel (node words, leaf boxes and slots): no level dependency any more;
//   4. bottom-up refit over the levels, children found through the LDS list.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void lbvh_subtree_kernel(EmitCtx c, int cap)
{
    extern __shared__ int smem[];
    __shared__ EmitShared sh;
    __shared__ unsigned int s_entCount, s_leafCtr;
    unsigned int* sKeys = reinterpret_cast<unsigned int*>(smem);           // [cap]
    SubEntry* ent = reinterpret_cast<SubEntry*>(smem + cap);               // [cap]: a subtree over m triangles has < m inner nodes
    const unsigned int numSub = c.st->numSub;
    const int tid = threadIdx.x;
    unsigned int deepest = 0;
    for (;;) {
        __syncthreads();
        if (tid == 0) sh.item = atomicAdd(&c.st->subNext, 1u);
        __syncthreads();
        const unsigned int item = sh.item;
        if (item >= numSub) break;
        const int4 root = c.subList[item];
        const int m = root.z - root.y;
        for (int k = tid; k < m; k += THREADS) sKeys[k] = c.keys[root.y + k];
        if (tid == 0) {
            ent[0].range = (unsigned int)m << 16;  // [0, m)
            s_entCount = 1; s_leafCtr = 0;
            sh.lvlOfs[0] = 0;
        }
        lds_barrier();

        // ---- 1. topology ------------------------------------------------------------------------------
        int lv = 0;
        unsigned int lvlBegin = 0, lvlEnd = 1;
        for (int lvl = root.w; lvl < 30 && lvlBegin < lvlEnd; lvl++, lv++) {
            const int levelBit = 29 - lvl;
            for (unsigned int e = lvlBegin + tid; e < lvlEnd; e += THREADS) {
                const unsigned int rg = ent[e].range;
                const int nStart = (int)(rg & 0xFFFFu), nEnd = (int)(rg >> 16);
                const unsigned int kFirst = sKeys[nStart], kLast = sKeys[nEnd - 1];
                const unsigned int diff = (kFirst ^ kLast) & ((2u << levelBit) - 1u);
                const int level = diff ? 31 - __clz((int)diff) : -1;
                const int split = level >= 0 ? find_split_small<3>(sKeys, nStart, nEnd, level, (kFirst >> level) & 1)
                                             : (nStart + nEnd) >> 1;  // identical keys: median (:282)
                const bool leaf0 = (split - nStart) <= c.leafSize || levelBit == 0;
                const bool leaf1 = (nEnd - split) <= c.leafSize || levelBit == 0;
                const unsigned int inner = (leaf0 ? 0u : 1u) + (leaf1 ? 0u : 1u);
                const unsigned int lf = (leaf0 ? ((unsigned int)(split - nStart) + 0x10000u) : 0u) +
                                        (leaf1 ? ((unsigned int)(nEnd - split) + 0x10000u) : 0u);
                unsigned int pos = inner ? atomicAdd(&s_entCount, inner) : 0u;
                const unsigned int leafOfs = lf ? atomicAdd(&s_leafCtr, lf) : 0u;
                unsigned int child = 0;
                if (!leaf0) { ent[pos].range = (unsigned int)nStart | ((unsigned int)split << 16); child = pos; pos++; }
                if (!leaf1) { ent[pos].range = (unsigned int)split | ((unsigned int)nEnd << 16); child |= pos << 16; }
                ent[e].split = (unsigned int)split | ((unsigned int)(level + 1) << 16) | (leaf0 ? (1u << 24) : 0u) | (leaf1 ? (1u << 25) : 0u);
                ent[e].child = child;
                ent[e].leaf = leafOfs;
            }
            lds_barrier();
            lvlBegin = lvlEnd;
            lvlEnd = s_entCount;
            if (tid == 0) sh.lvlOfs[lv + 1] = lvlBegin;
            lds_barrier();  // every thread has read s_entCount before the next level adds to it
        }
        const unsigned int numEnt = lvlBegin;  // every entry of the subtree
        deepest = max(deepest, (unsigned int)(root.w + lv));

        // ---- 2. node indices and leaf storage of the whole subtree ---------------------------------------
        if (tid == 0) {
            const unsigned int lc = s_leafCtr;
            sh.nodeBase = numEnt > 1 ? atomicAdd(&c.st->nodeCount, numEnt - 1) : 0u;
            sh.leafBase = lc ? atomicAdd(&c.st->leafPtr, ((unsigned long long)(lc & 0xFFFFu) << 32) | (unsigned long long)(lc >> 16)) : 0ull;
        }
        __syncthreads();
        const unsigned int nodeBase = sh.nodeBase;
        const unsigned long long leafBase = sh.leafBase;
        const bool overflow = numEnt > 1 && nodeBase + (numEnt - 1) > c.nodeCap;  // cannot happen with ntr_lbvh_capacity() buffers
        if (overflow && tid == 0) atomicOr(&c.st->overflow, 1u);

        // ---- 3. all entries at once ------------------------------------------------------------------------
        for (unsigned int e = tid; e < numEnt && !overflow; e += THREADS) {
            const SubEntry en = ent[e];
            const int nIdx = e == 0 ? root.x : (int)(nodeBase + e - 1);
            const int nStart = root.y + (int)(en.range & 0xFFFFu), nEnd = root.y + (int)(en.range >> 16);
            const int split = root.y + (int)(en.split & 0xFFFFu);
            const int level = (int)((en.split >> 16) & 0xFFu) - 1;
            const bool isLeaf[2] = {((en.split >> 24) & 1u) != 0u, ((en.split >> 25) & 1u) != 0u};
            const int cs[2] = {nStart, split}, ce[2] = {split, nEnd};
            const unsigned int cpos[2] = {en.child & 0xFFFFu, en.child >> 16};
            unsigned long long lp = leafBase + (((unsigned long long)(en.leaf & 0xFFFFu) << 32) | (unsigned long long)(en.leaf >> 16));
            int* nd = c.nodes + (size_t)nIdx * 16;
            int ch[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (isLeaf[k]) {
                    const int out = (int)(lp >> 32) * 3 + (int)(lp & 0xFFFFFFFFull);  // createLeaf (:176-181)
                    lp += ((unsigned long long)(ce[k] - cs[k]) << 32) + 1ull;
                    ch[k] = ~out;
                    emit_leaf(c, out, cs[k], ce[k], nd, k);
                } else {
                    ch[k] = (int)(nodeBase + cpos[k] - 1) * 64;
                }
            }
            nd[12] = ch[0]; nd[13] = ch[1]; nd[14] = level % 3; nd[15] = 0;
        }
        __syncthreads();  // the leaf boxes are visible to the whole workgroup from here

        // ---- 4. refit, deepest level first: an inner child's box is the union of that child's two boxes ---------
        for (int l = lv - 1; l >= 0 && !overflow; l--) {
            const unsigned int b0 = sh.lvlOfs[l], b1 = sh.lvlOfs[l + 1];
            for (unsigned int e = b0 + tid; e < b1; e += THREADS) {
                const SubEntry en = ent[e];
                const int nIdx = e == 0 ? root.x : (int)(nodeBase + e - 1);
                float* nf = reinterpret_cast<float*>(c.nodes + (size_t)nIdx * 16);
                const unsigned int cpos[2] = {en.child & 0xFFFFu, en.child >> 16};
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if ((en.split >> (24 + k)) & 1u) continue;
                    const float4* cn = reinterpret_cast<const float4*>(c.nodes + (size_t)(nodeBase + cpos[k] - 1) * 16);
                    const float4 a = cn[0], b = cn[1], cc = cn[2];
                    reinterpret_cast<float4*>(nf)[k] = make_float4(fminf(a.x, b.x), fmaxf(a.y, b.y), fminf(a.z, b.z), fmaxf(a.w, b.w));
                    reinterpret_cast<float2*>(nf)[4 + k] = make_float2(fminf(cc.x, cc.z), fmaxf(cc.y, cc.w));
                }
            }
            __syncthreads();  // the level above reads these boxes (same workgroup, same CU)
        }
    }
    if (tid == 0 && deepest) atomicMax(&c.st->maxLevel, deepest);
}

__global__ __launch_bounds__(TOP_THREADS) void lbvh_top_refit_kernel(const LbvhState* __restrict__ st, const int* __restrict__ topLst,
                                                                     int* nodes)
{
    __shared__ unsigned int ofs[34];
    const int lv = (int)st->topLevels;
    if ((int)threadIdx.x <= lv) ofs[threadIdx.x] = st->topLvlOfs[threadIdx.x];
    __syncthreads();
    refit_levels<TOP_THREADS>(ofs, lv, topLst, nodes);
}

#ifdef NTR_EXPERIMENTS // cell-table top pass
// Refit of the cell-table top: the oversize cells' levels first (deepest first), then the trie levels bottom-up, each
// level's top nodes found through the heap-indexed map.
__global__ __launch_bounds__(TOP_THREADS) void lbvh_top_cells_refit_kernel(const LbvhState* __restrict__ st, const int* __restrict__ topLst,
                                                                           const int* __restrict__ topIdx, int* nodes)
{
    __shared__ unsigned int ofs[34];
    const int lv = (int)st->topLevels;
    if ((int)threadIdx.x <= lv) ofs[threadIdx.x] = st->topLvlOfs[threadIdx.x];
    __syncthreads();
    if (lv) refit_levels<TOP_THREADS>(ofs, lv, topLst, nodes);
    for (int L = (int)st->topTrieLevels - 1; L >= 0; L--) {
        for (unsigned int h = (1u << L) + threadIdx.x; h < (2u << L); h += TOP_THREADS) {
            const int nIdx = topIdx[h];
            if (nIdx >= 0) refit_node(nodes + (size_t)nIdx * 16, nodes);
        }
        __syncthreads();  // the level above reads these boxes (same workgroup, same CU)
    }
}

#endif  // NTR_EXPERIMENTS


// =====================================================================================================================
// Bottom-up ("agglomerative") emit + refit in one pass, indices by prefix counts.
//
// The reference's tree over distinct keys is the binary radix tree of the sorted keys: a node is a maximal range of keys
// sharing a prefix, split where the next bit flips (emitTreeKernel.cu:233-381); a child becomes a leaf as soon as it holds at
// most leafSize triangles (:289-292).  Instead of descending level by level the tree is grown from its leaves:
//
//   * LEAVES come straight from the keys: the leaf of position i is the largest prefix group around i with at most leafSize
//     members, which only depends on the highest differing bits between key[i] and its leafSize neighbours on either side.
//   * A cluster [l, r) knows which neighbour is its sibling -- the side whose boundary keys share the longer prefix (smaller
//     key[x-1] ^ key[x]) -- and the two siblings MEET at their common boundary B: the first to arrive leaves its box there and
//     stops, the second reads it, forms the parent and goes on (one returning atomic per meeting, no spin anywhere, so no
//     ordering between workgroups is assumed).  Child boxes are complete when a node is formed: the refit comes for free, and
//     fminf / fmaxf unions are bit-identical in any order.
//   * A node is NAMED by its split position B (the root by 0), a leaf by its first sorted position, and the leaf starts are known
//     from the keys alone before the tree is formed (lbvh_leafmark_kernel).  A node's index is the rank of its split position among
//     the leaf starts (root = 0), a leaf's storage is 3 * start + leaves before: nothing is allocated, no atomic counter is on the
//     critical path, the numbering is deterministic -- and every node word, Woop row and triangle index is written ONCE, straight to
//     its final place, by the thread that produces it (no intermediate records, no final pass).
//   * Meetings whose parent range lies inside the workgroup's 512-key tile -- nearly all of them -- use LDS slots and LDS
//     atomics; only clusters that outgrow their tile meet through memory (agent-scope stores of the 40-byte slot, drained, then
//     the atomic; agent-scope loads after it).
//   * RUNS of more than leafSize equal keys are the reference's median-split subtrees (:282), whose leaf rule depends on the
//     depth (level bit 0, :289-292).  The bottom-up pass treats such a run as one opaque cluster that carries its height;
//     lbvh_runs_kernel writes its median nodes under the same naming scheme, and walks up the parent indices for the run's depth
//     only where the depth rule could bite.
// Kernels: lbvh_leafmark_kernel (marks + their prefix counts) -> lbvh_agglomerate_kernel (-> lbvh_agglomerate_top_kernel) -> lbvh_runs_kernel.
// =====================================================================================================================
constexpr int AGG_TILE = 512;
constexpr int AGG_HALO = 32;                // neighbour keys kept on either side of the tile (leafSize <= AGG_HALO)
constexpr int AGG_REF_RUN = 0x7FFFFFFF;     // child reference of a run of equal keys until lbvh_runs_kernel has patched it

struct AggSlot {             // what the first sibling leaves at the meeting point
    float b[6];              // lo.x hi.x lo.y hi.y lo.z hi.z
    unsigned int farKind;    // far end of its range | kind << 28
    unsigned int refH;       // node index (kind 1) | height << 27
};
static_assert(sizeof(AggSlot) == 32, "AggSlot must be 32 bytes");

struct AggSlotG {            // meeting slot in memory: the cluster also carries the key difference at its far end (second stage) and
    float b[6];              // the number of leaves that start before its far end
    unsigned int farKind, refH, dFar, lbFar, pad[2];
};
static_assert(sizeof(AggSlotG) == 48, "AggSlotG must be 48 bytes");

struct AggExport {           // a cluster that has outgrown its tile, handed to the second stage
    float b[6];
    int l;
    unsigned int rKind;      // r | kind << 28
    unsigned int refH;       // node index (kind 1) | height << 27
    unsigned int dl, dr;     // key[l-1] ^ key[l], key[r-1] ^ key[r] (0xFFFFFFFF at the ends of the array)
    unsigned int lbL, lbR;   // leaves that start before l / before r
    unsigned int pad[3];
};
static_assert(sizeof(AggExport) == 64, "AggExport must be 64 bytes");
constexpr int AGG_EXPORT_CAP = 64;   // a tile's clusters with a parent outside it are children of the <= 2 x 30 nodes that cross its two borders

constexpr int RANK_SHIFT = 10;
constexpr int RANK_BLOCK = 1 << RANK_SHIFT;    // positions per prefix-count entry (16 mask words)

struct AggCtx {
    const unsigned int* keys;
    const int* triSorted;        // sorted position -> triangle
    const TriVerts* triVerts;    // per triangle (mesh order): the three vertex positions, 36 B -- ONE random access per triangle
    float eps;
    int n, leafSize;
    // leaf-start marks of lbvh_leafmark_kernel and their prefix counts: every index and storage offset is a rank
    const unsigned long long* sBits;   // bit p: a leaf starts at sorted position p
    const unsigned long long* rBits;   // bit p: position p lies in a run of more than leafSize equal keys
    int numBitWords;
    const unsigned int* blockBase;     // leaf starts before each RANK_BLOCK positions
    const unsigned int* subBase;       // ... inside the block before each 256 positions
    int* nodes;                  // output: BVHLayout_Compact nodes
    float4* outWoop;             // output: Woop rows + terminators
    int* outIdx;                 // output: triangle indices, parallel to outWoop
    unsigned int* arrive;        // [n + 1] meeting counters (memory protocol), zeroed
    int* parentPos;              // [nodes] index of a node's parent (the root is node 0)
    int4* runs;                  // (parent node or -1, side, start, end) of the runs of more than leafSize equal keys
    unsigned int* runCount;
    LbvhState* st;
    int useLds;
    AggExport* exports;          // [tiles][AGG_EXPORT_CAP] (two-stage mode)
    unsigned int* exportCount;   // [tiles], zeroed
    AggSlotG* slotG;             // [n + 1][2] meeting slots in memory
    const unsigned int* abortFlag;     // non-zero: the sort gave up (a look-back timed out) -- the keys are not sorted, and the meeting
                                       // protocol (exactly two arrivals per boundary) only terminates on sorted keys: emit nothing
};

// exclusive rank of position p (set bits before p) = count before its 1024-block + count inside the block before its 256-tile
// (both passed in as `base`) + set bits of the tile's words before p
__device__ __forceinline__ unsigned int agg_rank(const unsigned long long* __restrict__ bits, unsigned int base, int p)
{
    const int w = p >> 6;
    unsigned int acc = base;
    for (int k = (p >> 8) << 2; k < w; k++) acc += (unsigned int)__popcll(bits[k]);
    return acc + (unsigned int)__popcll(bits[w] & ((1ull << (p & 63)) - 1ull));
}
// leaves that start before sorted position p (0 <= p <= n)
__device__ __forceinline__ unsigned int agg_leaves_before(const AggCtx& c, int p)
{
    return agg_rank(c.sBits, c.blockBase[p >> RANK_SHIFT] + c.subBase[p >> 8], p);
}
// A node is identified by its split position, which is where the first leaf of its right child starts: with the leaf starts
// p_0 = 0 < p_1 < ... the node splitting at p_j gets index j, except that the root takes index 0 and the nodes after it move up.
__device__ __forceinline__ int agg_node_index(const AggCtx& c, int pos, int rootSplit)
{
    if (pos == 0 || pos == rootSplit) return 0;
    return (int)agg_leaves_before(c, pos) - (pos > rootSplit ? 1 : 0);
}
// levels of median nodes in the subtree of a run of m > leafSize equal keys (the larger half holds ceil(m / 2)), the depth rule aside
__device__ __forceinline__ int agg_run_height(int m, int leafSize)
{
    int levels = 0;
    while (m > leafSize && levels < 31) { m = (m + 1) >> 1; levels++; }
    return levels;
}
// storage of the leaf that starts at sorted position p: 3 float4 per triangle before it + one terminator per leaf before it
// (createLeaf, emitTreeKernel.cu:176-181, with the leaves numbered in sorted order)
__device__ __forceinline__ int agg_leaf_storage(const AggCtx& c, int p) { return (int)(3u * (unsigned int)p + agg_leaves_before(c, p)); }

// The leaf of sorted position i among distinct-enough keys: the largest prefix group around i with at most leafSize members.  With
// d(x) = highest bit in which key[x-1] and key[x] differ (-1: equal; 64 at the two ends of the array), the highest differing bit
// between key[i] and a neighbour is the maximum of the d's in between (the keys are sorted), so the neighbours' values are two
// non-decreasing sequences, left and right; T = the leafSize-th smallest of both, and the leaf is i plus every neighbour below T.
// T == -1: at least leafSize neighbours carry the same key, i lies inside a run of more than leafSize equal keys (isRun; ls / le
// are not set).
template <class DFn>
__device__ __forceinline__ void agg_leaf_of(int i, int leafSize, DFn d, bool& isRun, int& ls, int& le)
{
    int a = 1, b = 1, T = 64;
    int hl = d(i), hr = d(i + 1);                  // values of the next neighbour to take on either side
    for (int step = 0; step < leafSize; step++) {
        if (hl <= hr) { T = hl; a++; hl = max(hl, d(i - a + 1)); } else { T = hr; b++; hr = max(hr, d(i + b)); }
    }
    isRun = T == -1;
    if (isRun) return;
    ls = i; le = i + 1;
    int m = d(i);
    while (i - ls < leafSize && m < T) { ls--; m = max(m, d(ls)); }
    m = d(i + 1);
    while (le - i <= leafSize && m < T) { le++; m = max(m, d(le)); }
}

__device__ __forceinline__ void agg_store_slot(AggSlotG* dst, const AggSlotG& v)
{
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(&v);
    unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
#pragma unroll
    for (int k = 0; k < 5; k++) __hip_atomic_store(d + k, s[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ AggSlotG agg_load_slot(const AggSlotG* src)
{
    AggSlotG v;
    unsigned long long* d = reinterpret_cast<unsigned long long*>(&v);
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
#pragma unroll
    for (int k = 0; k < 5; k++) d[k] = __hip_atomic_load(s + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.pad[0] = v.pad[1] = 0;
    return v;
}

// the term triangle t contributes to its leaf's box (calcLeaf, emitTreeKernel.cu:383-408): min / max over the vertices, -/+ epsilon
__device__ __forceinline__ void agg_tri_terms(const TriVerts& tv, float eps, float (&term)[6])
{
    const float v[9] = {tv.v[0].x, tv.v[0].y, tv.v[0].z, tv.v[1].x, tv.v[1].y, tv.v[1].z, tv.v[2].x, tv.v[2].y, tv.v[2].z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float a = v[k], b = v[3 + k], c = v[6 + k];
        term[2 * k] = fminf(a, fminf(b, c)) - eps;
        term[2 * k + 1] = fmaxf(a, fmaxf(b, c)) + eps;
    }
}

// box of the sorted positions [a, b), folded from FLT_MAX like calcLeaf (emitTreeKernel.cu:383-408)
__device__ __forceinline__ void agg_fold_box(const AggCtx& c, int a, int b, float (&box)[6])
{
    box[0] = box[2] = box[4] = FLT_MAX;
    box[1] = box[3] = box[5] = -FLT_MAX;
    for (int j = a; j < b; j++) {
        float term[6];
        agg_tri_terms(c.triVerts[c.triSorted[j]], c.eps, term);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            box[2 * k] = fminf(box[2 * k], term[2 * k]);
            box[2 * k + 1] = fmaxf(box[2 * k + 1], term[2 * k + 1]);
        }
    }
}

__device__ __forceinline__ void agg_write_record(int* rec, int id, const float* b0, const float* b1, int link0, int link1, int splitBit)
{
    int* nd = rec + (size_t)id * 16;
    float* nf = reinterpret_cast<float*>(nd);
    reinterpret_cast<float4*>(nf)[0] = make_float4(b0[0], b0[1], b0[2], b0[3]);
    reinterpret_cast<float4*>(nf)[1] = make_float4(b1[0], b1[1], b1[2], b1[3]);
    reinterpret_cast<float4*>(nf)[2] = make_float4(b0[4], b0[5], b1[4], b1[5]);
    reinterpret_cast<int4*>(nd)[3] = make_int4(link0, link1, splitBit, 0);
}

// The second sibling to arrive forms the parent of the clusters [l, r) (its own) and the sibling's, and writes the node -- boxes and
// child references -- to its final place.  No look-up is needed for that: a cluster carries the number of leaves that start before
// its two ends (lbL, lbR), which is all its parent's index (rank of the split position, agg_node_index) and a leaf child's storage
// (agg_leaf_storage) are made of; an inner child carries its index.  The caller's cluster becomes the parent.
// Returns true when the parent is the root.
__device__ __forceinline__ bool agg_form_parent(const AggCtx& c, int rootSplit, bool sibRight, int B, int hb, int& l, int& r, int& kind, int& ref,
                                                int& h, unsigned int& lbL, unsigned int& lbR, float (&box)[6], int sFar, int sKind, int sRef,
                                                int sH, unsigned int sLbFar, const float* sibBox)
{
    const int n = c.n;
    const int L = sibRight ? l : sFar, R = sibRight ? sFar : r;
    // children in tree order: 0 = [L, B), 1 = [B, R)
    const float* b0 = sibRight ? box : sibBox;
    const float* b1 = sibRight ? sibBox : box;
    const int ck[2] = {sibRight ? kind : sKind, sibRight ? sKind : kind};
    const int cr[2] = {sibRight ? ref : sRef, sibRight ? sRef : ref};
    const int hmax = max(h, sH);
    const unsigned int lbB = sibRight ? lbR : lbL;          // leaves before the split position, before L and before R
    const unsigned int lbLo = sibRight ? lbL : sLbFar, lbHi = sibRight ? sLbFar : lbR;
    const bool root = L == 0 && R == n;
    const int idx = (root || B == rootSplit) ? 0 : (int)lbB - (B > rootSplit ? 1 : 0);
    const int cs[2] = {L, B}, ce[2] = {B, R};
    const unsigned int clb[2] = {lbLo, lbB};
    int link[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (ck[k] == 0) {                          // createLeaf: the child is a leaf starting at cs[k]
            link[k] = ~(int)(3u * (unsigned int)cs[k] + clb[k]);
        } else if (ck[k] == 1) {
            link[k] = cr[k] * 64;
            c.parentPos[cr[k]] = idx;
        } else {
            link[k] = AGG_REF_RUN;
            const unsigned int g = atomicAdd(c.runCount, 1u);
            c.runs[g] = make_int4(idx, k | (hb << 1), cs[k], ce[k]);
        }
    }
    agg_write_record(c.nodes, idx, b0, b1, link[0], link[1], hb % 3);
    float ub[6];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        ub[2 * k] = fminf(b0[2 * k], b1[2 * k]);
        ub[2 * k + 1] = fmaxf(b0[2 * k + 1], b1[2 * k + 1]);
    }
    l = L; r = R; kind = 1; ref = idx; h = min(1 + hmax, 31); lbL = lbLo; lbR = lbHi;   // 5 bits travel; only min(h, 30) is used
#pragma unroll
    for (int k = 0; k < 6; k++) box[k] = ub[k];
    if (root) {                                    // the root: deepest level that holds an inner node, plus one
        atomicMax(&c.st->maxLevel, (unsigned int)min(h, 30));
        return true;
    }
    return false;
}

// EXPORT = false: clusters that outgrow their tile go on meeting through memory in this launch (small inputs: one launch less).
// EXPORT = true : they are handed to lbvh_agglomerate_top_kernel instead, so that a workgroup -- and its LDS -- is released as soon as
//                 the work inside its tile is done; the chains of meetings along the tile borders then run as plain threads.
template <bool EXPORT>
__global__ __launch_bounds__(AGG_TILE) void lbvh_agglomerate_kernel(AggCtx c)
{
    if (*c.abortFlag) return;   // workgroup-uniform
    constexpr int BIT_WORDS = AGG_TILE / 64 + 2;               // the tile's mark words and two beyond it (leafSize <= AGG_HALO look-ahead)
    __shared__ unsigned int sKeys[AGG_TILE + 2 * AGG_HALO];   // sKeys[AGG_HALO + k] = key of position tileBeg + k
    __shared__ unsigned int sMeet[AGG_TILE + 1];               // per boundary: 0, or 1 + the compacted index of the cluster waiting there
    __shared__ AggSlot sOwn[AGG_TILE];                        // per cluster (compacted index): what it shows to its sibling
    __shared__ int sWalker[AGG_TILE];                         // start positions of the tile's clusters, compacted; bit 31 = run of equal keys
    __shared__ int sWalkerEnd[AGG_TILE];
    __shared__ unsigned int sWaveCount[AGG_TILE / 64], sNumWalkers, sExports;
    __shared__ float sBox[AGG_TILE][6];                       // box terms of the tile's positions, gathered by all threads at once
    __shared__ unsigned long long sS[BIT_WORDS], sR[BIT_WORDS];   // leaf-start / in-run marks of positions tileBeg ...
    __shared__ unsigned int sPre[BIT_WORDS];                  // leaves that start before the first position of each mark word
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = c.n;
    const int tileBeg = blockIdx.x * AGG_TILE;
    const int tileEnd = min(tileBeg + AGG_TILE, n);
    for (int k = tid; k < AGG_TILE + 2 * AGG_HALO; k += AGG_TILE) {
        const int x = tileBeg - AGG_HALO + k;
        sKeys[k] = (x >= 0 && x < n) ? c.keys[x] : 0u;
    }
    if (tid < BIT_WORDS) {
        const int w = (tileBeg >> 6) + tid;
        sS[tid] = w < c.numBitWords ? c.sBits[w] : 0ull;
        sR[tid] = w < c.numBitWords ? c.rBits[w] : 0ull;
    }
    const int rootSplit = (int)c.st->rootSplit;
    int triOfMine = 0;
    TriVerts tv;
    if (tileBeg + tid < n) {   // one index -> vertex gather (36 contiguous bytes) per position, all in flight together
        triOfMine = c.triSorted[tileBeg + tid];
        tv = c.triVerts[triOfMine];
        float term[6];
        agg_tri_terms(tv, c.eps, term);
#pragma unroll
        for (int k = 0; k < 6; k++) sBox[tid][k] = term[k];
    }
    sMeet[tid] = 0;
    if (tid == 0) { sMeet[AGG_TILE] = 0; sExports = 0; }
    __syncthreads();
    if (tid == 0) {            // AGG_TILE is a multiple of 256: the count before the tile is a block base plus a sub base
        unsigned int acc = c.blockBase[tileBeg >> RANK_SHIFT] + c.subBase[tileBeg >> 8];
        for (int w = 0; w < BIT_WORDS; w++) { sPre[w] = acc; acc += (unsigned int)__popcll(sS[w]); }
    }
    __syncthreads();

    auto key = [&](int x) -> unsigned int {   // sorted key at position x (0 <= x < n)
        const int rel = x - tileBeg + AGG_HALO;
        return (rel >= 0 && rel < AGG_TILE + 2 * AGG_HALO) ? sKeys[rel] : c.keys[x];
    };
    auto leafStartsAt = [&](int x) -> bool {  // tileBeg <= x < tileBeg + 64 * BIT_WORDS
        const int rel = x - tileBeg;
        return (sS[rel >> 6] >> (rel & 63)) & 1ull;
    };
    auto leavesBefore = [&](int x) -> unsigned int {   // leaves that start before position x: LDS near the tile, memory elsewhere
        const int rel = x - tileBeg;
        if (rel >= 0 && rel < 64 * BIT_WORDS) return sPre[rel >> 6] + (unsigned int)__popcll(sS[rel >> 6] & ((1ull << (rel & 63)) - 1ull));
        return agg_leaves_before(c, x);
    };

    const int i = tileBeg + tid;
    if (i < n) {
        // the triangle's Woop rows and index go straight to their final place (calcWoopKernel, emitTreeKernel.cu:574-645): 3 float4
        // per triangle before it + one terminator per leaf that ended before it
        const int o = (int)(3u * (unsigned int)i + leavesBefore(i + 1)) - 1;
        float4 r0, r1, r2;
        woop_rows_verts(tv.v[0].x, tv.v[0].y, tv.v[0].z, tv.v[1].x, tv.v[1].y, tv.v[1].z, tv.v[2].x, tv.v[2].y, tv.v[2].z, r0, r1, r2);
        c.outWoop[o + 0] = r0;
        c.outWoop[o + 1] = r1;
        c.outWoop[o + 2] = r2;
        c.outIdx[o + 0] = triOfMine;
        c.outIdx[o + 1] = 0;
        c.outIdx[o + 2] = 0;
        if (i + 1 == n || leafStartsAt(i + 1)) {      // last triangle of its leaf: the terminator
            const float nz = __uint_as_float(0x80000000u);
            c.outWoop[o + 3] = make_float4(nz, nz, nz, nz);
            c.outIdx[o + 3] = 0;
        }
    }

    // ---- the cluster every position starts in: its leaf (the marks of lbvh_leafmark_kernel), or the whole run of equal keys --------
    bool starts = false, isRun = false;
    int cEnd = 0;
    if (i < n) {
        const unsigned int myKey = sKeys[AGG_HALO + tid];
        isRun = (sR[tid >> 6] >> (tid & 63)) & 1ull;
        if (isRun) {
            starts = i == 0 || key(i - 1) != myKey;
            if (starts) {
                int r = i + 1;
                while (r < n && key(r) == myKey) r++;
                cEnd = r;
            }
        } else {
            starts = leafStartsAt(i);
            if (starts) {                              // a leaf holds at most leafSize <= AGG_HALO positions
                int r = i + 1;
                while (r < n && !leafStartsAt(r)) r++;
                cEnd = r;
            }
        }
    }
    // compact the starting clusters so that they occupy the first lanes of the workgroup
    const unsigned long long m = __ballot(starts);
    if (lane == 0) sWaveCount[wave] = (unsigned int)__popcll(m);
    __syncthreads();
    unsigned int before = 0, all = 0;
    for (int w = 0; w < AGG_TILE / 64; w++) {
        if (w < wave) before += sWaveCount[w];
        all += sWaveCount[w];
    }
    if (starts) {
        const unsigned int wi = before + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        sWalker[wi] = i | (isRun ? (int)0x80000000u : 0);
        sWalkerEnd[wi] = cEnd;
    }
    if (tid == 0) sNumWalkers = all;
    __syncthreads();
    if ((unsigned int)tid >= sNumWalkers) return;

    int l = sWalker[tid] & 0x7FFFFFFF, r = sWalkerEnd[tid];
    int kind = (sWalker[tid] < 0) ? 2 : 0;            // 0 leaf, 1 inner node, 2 run of equal keys
    int ref = 0;                                      // kind 1: the node's index
    int h = kind == 2 ? agg_run_height(r - l, c.leafSize) : 0;   // levels of inner nodes below and including this cluster
    unsigned int lbL = leavesBefore(l), lbR = leavesBefore(r);
    float box[6];
    {   // the cluster's box, folded from FLT_MAX like calcLeaf (:383-408): tile positions from LDS, the few beyond it from memory
        box[0] = box[2] = box[4] = FLT_MAX;
        box[1] = box[3] = box[5] = -FLT_MAX;
        const int inTileEnd = min(r, tileEnd);
        for (int j = l; j < inTileEnd; j++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                box[2 * k] = fminf(box[2 * k], sBox[j - tileBeg][2 * k]);
                box[2 * k + 1] = fmaxf(box[2 * k + 1], sBox[j - tileBeg][2 * k + 1]);
            }
        }
        if (r > inTileEnd) {
            float rest[6];
            agg_fold_box(c, inTileEnd, r, rest);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                box[2 * k] = fminf(box[2 * k], rest[2 * k]);
                box[2 * k + 1] = fmaxf(box[2 * k + 1], rest[2 * k + 1]);
            }
        }
    }

    for (;;) {
        if (l == 0 && r == n) {                       // only a single run can get here unmerged: all keys equal
            const unsigned int g = atomicAdd(c.runCount, 1u);
            c.runs[g] = make_int4(-1, 0, 0, n);
            break;
        }
        const unsigned int dl = l > 0 ? (key(l - 1) ^ key(l)) : 0xFFFFFFFFu;
        const unsigned int dr = r < n ? (key(r - 1) ^ key(r)) : 0xFFFFFFFFu;
        const bool sibRight = dr < dl;                // the sibling lies beyond r: this cluster is the left child
        const int B = sibRight ? r : l;
        const int hb = 31 - __clz((int)(sibRight ? dr : dl));
        // the parent's whole range lies inside this tile iff neither neighbour key of the tile shares the parent's prefix
        bool inTile = false;
        if (c.useLds && B > tileBeg && B < tileEnd) {
            const unsigned int pfx = key(B) >> (hb + 1);
            const bool leftOut = tileBeg == 0 || (sKeys[AGG_HALO - 1] >> (hb + 1)) != pfx;
            const bool rightOut = tileEnd >= n || (sKeys[AGG_HALO + AGG_TILE] >> (hb + 1)) != pfx;
            inTile = leftOut && rightOut;
        }
        const unsigned int farKind = (unsigned int)(sibRight ? l : r) | ((unsigned int)kind << 28);
        const unsigned int refH = (unsigned int)ref | ((unsigned int)h << 27);
        const int side = sibRight ? 0 : 1;
        float sibBox[6];
        unsigned int sFarKind, sRefH, sLbFar;
        if (inTile) {
            const int bl = B - tileBeg;
            // Every cluster keeps its record in its own LDS slot and the two siblings exchange slot numbers at the boundary: the
            // first finds 0 and stops (its record stays put), the second finds the first's number.  LDS executes a wave's operations
            // in order and the exchange serialises the two, so only LDS counters are waited for -- the global stores of the node just
            // formed stay in flight.
            AggSlot mine;
#pragma unroll
            for (int k = 0; k < 6; k++) mine.b[k] = box[k];
            mine.farKind = farKind;
            mine.refH = refH;
            sOwn[tid] = mine;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned int old = atomicExch(&sMeet[bl], (unsigned int)tid + 1u);
            if (old == 0) break;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const AggSlot sib = sOwn[old - 1u];
#pragma unroll
            for (int k = 0; k < 6; k++) sibBox[k] = sib.b[k];
            sFarKind = sib.farKind; sRefH = sib.refH;
            sLbFar = leavesBefore((int)(sib.farKind & 0x0FFFFFFFu));   // the far end lies inside the tile
        } else if (EXPORT) {
            const unsigned int k = atomicAdd(&sExports, 1u);
            if (k < (unsigned int)AGG_EXPORT_CAP) {
                AggExport e;
#pragma unroll
                for (int q = 0; q < 6; q++) e.b[q] = box[q];
                e.l = l; e.rKind = (unsigned int)r | ((unsigned int)kind << 28); e.refH = refH; e.dl = dl; e.dr = dr;
                e.lbL = lbL; e.lbR = lbR; e.pad[0] = e.pad[1] = e.pad[2] = 0;
                c.exports[(size_t)blockIdx.x * AGG_EXPORT_CAP + k] = e;
                atomicAdd(&c.exportCount[blockIdx.x], 1u);
            } else {
                atomicOr(&c.st->overflow, 2u);   // cannot happen: at most 2 x 30 nodes cross a tile's borders
            }
            break;
        } else {
            // Peek first: along a tile border the siblings of a growing cluster are usually waiting already (their arrival was
            // announced after their slot had reached memory), and then this cluster is the second for certain -- it neither
            // publishes its own slot nor touches the counter, it just reads the sibling's: two round trips instead of four.
            if (__hip_atomic_load(&c.arrive[B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                AggSlotG mine;
#pragma unroll
                for (int k = 0; k < 6; k++) mine.b[k] = box[k];
                mine.farKind = farKind; mine.refH = refH; mine.dFar = 0; mine.lbFar = sibRight ? lbL : lbR;
                mine.pad[0] = mine.pad[1] = 0;
                agg_store_slot(&c.slotG[2 * (size_t)B + side], mine);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the slot has reached memory before the arrival is announced
                const unsigned int old = __hip_atomic_fetch_add(&c.arrive[B], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == 0) break;
            }
            asm volatile("" ::: "memory");
            const AggSlotG sib = agg_load_slot(&c.slotG[2 * (size_t)B + (side ^ 1)]);
#pragma unroll
            for (int k = 0; k < 6; k++) sibBox[k] = sib.b[k];
            sFarKind = sib.farKind; sRefH = sib.refH; sLbFar = sib.lbFar;
        }
        // ---- second to arrive: form the parent ------------------------------------------------------------------------------
        if (agg_form_parent(c, rootSplit, sibRight, B, hb, l, r, kind, ref, h, lbL, lbR, box, (int)(sFarKind & 0x0FFFFFFFu), (int)(sFarKind >> 28),
                            (int)(sRefH & 0x07FFFFFFu), (int)(sRefH >> 27), sLbFar, sibBox))
            break;
    }
}

// Second stage of the two-stage mode: one thread per cluster that outgrew its tile; the same meetings, all through memory, with the
// key differences at the cluster's two ends carried along (no key is read any more).
__global__ __launch_bounds__(256) void lbvh_agglomerate_top_kernel(AggCtx c, int numTiles)
{
    if (*c.abortFlag) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int tile = g / AGG_EXPORT_CAP, k = g % AGG_EXPORT_CAP;
    if (tile >= numTiles || (unsigned int)k >= min(c.exportCount[tile], (unsigned int)AGG_EXPORT_CAP)) return;
    const AggExport e = c.exports[(size_t)tile * AGG_EXPORT_CAP + k];
    const int rootSplit = (int)c.st->rootSplit;
    int l = e.l, r = (int)(e.rKind & 0x0FFFFFFFu), kind = (int)(e.rKind >> 28), ref = (int)(e.refH & 0x07FFFFFFu), h = (int)(e.refH >> 27);
    unsigned int dl = e.dl, dr = e.dr;
    unsigned int lbL = e.lbL, lbR = e.lbR;
    float box[6];
#pragma unroll
    for (int q = 0; q < 6; q++) box[q] = e.b[q];
    for (;;) {
        const bool sibRight = dr < dl;
        const int B = sibRight ? r : l;
        const int hb = 31 - __clz((int)(sibRight ? dr : dl));
        const int side = sibRight ? 0 : 1;
        if (__hip_atomic_load(&c.arrive[B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            AggSlotG mine;
#pragma unroll
            for (int q = 0; q < 6; q++) mine.b[q] = box[q];
            mine.farKind = (unsigned int)(sibRight ? l : r) | ((unsigned int)kind << 28);
            mine.refH = (unsigned int)ref | ((unsigned int)h << 27);
            mine.dFar = sibRight ? dl : dr;
            mine.lbFar = sibRight ? lbL : lbR;
            mine.pad[0] = mine.pad[1] = 0;
            agg_store_slot(&c.slotG[2 * (size_t)B + side], mine);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the slot has reached memory before the arrival is announced
            const unsigned int old = __hip_atomic_fetch_add(&c.arrive[B], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == 0) return;
        }
        asm volatile("" ::: "memory");
        const AggSlotG sib = agg_load_slot(&c.slotG[2 * (size_t)B + (side ^ 1)]);
        if (sibRight) dr = sib.dFar; else dl = sib.dFar;
        if (agg_form_parent(c, rootSplit, sibRight, B, hb, l, r, kind, ref, h, lbL, lbR, box, (int)(sib.farKind & 0x0FFFFFFFu), (int)(sib.farKind >> 28),
                            (int)(sib.refH & 0x07FFFFFFu), (int)(sib.refH >> 27), sib.lbFar, sib.b))
            return;
    }
}

// Runs of more than leafSize equal keys: the reference splits them at the median (emitTreeKernel.cu:282) until a part holds at
// most leafSize triangles or the level bit reaches 0 (:289-292), so the subtree depends on the run's depth -- the number of its
// ancestors, found by walking the parent positions.  One wave per run writes the median nodes (named by their split position, which
// lies strictly inside the run; lbvh_leafmark_kernel has marked the leaf starts of the run's subtree as it is WITHOUT the depth rule,
// so indices and storage are ranks like everywhere else) and patches the reference its parent holds.  Boxes are folded per child range.
// Where the depth rule cuts the subtree short (a node at depth 29 only has leaf children; a run at depth 30 is a leaf whatever its
// size) the leaf is larger than the marks assume: its triangles are written again, contiguously, at the leaf's storage, and the
// storage and node indices the marks had set aside inside it stay unused (zero-filled).
// box of the sorted positions [a, b) by a whole wave: lanes stride over the range, min / max across the lanes by shuffles
__device__ __forceinline__ void agg_fold_box_wave(const AggCtx& c, int a, int b, float (&box)[6])
{
    const int lane = threadIdx.x & 63;
    box[0] = box[2] = box[4] = FLT_MAX;
    box[1] = box[3] = box[5] = -FLT_MAX;
    for (int j = a + lane; j < b; j += 64) {
        float term[6];
        agg_tri_terms(c.triVerts[c.triSorted[j]], c.eps, term);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            box[2 * k] = fminf(box[2 * k], term[2 * k]);
            box[2 * k + 1] = fmaxf(box[2 * k + 1], term[2 * k + 1]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            box[2 * k] = fminf(box[2 * k], __shfl_xor(box[2 * k], off));
            box[2 * k + 1] = fmaxf(box[2 * k + 1], __shfl_xor(box[2 * k + 1], off));
        }
    }
}

// a leaf [a, b) of more than leafSize equal keys (depth rule): rows, indices, terminator; zeroes what the marks had set aside
__device__ __forceinline__ void agg_rewrite_big_leaf(const AggCtx& c, int rootSplit, int a, int b)
{
    const int lane = threadIdx.x & 63;
    const int base = agg_leaf_storage(c, a);
    for (int j = a + lane; j < b; j += 64) {
        const int t = c.triSorted[j];
        const TriVerts tv = c.triVerts[t];
        float4 r0, r1, r2;
        woop_rows_verts(tv.v[0].x, tv.v[0].y, tv.v[0].z, tv.v[1].x, tv.v[1].y, tv.v[1].z, tv.v[2].x, tv.v[2].y, tv.v[2].z, r0, r1, r2);
        const int o = base + 3 * (j - a);
        c.outWoop[o + 0] = r0; c.outWoop[o + 1] = r1; c.outWoop[o + 2] = r2;
        c.outIdx[o + 0] = t; c.outIdx[o + 1] = 0; c.outIdx[o + 2] = 0;
        const bool reserved = j > a && ((c.sBits[j >> 6] >> (j & 63)) & 1ull);   // a node index the marks set aside inside the leaf
        if (reserved) {
            const int idx = agg_node_index(c, j, rootSplit);
            int4* nd = reinterpret_cast<int4*>(c.nodes + (size_t)idx * 16);
            nd[0] = nd[1] = nd[2] = nd[3] = make_int4(0, 0, 0, 0);
        }
        const unsigned long long rm = __ballot(reserved);
        if (lane == 0 && rm) atomicAdd(&c.st->holes, (unsigned int)__popcll(rm));
    }
    const int term = base + 3 * (b - a), endAll = agg_leaf_storage(c, b);   // b is a marked leaf start (or n)
    const float nz = __uint_as_float(0x80000000u);
    for (int o = term + lane; o < endAll; o += 64) {
        c.outWoop[o] = o == term ? make_float4(nz, nz, nz, nz) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        c.outIdx[o] = 0;
    }
}

__global__ __launch_bounds__(64) void lbvh_runs_kernel(AggCtx c)
{
    if (*c.abortFlag) return;
    const unsigned int numRuns = *c.runCount;
    const int rootSplit = (int)c.st->rootSplit;
    const int lane = threadIdx.x;
    for (unsigned int g = blockIdx.x; g < numRuns; g += gridDim.x) {   // one wave per run; control flow is wave-uniform
        int4 q = c.runs[g];
        const int parentBit = q.y >> 1;                 // split bit of the run's parent: the parent has at most 29 - parentBit ancestors
        q.y &= 1;
        // The depth rule can only bite when the run's subtree could reach level 30: its root lies at depth <= 30 - parentBit, so with
        // no more median levels than parentBit it cannot, and the walk up the parent indices (a chain of dependent loads) is skipped;
        // the tree's level count does not need it either -- a run cluster carries its height to the root.
        const bool walk = q.x >= 0 && agg_run_height(q.w - q.z, c.leafSize) > parentBit;
        int depth = 0;
        if (walk) {
            depth = 1;
            for (int p = q.x; p != 0; p = c.parentPos[p]) depth++;
        }
        int* parentLink = q.x >= 0 ? c.nodes + (size_t)q.x * 16 + 12 + q.y : nullptr;
        if (depth >= 30) {                              // the parent's level bit is 0: a leaf whatever its size
            if (lane == 0) *parentLink = ~agg_leaf_storage(c, q.z);
            agg_rewrite_big_leaf(c, rootSplit, q.z, q.w);
            continue;
        }
        // a run of at most 64 triangles (the common case) is gathered ONCE, one triangle per lane; the child boxes of its median nodes
        // are then folded across lanes instead of being gathered again per node
        const bool inRegs = (q.w - q.z) <= 64;
        float myTerm[6] = {FLT_MAX, -FLT_MAX, FLT_MAX, -FLT_MAX, FLT_MAX, -FLT_MAX};
        if (inRegs && q.z + lane < q.w) agg_tri_terms(c.triVerts[c.triSorted[q.z + lane]], c.eps, myTerm);
        auto fold = [&](int a, int b, float (&box)[6]) {
            if (!inRegs) { agg_fold_box_wave(c, a, b, box); return; }
            const bool in = lane >= a - q.z && lane < b - q.z;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                box[2 * k] = in ? myTerm[2 * k] : FLT_MAX;
                box[2 * k + 1] = in ? myTerm[2 * k + 1] : -FLT_MAX;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    box[2 * k] = fminf(box[2 * k], __shfl_xor(box[2 * k], off));
                    box[2 * k + 1] = fmaxf(box[2 * k + 1], __shfl_xor(box[2 * k + 1], off));
                }
            }
        };
        // explicit stack of (start, end, depth, split position); a node at depth 29 only has leaf children
        int stS[32], stE[32], stD[32], stP[32];
        int sp = 0;
        const int top = (q.z + q.w) >> 1;               // (the root of an all-equal scene: rootSplit, i.e. index 0)
        if (parentLink && lane == 0) *parentLink = agg_node_index(c, top, rootSplit) * 64;
        stS[0] = q.z; stE[0] = q.w; stD[0] = depth; stP[0] = top;
        sp = 1;
        unsigned int deepest = 0;
        while (sp > 0) {
            sp--;
            const int a = stS[sp], b = stE[sp], d = stD[sp], id = stP[sp];
            const int mid = (a + b) >> 1;
            deepest = max(deepest, (unsigned int)d + 1u);
            float b0[6], b1[6];
            fold(a, mid, b0);
            fold(mid, b, b1);
            const int cs[2] = {a, mid}, ce[2] = {mid, b};
            int link[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if ((ce[k] - cs[k]) <= c.leafSize || d == 29) {
                    link[k] = ~agg_leaf_storage(c, cs[k]);
                    if ((ce[k] - cs[k]) > c.leafSize) agg_rewrite_big_leaf(c, rootSplit, cs[k], ce[k]);
                } else {
                    const int cm = (cs[k] + ce[k]) >> 1;
                    link[k] = agg_node_index(c, cm, rootSplit) * 64;
                    stS[sp] = cs[k]; stE[sp] = ce[k]; stD[sp] = d + 1; stP[sp] = cm;
                    sp++;
                }
            }
            if (lane == 0) {
                // split word of a median split: level = -1 in the reference (no differing bit), and -1 % 3 == -1
                agg_write_record(c.nodes, agg_node_index(c, id, rootSplit), b0, b1, link[0], link[1], -1);
            }
        }
        if (lane == 0 && (walk || q.x < 0)) atomicMax(&c.st->maxLevel, min(deepest, 30u));
    }
}

// Leaf starts of every sorted position, from the keys alone -- BEFORE the tree is formed, so that the bottom-up pass can write nodes,
// Woop rows and indices straight to their final places (indices and storage are ranks of these marks).  Distinct-enough keys: the
// start of agg_leaf_of's leaf.  Inside a run of more than leafSize equal keys: the leaves of the reference's median splits
// (emitTreeKernel.cu:282) taken until a part holds at most leafSize triangles.  Also finds the root's split position, the one boundary
// where the highest bit in which the first and the last key differ flips.  One workgroup per RANK_BLOCK positions: bit masks, the
// block's count and the counts before each 256 positions inside it.
constexpr int MARK_THREADS = 256;              // one workgroup marks one prefix-count block, four positions per thread
constexpr int MARK_SUBS = RANK_BLOCK / MARK_THREADS;
// Exclusive scan of the block counts by one workgroup, four per thread and round with the next round's loads already in flight; the
// total gives the builder state's nodeCount (one inner node per leaf but the first) and leafPtr.  (Folding this into the mark kernel --
// the last block to report in scans -- was measured slower than the extra launch: 103 against 67 us at 10 M triangles.)
__global__ __launch_bounds__(MARK_THREADS) void lbvh_markscan_kernel(int n, int numBlocks, const unsigned int* __restrict__ blockCount,
                                                                     unsigned int* __restrict__ blockBase, LbvhState* st)
{
    __shared__ unsigned int s_scan[MARK_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = 4, ROUND = MARK_THREADS * PER;
    unsigned int carry = 0;
    unsigned int nextv[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int k = tid * PER + q;
        nextv[q] = k < numBlocks ? blockCount[k] : 0u;
    }
    for (int base = 0; base < numBlocks; base += ROUND) {
        unsigned int v[PER], mine = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            v[q] = nextv[q];
            mine += v[q];
            const int k = base + ROUND + tid * PER + q;
            nextv[q] = k < numBlocks ? blockCount[k] : 0u;
        }
        unsigned int ia = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int ua = (unsigned int)__shfl_up((int)ia, off);
            if (lane >= off) ia += ua;
        }
        __syncthreads();
        if (lane == 63) s_scan[wave] = ia;
        __syncthreads();
        unsigned int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < MARK_THREADS / 64; w++) {
            if (w < wave) before += s_scan[w];
            all += s_scan[w];
        }
        unsigned int run = carry + before + ia - mine;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int k = base + tid * PER + q;
            if (k < numBlocks) blockBase[k] = run;
            run += v[q];
        }
        carry += all;
    }
    if (tid == 0) {
        st->nodeCount = carry - 1u;
        st->leafPtr = ((unsigned long long)n << 32) | (unsigned long long)carry;
    }
}

__global__ __launch_bounds__(MARK_THREADS) void lbvh_leafmark_kernel(int n, int leafSize, const unsigned int* __restrict__ keys,
                                                                     unsigned long long* __restrict__ sBits, unsigned long long* __restrict__ rBits,
                                                                     unsigned int* __restrict__ blockCount, unsigned int* __restrict__ subBase,
                                                                     LbvhState* st)
{
    constexpr int SPAN = RANK_BLOCK + 2 * AGG_HALO;
    __shared__ unsigned int sKeys[SPAN];
    __shared__ int sD[SPAN + 2];                              // sD[k] = d(beg - AGG_HALO + k), valid for 1 <= k < SPAN
    __shared__ unsigned int s_c[MARK_SUBS][MARK_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int beg = blockIdx.x * RANK_BLOCK;
    for (int k = tid; k < SPAN; k += MARK_THREADS) {
        const int x = beg - AGG_HALO + k;
        sKeys[k] = (x >= 0 && x < n) ? keys[x] : 0u;
    }
    __syncthreads();
    auto hbit = [](unsigned int x) -> int { return x ? 31 - __clz((int)x) : -1; };
    for (int k = tid; k < SPAN + 2; k += MARK_THREADS) {
        const int x = beg - AGG_HALO + k;
        sD[k] = (k >= 1 && k < SPAN && x > 0 && x < n) ? hbit(sKeys[k - 1] ^ sKeys[k]) : 64;
    }
    __syncthreads();
    // highest differing bit across the boundary before position x; every x agg_leaf_of asks for lies within AGG_HALO of the block
    auto dAt = [&](int x) -> int { return sD[x - beg + AGG_HALO]; };
    const int topBit = hbit(keys[0] ^ keys[n - 1]);
    if (blockIdx.x == 0 && tid == 0 && topBit < 0) st->rootSplit = (unsigned int)(n >> 1);   // all keys equal: the first median
    for (int r = 0; r < MARK_SUBS; r++) {
        const int i = beg + r * MARK_THREADS + tid;
        bool mark = false, isRun = false;
        if (i < n) {
            const unsigned int myKey = sKeys[AGG_HALO + r * MARK_THREADS + tid];
            int ls = 0, le = 0;
            agg_leaf_of(i, leafSize, dAt, isRun, ls, le);
            if (!isRun) {
                mark = ls == i;
            } else {
                // the run [s, e) of myKey: the boundary bits held in LDS first; when the run reaches beyond them, gallop over the
                // sorted keys in memory (doubling steps, then a binary search inside the last step)
                const int loEdge = max(beg - AGG_HALO + 1, 0), hiEdge = min(beg + RANK_BLOCK + AGG_HALO, n);   // d valid on [loEdge, hiEdge)
                int s0 = i, e0 = i + 1;
                while (s0 > loEdge && dAt(s0) == -1) s0--;
                if (s0 > 0 && s0 == loEdge && dAt(s0) == -1) {
                    int hi = s0, stepw = 64, lo = max(hi - stepw, 0);        // key[hi] == myKey; find the first position holding myKey
                    while (lo > 0 && keys[lo] == myKey) { hi = lo; stepw *= 2; lo = max(hi - stepw, 0); }
                    if (keys[lo] == myKey) { s0 = lo; }
                    else {
                        while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (keys[m] == myKey) hi = m; else lo = m; }
                        s0 = hi;
                    }
                }
                while (e0 < hiEdge && dAt(e0) == -1) e0++;
                if (e0 < n && e0 == hiEdge) {                                 // d(hiEdge) is not held: compare the keys
                    if (keys[e0] == myKey) {
                        int lo = e0, stepw = 64, hi = min(lo + stepw, n);    // key[lo] == myKey; find the first position past the run
                        while (hi < n && keys[hi] == myKey) { lo = hi; stepw *= 2; hi = min(lo + stepw, n); }
                        while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (keys[m] == myKey) lo = m; else hi = m; }
                        e0 = hi;
                    }
                }
                int a = s0, b = e0;
                while (b - a > leafSize) {
                    const int m = (a + b) >> 1;
                    if (i < m) b = m; else a = m;
                }
                mark = i == a;
            }
            if (i > 0 && topBit >= 0 && dAt(i) == topBit) st->rootSplit = (unsigned int)i;
        }
        const unsigned long long mb = __ballot(mark), rb = __ballot(isRun);
        if (lane == 0) {
            sBits[(size_t)((beg + r * MARK_THREADS + wave * 64) >> 6)] = mb;
            rBits[(size_t)((beg + r * MARK_THREADS + wave * 64) >> 6)] = rb;
            s_c[r][wave] = (unsigned int)__popcll(mb);
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned int acc = 0;
        for (int q = 0; q < MARK_SUBS; q++) {
            subBase[(size_t)blockIdx.x * MARK_SUBS + q] = acc;
            for (int w = 0; w < MARK_THREADS / 64; w++) acc += s_c[q][w];
        }
        blockCount[blockIdx.x] = acc;
    }
}

}  // namespace ntr

using namespace ntr;

namespace {
// Phase boundaries are recorded as events on the stream and read back after ONE synchronisation at
// the end of the build, so the timed build has no host round trips inside it.
struct PhaseEvents {
    enum { N = 7 };
    hipEvent_t ev[N] = {};
    hipStream_t s;
    explicit PhaseEvents(hipStream_t st) : s(st) { for (auto& e : ev) (void)hipEventCreate(&e); }
    ~PhaseEvents() { for (auto& e : ev) (void)hipEventDestroy(e); }
    void mark(int i) { (void)hipEventRecord(ev[i], s); }
    float ms(int a, int b) { float v = 0; (void)hipEventElapsedTime(&v, ev[a], ev[b]); return v; }
};

// Grow-only scratch memory of the builder, kept between builds: a rebuild per frame must not pay nine
// hipMalloc/hipFree pairs.  One workspace PER DEVICE (one caller per device at a time, as the rest of the
// C-ABI; host threads driving different devices never touch each other's workspace).  A workspace is only
// regrown after the device has drained, so a build still in flight on another stream keeps its memory.
struct Workspace {
    void* p = nullptr;
    size_t bytes = 0;
};
constexpr int kMaxDevices = 64;
Workspace g_ws[kMaxDevices];
std::mutex g_wsMu;

int workspace_reserve(size_t bytes, void** out)
{
    int dev = 0;
    NTR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return set_error(NTR_ERR_INVALID, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_wsMu);
    Workspace& w = g_ws[dev];
    if (w.p && w.bytes < bytes) {
        NTR_HIP(hipDeviceSynchronize());
        NTR_HIP(hipFree(w.p));
        w.p = nullptr; w.bytes = 0;
    }
    if (!w.p) {
        NTR_HIP(hipMalloc(&w.p, bytes));
        w.bytes = bytes;
    }
    *out = w.p;
    return NTR_OK;
}

int workspace_release()
{
    int dev = 0;
    NTR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return set_error(NTR_ERR_INVALID, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_wsMu);
    Workspace& w = g_ws[dev];
    if (w.p) {
        NTR_HIP(hipDeviceSynchronize());
        NTR_HIP(hipFree(w.p));
        w.p = nullptr; w.bytes = 0;
    }
    return NTR_OK;
}

struct Carver {  // 256-byte aligned slices of the workspace
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};

}  // namespace

extern "C" {

int ntr_lbvh_release_workspace(void) { return workspace_release(); }

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
    if (n >= (1 << 28)) return set_error(NTR_ERR_INVALID, "ntr_lbvh_build: at most 2^28 - 1 triangles");
#ifdef NTR_EXPERIMENTS
    const int nb = (n + SORT_TILE - 1) / SORT_TILE;
#endif
    // one-sweep tiles: 2048 keys while the launch is latency-bound; 6144 / 8192 for large inputs (fewer tiles to look back over, longer
    // runs per digit in the scatter: 10 M keys 80 -> 71 us per pass, scripts/jobs/gpu_job_r02sort.sh)
    const int osItems = n >= (1 << 23) ? 32 : (n >= (1 << 21) ? 24 : 8);
    const int osTiles = (n + OS_THREADS * osItems - 1) / (OS_THREADS * osItems);
    const Tunables tun = tunables();
#ifdef NTR_EXPERIMENTS   // A/B scaffolding of rounds 1-2, compiled into libntrace_amd_exp.so only (tests/test_lbvh_gpu.py runs them against it)
    const bool levelSync = tun.lbvhLevelSync != 0;
    const bool legacySort = levelSync || tun.lbvhLegacySort != 0;
    const bool cellsTop = tun.lbvhEmit == 1;
    const bool legacyTop = tun.lbvhLegacyTop != 0;
#else
    constexpr bool levelSync = false, legacySort = false, cellsTop = false, legacyTop = false;
#endif
    if (n >= (1 << 27)) return set_error(NTR_ERR_INVALID, "ntr_lbvh_build: at most 2^27 - 1 triangles");

    int spillSize = tun.lbvhSplit;
    if (spillSize < 2) spillSize = 2;
    if (spillSize > 7168) spillSize = 7168;  // 20 bytes of LDS per triangle of a subtree (140 KB), 16-bit positions
    // 0: the whole tree is one hand-over root (scenes of at most `spill` triangles: one subtree workgroup);
    // 2: level-by-level top pass with key probes + subtree workgroups (n <= leafSize: a root over two leaves; leaves of more than
    //    AGG_HALO triangles: the bottom-up path keeps that many neighbours of a tile in LDS);
    // 3: bottom-up emit with ranked indices -- the default; [experiments: 1 = cell-table top + subtree workgroups]
    const int topMode = n <= spillSize ? 0 : ((legacyTop || n <= leafSize || leafSize > AGG_HALO) ? 2 : (cellsTop ? 1 : 3));
    const bool bottomUp = !levelSync && topMode == 3;
    const bool topDown = !levelSync && topMode != 3;

    // workspace: only the slices of the path that runs are reserved (bottom-up: about 175 B per triangle, top-down: about 100 B)
    Carver cv;
    auto takeIf = [&](bool cond, size_t bytes) { return cv.take(cond ? bytes : 0); };
    const size_t oKeysA = cv.take((size_t)n * 4), oKeysB = cv.take((size_t)n * 4);
    const size_t oIdxA = cv.take((size_t)n * 4), oIdxB = cv.take((size_t)n * 4);
    const size_t oWoop = takeIf(levelSync || topDown, (size_t)n * (levelSync ? 48 : 24));  // per-level path: Woop rows in mesh order; top-down path: box terms in mesh order
    const size_t oQ0 = takeIf(levelSync || topDown, ((size_t)n + 2) * 16), oQ1 = takeIf(levelSync || topDown, ((size_t)n + 2) * 16);
#ifdef NTR_EXPERIMENTS
    const size_t oHist = takeIf(legacySort, ((size_t)nb * 256 + 256) * 4);
#endif
    // cleared by ONE memset per build: builder state, one-sweep digit histograms, error flag and tickets
    const size_t oState = cv.take(sizeof(LbvhState));
    const size_t oOsHist = cv.take(4 * 256 * 4);
    const size_t oOsMisc = cv.take(64);            // [0..3] tickets, [4] error flag
    const size_t oClearEnd = cv.off;
    // bottom-up emit: its zeroed region (meeting counters, export counts, run count, report-in counter) follows, so that ONE memset
    // clears both
    const int aggTiles = (n + AGG_TILE - 1) / AGG_TILE;
    const size_t oArrive = takeIf(bottomUp, ((size_t)n + 1) * 4);
    const size_t oExportCount = takeIf(bottomUp, (size_t)aggTiles * 4);
    const size_t oAggMisc = takeIf(bottomUp, 64);           // [0] number of runs of more than leafSize equal keys
    const int cntTiles = (n + 1 + RANK_BLOCK - 1) / RANK_BLOCK;    // prefix-count blocks
    const size_t oAggZeroEnd = cv.off;
    const size_t oOsState = cv.take((size_t)osTiles * 256 * 4);
    const size_t oSubList = takeIf(topDown, ((size_t)n / 2 + 2) * 16);
    const size_t oTopLst = takeIf(topDown, ((size_t)n + 2) * 4);
    const size_t oTriBox = takeIf(topDown, (size_t)n * 24), oTriOut = takeIf(topDown, (size_t)n * 4);
#ifdef NTR_EXPERIMENTS
    const size_t oCell = takeIf(topMode == 1, ((size_t)TOP_CELLS + 1) * 4), oTopIdx = takeIf(topMode == 1, (size_t)TOP_HEAP * 4);
#endif
    // bottom-up emit: slots of the border meetings, exported roots, vertex records, parent indices, runs, leaf / run marks and their counts
    const size_t oSlot = takeIf(bottomUp, ((size_t)n + 1) * 96);
    const size_t oExports = takeIf(bottomUp, (size_t)aggTiles * AGG_EXPORT_CAP * sizeof(AggExport));
    const size_t oTriVerts = takeIf(bottomUp, (size_t)n * 36);
    const size_t oParentPos = takeIf(bottomUp, ((size_t)n + 1) * 4);
    const size_t oRuns = takeIf(bottomUp, ((size_t)n / 2 + 2) * 16);
    const size_t oLeafBits = takeIf(bottomUp, (size_t)cntTiles * (RANK_BLOCK / 8)), oRunBits = takeIf(bottomUp, (size_t)cntTiles * (RANK_BLOCK / 8));
    const size_t oTileCount = takeIf(bottomUp, (size_t)cntTiles * 4), oTileBase = takeIf(bottomUp, (size_t)cntTiles * 4);
    const size_t oSubBase = takeIf(bottomUp, (size_t)cntTiles * MARK_SUBS * 4);
    void* wsBase = nullptr;
    {
        const int rc = workspace_reserve(cv.off, &wsBase);
        if (rc != NTR_OK) return rc;
    }
    char* ws = (char*)wsBase;
    LbvhState* state = (LbvhState*)(ws + oState);
    unsigned int* osHist = (unsigned int*)(ws + oOsHist);
    unsigned int* osMisc = (unsigned int*)(ws + oOsMisc);

    PhaseEvents pe(s);
    pe.mark(0);

    NTR_HIP(hipMemsetAsync(ws + oState, 0, (bottomUp ? oAggZeroEnd : oClearEnd) - oState, s));

    // L1: Morton codes (step = (max - min) / 1024 on the host, HLBVHBuilder.cpp:76-81)
    F3 lo = {sceneMin[0], sceneMin[1], sceneMin[2]};
    F3 step = {(sceneMax[0] - sceneMin[0]) / 1024.0f, (sceneMax[1] - sceneMin[1]) / 1024.0f, (sceneMax[2] - sceneMin[2]) / 1024.0f};
    unsigned int *kIn = (unsigned int*)(ws + oKeysA), *kOut = (unsigned int*)(ws + oKeysB);
    int *vIn = (int*)(ws + oIdxA), *vOut = (int*)(ws + oIdxB);
#ifdef NTR_EXPERIMENTS
    if (levelSync) {
        hipLaunchKernelGGL(lbvh_morton_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_triVtxIndex, d_vtxPos, lo, step, kIn, vIn);
    } else
#endif
    {
        int mb = (n + MORTON_THREADS * 4 - 1) / (MORTON_THREADS * 4);
        if (mb > MORTON_MAX_BLOCKS) mb = MORTON_MAX_BLOCKS;
        hipLaunchKernelGGL(lbvh_morton_hist_kernel, dim3(mb), dim3(MORTON_THREADS), 0, s, n, d_triVtxIndex, d_vtxPos, lo, step, epsilon, kIn, vIn,
                           bottomUp ? (float2*)nullptr : (float2*)(ws + oWoop), bottomUp ? (TriVerts*)(ws + oTriVerts) : (TriVerts*)nullptr, osHist,
                           (unsigned int*)(ws + oOsState), legacySort ? 0 : osTiles * 256);
    }
    pe.mark(1);

    // L2: stable radix sort by key, 4 passes of 8 bits (the 30-bit code fits)
    for (int pass = 0; pass < 4; pass++) {
        const int shift = pass * 8;
#ifdef NTR_EXPERIMENTS
        if (legacySort) {
            unsigned int* hist = (unsigned int*)(ws + oHist);
            hipLaunchKernelGGL(sort_hist_kernel<false>, dim3(nb), dim3(SORT_THREADS), 0, s, n, kIn, (const int*)vIn, 1, shift, hist, nb);
            hipLaunchKernelGGL(sort_scan_rows_kernel, dim3(256), dim3(256), 0, s, hist, nb, hist + (size_t)nb * 256);
            hipLaunchKernelGGL(sort_scatter_kernel<false>, dim3(nb), dim3(SORT_THREADS), 0, s, n, kIn, (const int*)vIn, kOut, vOut, 1,
                               shift, (const unsigned int*)hist, (const unsigned int*)hist + (size_t)nb * 256, nb);
        } else
#endif
        {
            if (osItems == 32)
                hipLaunchKernelGGL((onesweep_pass_kernel<32, 0>), dim3(osTiles), dim3(OS_THREADS), 0, s, n, (const unsigned int*)kIn, (const int*)vIn, kOut, vOut,
                                   1, shift, pass, (const unsigned int*)(osHist + pass * 256), (unsigned int*)(ws + oOsState), osMisc + pass, osMisc + 4);
            else if (osItems == 24)
                hipLaunchKernelGGL((onesweep_pass_kernel<24, 0>), dim3(osTiles), dim3(OS_THREADS), 0, s, n, (const unsigned int*)kIn, (const int*)vIn, kOut, vOut,
                                   1, shift, pass, (const unsigned int*)(osHist + pass * 256), (unsigned int*)(ws + oOsState), osMisc + pass, osMisc + 4);
            else if (osItems == 16)
                hipLaunchKernelGGL((onesweep_pass_kernel<16, 0>), dim3(osTiles), dim3(OS_THREADS), 0, s, n, (const unsigned int*)kIn, (const int*)vIn, kOut, vOut,
                                   1, shift, pass, (const unsigned int*)(osHist + pass * 256), (unsigned int*)(ws + oOsState), osMisc + pass, osMisc + 4);
            else
                hipLaunchKernelGGL((onesweep_pass_kernel<8, 0>), dim3(osTiles), dim3(OS_THREADS), 0, s, n, (const unsigned int*)kIn, (const int*)vIn, kOut, vOut,
                                   1, shift, pass, (const unsigned int*)(osHist + pass * 256), (unsigned int*)(ws + oOsState), osMisc + pass, osMisc + 4);
        }
        unsigned int* tk = kIn; kIn = kOut; kOut = tk;
        int* tv = vIn; vIn = vOut; vOut = tv;
    }
    pe.mark(2);
    const unsigned int* keys = kIn;  // after 4 passes the sorted data is back in the A buffers
    const int* triSorted = vIn;

    // L4: Woop rows in original triangle order (per-level path), or the per-triangle box terms in sorted order plus the
    // cell table of the top pass (subtree path; its Woop rows are produced by lbvh_place_kernel once the leaves have their slots)
#ifdef NTR_EXPERIMENTS
    if (levelSync)
        hipLaunchKernelGGL(lbvh_woop_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_triVtxIndex, d_vtxPos, (float4*)(ws + oWoop));
    else if (topMode != 3)
        hipLaunchKernelGGL(lbvh_gather_box_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, keys, triSorted, (const float2*)(ws + oWoop),
                           (float2*)(ws + oTriBox), topMode == 1 ? (unsigned int*)(ws + oCell) : (unsigned int*)nullptr);
#else
    if (topMode != 3)
        hipLaunchKernelGGL(lbvh_gather_box_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, keys, triSorted, (const float2*)(ws + oWoop),
                           (float2*)(ws + oTriBox), (unsigned int*)nullptr);
#endif
    pe.mark(3);

    // L3 + L5: emit and refit
    const unsigned int nodeCap = (unsigned int)(nodesCapacity / 64);
    LbvhState h;
#ifdef NTR_EXPERIMENTS
    if (levelSync) {
        LbvhState init;
        memset(&init, 0, sizeof(init));
        init.lvlNodes[0] = 1;
        init.nodeCount = 1;
        NTR_HIP(hipMemcpyAsync(state, &init, sizeof(init), hipMemcpyHostToDevice, s));
        int* q0 = (int*)(ws + oQ0);
        int* q1 = (int*)(ws + oQ1);
        const int root[3] = {0, 0, n};
        NTR_HIP(hipMemcpyAsync(q0, root, 12, hipMemcpyHostToDevice, s));
        int emitBlocks = (n / 2 + 255) / 256;
        if (emitBlocks < 1) emitBlocks = 1;
        if (emitBlocks > 2048) emitBlocks = 2048;
        int* qIn = q0;
        int* qOut = q1;
        for (int lvl = 0; lvl < 30; lvl++) {  // kernel bit = 29 - lvl (HLBVHBuilder.cpp:344)
            hipLaunchKernelGGL(lbvh_emit_kernel, dim3(emitBlocks), dim3(256), 0, s, lvl, 29 - lvl, leafSize, state, keys, triSorted,
                               (const float4*)(ws + oWoop), qIn, qOut, (int*)d_nodes, nodeCap, (float4*)d_triWoop, d_triIndex);
            int* t = qIn; qIn = qOut; qOut = t;
        }
        pe.mark(4);
        // the refit launches are sized from the level counts: one read-back, as the reference does per level
        NTR_HIP(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, s));
        NTR_HIP(hipStreamSynchronize(s));
        int numLevels = 0;
        while (numLevels < 31 && h.lvlNodes[numLevels] > 0) numLevels++;
        for (int lvl = numLevels - 1; lvl >= 0; lvl--) {
            int blocks = (int)((h.lvlNodes[lvl] + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(lbvh_refit_kernel, dim3(blocks), dim3(256), 0, s, lvl, epsilon, (const LbvhState*)state, d_triVtxIndex,
                               d_vtxPos, triSorted, (int*)d_nodes);
        }
        pe.mark(5);
        pe.mark(6);
    } else
#endif
    {
        EmitCtx c;
        c.st = state; c.keys = keys; c.triBox = (const float2*)(ws + oTriBox); c.triOut = (int*)(ws + oTriOut);
        c.nodes = (int*)d_nodes; c.nodeCap = nodeCap; c.outWoop = (float4*)d_triWoop; c.outIdx = d_triIndex;
        c.leafSize = leafSize; c.subList = (int4*)(ws + oSubList);
        // ranges of at most `spill` triangles become one workgroup's subtree: about 1.5 n / spill of them
        // as large as a workgroup's LDS entry list allows: the LDS levels of a subtree are cheaper than the top pass's
        // global ones (sweep: scripts/lbvh_split_sweep.sh)
        c.spill = spillSize;
        int4* q0 = (int4*)(ws + oQ0);
        int4* q1 = (int4*)(ws + oQ1);
        unsigned int* aggMisc = (unsigned int*)(ws + oAggMisc);
        auto launch_subtrees = [&](int blocks) -> int {
            const int subThreads = tun.lbvhSubThreads;
            const size_t subLds = (size_t)c.spill * (4 + 16);  // keys + entry list
            if (subLds > 65536) {
                const void* fn = subThreads == 64 ? (const void*)lbvh_subtree_kernel<64>
                               : subThreads == 256 ? (const void*)lbvh_subtree_kernel<256> : (const void*)lbvh_subtree_kernel<128>;
                NTR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)subLds));
            }
            if (subThreads == 64)
                hipLaunchKernelGGL(lbvh_subtree_kernel<64>, dim3(blocks), dim3(64), subLds, s, c, c.spill);
            else if (subThreads == 256)
                hipLaunchKernelGGL(lbvh_subtree_kernel<256>, dim3(blocks), dim3(256), subLds, s, c, c.spill);
            else
                hipLaunchKernelGGL(lbvh_subtree_kernel<128>, dim3(blocks), dim3(128), subLds, s, c, c.spill);
            return NTR_OK;
        };
        if (topMode == 3) {
            AggCtx a;
            a.keys = keys; a.triSorted = triSorted; a.triVerts = (const TriVerts*)(ws + oTriVerts);
            a.eps = epsilon; a.n = n; a.leafSize = leafSize;
            a.sBits = (const unsigned long long*)(ws + oLeafBits); a.rBits = (const unsigned long long*)(ws + oRunBits);
            a.numBitWords = cntTiles * (RANK_BLOCK / 64); a.blockBase = (const unsigned int*)(ws + oTileBase);
            a.subBase = (const unsigned int*)(ws + oSubBase);
            a.nodes = (int*)d_nodes; a.outWoop = (float4*)d_triWoop; a.outIdx = d_triIndex;
            a.arrive = (unsigned int*)(ws + oArrive); a.parentPos = (int*)(ws + oParentPos);
            a.runs = (int4*)(ws + oRuns); a.runCount = aggMisc; a.st = state; a.useLds = tun.lbvhAggLds;
            a.exports = (AggExport*)(ws + oExports); a.exportCount = (unsigned int*)(ws + oExportCount); a.slotG = (AggSlotG*)(ws + oSlot);
            a.abortFlag = osMisc + 4;
            // leaf starts and their prefix counts first: everything after it writes to final places
            hipLaunchKernelGGL(lbvh_leafmark_kernel, dim3(cntTiles), dim3(MARK_THREADS), 0, s, n, leafSize, keys, (unsigned long long*)(ws + oLeafBits),
                               (unsigned long long*)(ws + oRunBits), (unsigned int*)(ws + oTileCount), (unsigned int*)(ws + oSubBase), state);
            hipLaunchKernelGGL(lbvh_markscan_kernel, dim3(1), dim3(MARK_THREADS), 0, s, n, cntTiles, (const unsigned int*)(ws + oTileCount),
                               (unsigned int*)(ws + oTileBase), state);
            pe.mark(4);
            // two stages for large inputs (a workgroup leaves as soon as its tile is done, the chains along the tile borders run as plain
            // threads of a second launch); one launch for small ones, where the extra launch costs more than it saves
            // (a third variant -- the exported roots of 16 tiles meeting through LDS in one workgroup, stage after stage until one workgroup
            // holds the root -- was built and measured slower at every size: a lone thread's chain of meetings costs about 0.7 us per level
            // in instruction issue alone, little less than the two agent-scope round trips it replaces, and every stage adds a launch and
            // 8-10 us of set-up; 262 k triangles 0.205 against 0.170 ms, 10 M 1.276 against 1.264 ms)
            const bool staged = tun.lbvhAggStaged < 0 ? n >= (1 << 20) : tun.lbvhAggStaged != 0;
            if (staged && a.useLds) {
                hipLaunchKernelGGL(lbvh_agglomerate_kernel<true>, dim3(aggTiles), dim3(AGG_TILE), 0, s, a);
                hipLaunchKernelGGL(lbvh_agglomerate_top_kernel, dim3((aggTiles * AGG_EXPORT_CAP + 255) / 256), dim3(256), 0, s, a, aggTiles);
            } else {
                hipLaunchKernelGGL(lbvh_agglomerate_kernel<false>, dim3(aggTiles), dim3(AGG_TILE), 0, s, a);
            }
            pe.mark(5);
            hipLaunchKernelGGL(lbvh_runs_kernel, dim3(2048), dim3(64), 0, s, a);
            pe.mark(6);
        } else {
        if (topMode == 0) {
            // node 0 over all triangles at depth 0 goes straight to a subtree workgroup
            LbvhState init;
            memset(&init, 0, sizeof(init));
            init.nodeCount = 1;
            init.numSub = 1;
            NTR_HIP(hipMemcpyAsync(state, &init, sizeof(init), hipMemcpyHostToDevice, s));
            const int root[4] = {0, 0, n, 0};
            NTR_HIP(hipMemcpyAsync(c.subList, root, 16, hipMemcpyHostToDevice, s));
#ifdef NTR_EXPERIMENTS
        } else if (topMode == 1) {
            NTR_HIP(hipFuncSetAttribute((const void*)lbvh_top_cells_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TopLds)));
            hipLaunchKernelGGL(lbvh_top_cells_kernel, dim3(1), dim3(TOP_THREADS), sizeof(TopLds), s, c, n, (const unsigned int*)(ws + oCell),
                               (int*)(ws + oTopIdx), q0, q1, (int*)(ws + oTopLst));
#endif
        } else {
            hipLaunchKernelGGL(lbvh_top_kernel, dim3(1), dim3(TOP_THREADS), 0, s, c, n, q0, q1, (int*)(ws + oTopLst));
        }
        pe.mark(4);
        int subBlocks = n / 2 + 1;
        const int subMax = 256 * (2048 / (tun.lbvhSubThreads > 0 ? tun.lbvhSubThreads : 128));
        if (subBlocks > subMax) subBlocks = subMax;
        if (topMode == 0) subBlocks = 1;
        {
            const int rc = launch_subtrees(subBlocks);
            if (rc != NTR_OK) return rc;
        }
        pe.mark(5);
#ifdef NTR_EXPERIMENTS
        if (topMode == 1)
            hipLaunchKernelGGL(lbvh_top_cells_refit_kernel, dim3(1), dim3(TOP_THREADS), 0, s, (const LbvhState*)state, (const int*)(ws + oTopLst),
                               (const int*)(ws + oTopIdx), (int*)d_nodes);
        else
#endif
        if (topMode == 2)
            hipLaunchKernelGGL(lbvh_top_refit_kernel, dim3(1), dim3(TOP_THREADS), 0, s, (const LbvhState*)state, (const int*)(ws + oTopLst),
                               (int*)d_nodes);
        hipLaunchKernelGGL(lbvh_place_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_triVtxIndex, d_vtxPos, triSorted,
                           (const int*)(ws + oTriOut), (float4*)d_triWoop, d_triIndex);
        pe.mark(6);
        }
    }
    NTR_HIP(hipGetLastError());
    unsigned int sortErr = 0;
    NTR_HIP(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, s));
    NTR_HIP(hipMemcpyAsync(&sortErr, osMisc + 4, sizeof(sortErr), hipMemcpyDeviceToHost, s));
    NTR_HIP(hipStreamSynchronize(s));
    result->mortonMs = pe.ms(0, 1);
    result->sortMs = pe.ms(1, 2);
    result->woopMs = pe.ms(2, 3);
    result->emitMs = pe.ms(3, 4);
    result->refitMs = pe.ms(4, 6);
    result->seconds = pe.ms(0, 6) * 1e-3f;
    if (h.overflow) return set_error(NTR_ERR_OVERFLOW, "ntr_lbvh_build: node buffer overflow");
    if (sortErr) return set_error(NTR_ERR_HIP, "ntr_lbvh_build: a chained scan timed out waiting for a predecessor tile (status %u)", sortErr);
    int numLevels = 0;
    unsigned int numNodes = 0;
    if (levelSync) {
        while (numLevels < 31 && h.lvlNodes[numLevels] > 0) { numNodes += h.lvlNodes[numLevels]; numLevels++; }
    } else {
        numNodes = h.nodeCount;
        numLevels = (int)h.maxLevel;
    }
    (void)legacySort; (void)cellsTop;

    // Compact child references are S32 byte offsets below the sentinel 0x76543210 (CudaBVH.hpp:42-46): a tree with more nodes than that
    // cannot be expressed (the buffers were sized for it, so nothing was written out of bounds; the references are what overflowed)
    if ((unsigned long long)numNodes * 64ull > 0x76543200ull)
        return set_error(NTR_ERR_OVERFLOW, "ntr_lbvh_build: %u nodes exceed what BVHLayout_Compact's 32-bit child offsets address", numNodes);
    // Bottom-up path: where the depth rule (level bit 0) made a leaf of more than leafSize equal keys, the node indices and terminator
    // slots the leaf marks had set aside inside it stay unused (zero-filled): the buffers' extents include them, the counts do not.
    const unsigned int leafs = (unsigned int)(h.leafPtr & 0xFFFFFFFFull);
    result->numNodes = (int32_t)(numNodes - h.holes);
    result->numLeaves = (int32_t)(leafs - h.holes);
    result->numLevels = numLevels;
    result->nodesBytes = (int64_t)numNodes * 64;                // HLBVHBuilder.cpp:382-386
    result->triWoopBytes = ((int64_t)n * 3 + leafs) * 16;
    result->triIndexBytes = ((int64_t)n * 3 + leafs) * 4;
    return NTR_OK;
}

}  // extern "C"
