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
// Fallback, top-down (scenes of at most NTR_LBVH_SPLIT = 3072 triangles, n <= leafSize, leaves of more than 32 triangles): lbvh_topdown.h --
//   lbvh_gather_box_kernel (box terms in sorted order), lbvh_top_kernel (one workgroup splits ranges larger than the split size level
//   by level), lbvh_subtree_kernel (one workgroup per smaller range: topology in an LDS entry list, one pair of global atomics, bottom-up
//   refit), lbvh_top_refit_kernel, lbvh_place_kernel (Woop rows straight into the slots the leaves reserved).
// The round-1 / round-2 paths (one launch per level, three-kernel sort passes, cell-table top pass) are kept as a patch:
// scripts/studies/rejected_patches/lbvh_superseded_paths.patch.
// This file is the shipped bottom-up pipeline and the build driver (ntr_lbvh_build); lbvh_workspace.h holds the driver's host helpers.
//
// The tree is the reference's tree: same split rule (highest differing Morton bit at or below the
// level's bit, median when none), same leaf rule (count <= leafSize, or the level's bit is 0), same
// Woop rows and boxes (strict IEEE evaluation of the reference expressions; the reference builds
// these kernels with -use_fast_math so its own bits are toolchain dependent).  Node numbering and
// leaf placement depend on atomic order in the reference (emitTreeKernel.cu:176,303) and are position ranks here; parity is
// checked on the canonical (numbering-independent) form.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
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

// Morton codes as lbvh_morton_kernel, fused with everything else that one pass over the mesh can produce:
//   * the digit histograms of all four radix passes (LDS, then one global add per non-empty bin and workgroup), so
//     that the sort is four one-sweep launches and nothing else;
//   * every triangle's term of its leaf's box (calcLeaf, emitTreeKernel.cu:383-408: min/max over the three
//     vertices, -/+ epsilon) in MESH order -- the vertices are in registers anyway; after the sort one 24-byte
//     gather per triangle replaces the index -> vertex double gather;
//   * clearing the one-sweep tile state.
// Grid-stride over at most a chip's worth of threads, in workgroups of 1024: a workgroup flushes its histograms once (<= 512 x 1024 atomics).
constexpr int MORTON_SLOTS = 2048 * 256;   // threads of the largest grid (a chip's worth: 256 CUs x 2048)

template <int MORTON_THREADS>
__global__ __launch_bounds__(MORTON_THREADS) void lbvh_morton_hist_kernel(int n, const int* __restrict__ tri, const float* __restrict__ pos,
                                                                          F3 lo, F3 step, float eps, unsigned int* __restrict__ keys,
                                                                          int* __restrict__ idx /* or null: the first sort pass numbers the keys itself */, float2* __restrict__ boxMesh /* or null */,
                                                                          TriVerts* __restrict__ triVerts /* mesh order, or null */,
                                                                          unsigned int* __restrict__ hist /* [4][256] */,
                                                                          unsigned long long* __restrict__ tileState, int tileStateWords)
{
    __shared__ unsigned int s_hist[4][256];
    for (int i = threadIdx.x; i < 1024; i += MORTON_THREADS) (&s_hist[0][0])[i] = 0;
    const int gtid = blockIdx.x * MORTON_THREADS + threadIdx.x, gstride = gridDim.x * MORTON_THREADS;
    for (int i = gtid; i < tileStateWords; i += gstride) tileState[i] = 0ull;
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
        if (idx) idx[t] = t;
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

#include "lbvh_topdown.h"   // the top-down fallback (small scenes, oversize leaves)

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
    const unsigned char* runDepth;   // [n] depth (number of ancestors) of the run of equal keys that STARTS at a position, written by
                                     // lbvh_leafmark_kernel for the runs long enough for the depth rule to matter (others: not written, not read)
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

    // the key differences at the cluster's two ends: from the keys (LDS near the tile), or -- after a meeting through memory -- carried
    // along like the second stage does: the sibling's slot holds the difference at ITS far end, which is the parent's (one round trip
    // per level of the border chain less than reading four keys from memory)
    unsigned int dl = 0u, dr = 0u;
    bool carried = false;
    for (;;) {
        if (l == 0 && r == n) {                       // only a single run can get here unmerged: all keys equal
            const unsigned int g = atomicAdd(c.runCount, 1u);
            c.runs[g] = make_int4(-1, 0, 0, n);
            break;
        }
        if (!carried) {
            dl = l > 0 ? (key(l - 1) ^ key(l)) : 0xFFFFFFFFu;
            dr = r < n ? (key(r - 1) ^ key(r)) : 0xFFFFFFFFu;
        }
        carried = false;
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
                mine.farKind = farKind; mine.refH = refH; mine.dFar = sibRight ? dl : dr; mine.lbFar = sibRight ? lbL : lbR;
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
            if (sibRight) dr = sib.dFar; else dl = sib.dFar;   // the parent's far end is the sibling's
            carried = true;
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
// lies strictly inside the run; lbvh_leafmark_kernel has marked the leaf starts of the run's subtree exactly, depth rule included,
// so indices and storage are ranks like everywhere else) and patches the reference its parent holds.  Boxes are folded per child range.
// Where the depth rule cuts the subtree short (a node at depth 29 only has leaf children; a run at depth 30 is a leaf whatever its
// size) the leaf is larger than leafSize; its rows were written by the agglomerate kernel like any leaf's.
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

// A leaf [a, b) the depth rule made larger than leafSize: lbvh_leafmark_kernel marked it as ONE leaf (it counts the run's depth itself),
// so the agglomerate kernel has written its rows, indices and terminator like any other leaf's.  Checked here: a mark inside such a
// leaf would mean the two depth computations disagree -- the build then fails loudly (overflow bit 4) instead of returning a wrong tree.
__device__ __forceinline__ void agg_check_big_leaf(const AggCtx& c, int a, int b)
{
    const int lane = threadIdx.x & 63;
    bool bad = false;
    for (int j = a + 1 + lane; j < b; j += 64) bad = bad || ((c.sBits[j >> 6] >> (j & 63)) & 1ull);
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(&c.st->overflow, 4u);
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
        // no more median levels than parentBit it cannot, and the depth is not needed (the tree's level count does not need it either --
        // a run cluster carries its height to the root).  Otherwise lbvh_leafmark_kernel has counted it (the same test, the same run)
        // and left it at the run's first position: one load instead of a walk up to thirty parents.
        const bool walk = q.x >= 0 && agg_run_height(q.w - q.z, c.leafSize) > parentBit;
        const int depth = walk ? (int)c.runDepth[q.z] : 0;
        if (depth == 0xFF) {                            // lbvh_leafmark_kernel did not count this run's depth: the two depth tests disagree
            if (lane == 0) atomicOr(&c.st->overflow, 4u);
            continue;
        }
        int* parentLink = q.x >= 0 ? c.nodes + (size_t)q.x * 16 + 12 + q.y : nullptr;
        if (depth >= 30) {                              // the parent's level bit is 0: a leaf whatever its size
            if (lane == 0) *parentLink = ~agg_leaf_storage(c, q.z);
            agg_check_big_leaf(c, q.z, q.w);
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
                    if ((ce[k] - cs[k]) > c.leafSize) agg_check_big_leaf(c, cs[k], ce[k]);
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
// (emitTreeKernel.cu:282) taken until a part holds at most leafSize triangles or the depth rule ends the subtree (round 5: the marks
// are EXACT -- until round 4 they ignored the depth rule, and the slots set aside inside leaves it enlarged were squeezed out by a
// relocation pass of five kernels, which the bench scene paid on every build).  Also finds the root's split position, the one boundary
// where the highest bit in which the first and the last key differ flips.  One workgroup per RANK_BLOCK positions: bit masks, the
// block's count and the counts before each 256 positions inside it.
constexpr int MARK_THREADS = 256;              // threads of the mark scan's workgroup
constexpr int MARK_SUBS = RANK_BLOCK / 256;    // prefix counts inside a block: one per 256 positions
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

// THREADS: 256 (four positions per thread, one after the other) or 1024 (one position per thread: the four rounds of a block side by side)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void lbvh_leafmark_kernel(int n, int leafSize, const unsigned int* __restrict__ keys,
                                                                     unsigned long long* __restrict__ sBits, unsigned long long* __restrict__ rBits,
                                                                     unsigned int* __restrict__ blockCount, unsigned int* __restrict__ subBase,
                                                                     unsigned char* __restrict__ runDepth, LbvhState* st)
{
    constexpr int SPAN = RANK_BLOCK + 2 * AGG_HALO;
    __shared__ unsigned int sKeys[SPAN];
    __shared__ int sD[SPAN + 2];                              // sD[k] = d(beg - AGG_HALO + k), valid for 1 <= k < SPAN
    __shared__ unsigned int s_c[MARK_SUBS][4];                // [256 positions][their four waves]
    constexpr int ROUNDS = RANK_BLOCK / THREADS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int beg = blockIdx.x * RANK_BLOCK;
    for (int k = tid; k < SPAN; k += THREADS) {
        const int x = beg - AGG_HALO + k;
        sKeys[k] = (x >= 0 && x < n) ? keys[x] : 0u;
    }
    __syncthreads();
    auto hbit = [](unsigned int x) -> int { return x ? 31 - __clz((int)x) : -1; };
    for (int k = tid; k < SPAN + 2; k += THREADS) {
        const int x = beg - AGG_HALO + k;
        sD[k] = (k >= 1 && k < SPAN && x > 0 && x < n) ? hbit(sKeys[k - 1] ^ sKeys[k]) : 64;
    }
    __syncthreads();
    // highest differing bit across the boundary before position x; every x agg_leaf_of asks for lies within AGG_HALO of the block
    auto dAt = [&](int x) -> int { return sD[x - beg + AGG_HALO]; };
    const int topBit = hbit(keys[0] ^ keys[n - 1]);
    if (blockIdx.x == 0 && tid == 0 && topBit < 0) st->rootSplit = (unsigned int)(n >> 1);   // all keys equal: the first median
    for (int rr = 0; rr < ROUNDS; rr++) {
        const int rel = rr * THREADS + tid;       // position inside the block
        const int i = beg + rel;
        bool mark = false, isRun = false;
        int s0 = 0, e0 = 0;               // the run of equal keys position i lies in (isRun)
        bool deepRun = false;             // ... and the depth rule could cut its subtree short: its depth is needed
        unsigned int myKey = 0;
        if (i < n) {
            myKey = sKeys[AGG_HALO + rel];
            int ls = 0, le = 0;
            agg_leaf_of(i, leafSize, dAt, isRun, ls, le);
            if (!isRun) {
                mark = ls == i;
            } else {
                // the run [s, e) of myKey: the boundary bits held in LDS first; when the run reaches beyond them, gallop over the
                // sorted keys in memory (doubling steps, then a binary search inside the last step)
                const int loEdge = max(beg - AGG_HALO + 1, 0), hiEdge = min(beg + RANK_BLOCK + AGG_HALO, n);   // d valid on [loEdge, hiEdge)
                s0 = i; e0 = i + 1;
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
                // The run's parent splits at the lower of the two bits in which the run's key differs from its neighbours'; every
                // ancestor splits at a bit of its own in [parentBit, 29], so the run's first median node lies at depth <= 30 - parentBit,
                // and with no more median levels than parentBit the depth rule (below) cannot bite.  Otherwise the depth is counted.
                const int dS = s0 > 0 ? hbit(keys[s0 - 1] ^ myKey) : 64, dE = e0 < n ? hbit(keys[e0] ^ myKey) : 64;
                const int parentBit = min(dS, dE);
                deepRun = parentBit != 64 && agg_run_height(e0 - s0, leafSize) > parentBit;
                // (a run whose depth is NOT counted says so at its first position: lbvh_runs_kernel derives "is the depth needed" from the
                // parent bit the agglomerate kernel carries, not from the neighbouring keys as above -- should the two ever disagree it must
                // find this mark, not a byte left over from an earlier build, and fail the build loudly: ADVICE r05)
                if (!deepRun && i == s0) runDepth[s0] = 0xFF;
            }
            if (i > 0 && topBit >= 0 && dAt(i) == topBit) st->rootSplit = (unsigned int)i;
        }
        // Depth of a run = number of its ancestors = bits b at which the keys that share the run key's bits above b hold both values
        // of bit b (that prefix group is split there).  One lane per bit: a lower bound over the sorted keys for the first key of the
        // group's OTHER half, which exists iff the key found still lies in that half.  Wave-uniform; rare (runs long enough to matter).
        int depth = 0;
        for (unsigned long long todo = __ballot(deepRun); todo != 0ull;) {
            const int leader = (int)__builtin_ctzll(todo);
            const unsigned int k = (unsigned int)__shfl((int)myKey, leader);
            const int sLead = __shfl(s0, leader);
            bool split = false;
            if (lane < 30) {
                const unsigned int bit = 1u << lane, group = k & ~(2u * bit - 1u);
                const unsigned int want = (k & bit) ? group : (group | bit);      // first key value of the other half
                int lo = 0, hi = n;                                               // lower bound of `want`
                while (lo < hi) { const int m = (lo + hi) >> 1; if (keys[m] < want) lo = m + 1; else hi = m; }
                split = lo < n && keys[lo] < want + bit;
            }
            const int d = __popcll(__ballot(split));
            const bool same = deepRun && s0 == sLead;
            if (same) depth = d;
            if (same && i == s0) runDepth[s0] = (unsigned char)d;   // for lbvh_runs_kernel (a depth is at most 30)
            todo &= ~__ballot(same);
        }
        if (isRun) {
            // the leaves of the reference's median splits (emitTreeKernel.cu:282) of the run, WITH its depth rule (:289-292, oldLevel == 0):
            // a node at depth 29 only has leaf children, a run whose first node would lie at depth 30 is one leaf -- whatever the sizes
            int a = s0, b = e0, level = depth;
            while (b - a > leafSize && level < 30) {
                const int m = (a + b) >> 1;
                if (i < m) b = m; else a = m;
                level++;
            }
            mark = i == a;
        }
        const unsigned long long mb = __ballot(mark), rb = __ballot(isRun);
        if (lane == 0) {
            sBits[(size_t)((beg + rel) >> 6)] = mb;
            rBits[(size_t)((beg + rel) >> 6)] = rb;
            s_c[rel >> 8][(rel >> 6) & 3] = (unsigned int)__popcll(mb);
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned int acc = 0;
        for (int q = 0; q < MARK_SUBS; q++) {
            subBase[(size_t)blockIdx.x * MARK_SUBS + q] = acc;
            for (int w = 0; w < 4; w++) acc += s_c[q][w];
        }
        blockCount[blockIdx.x] = acc;
    }
}

}  // namespace ntr

using namespace ntr;

#include "lbvh_workspace.h"   // PhaseEvents, the per-device workspace, Carver

extern "C" {

int ntr_lbvh_release_workspace(void)
{
    const int rc = workspace_release();
    const int rc2 = ntr::raysort_scratch_release();
    return rc != NTR_OK ? rc : rc2;
}


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
    // one-sweep tiles: 2048 keys while the launch is latency-bound; 6144 / 8192 for large inputs (fewer tiles to look back over, longer
    // runs per digit in the scatter: 10 M keys 80 -> 71 us per pass, scripts/jobs/gpu_job_r02sort.sh)
    const Tunables tun = tunables();
    const int osItems = (tun.lbvhSortItems == 8 || tun.lbvhSortItems == 16 || tun.lbvhSortItems == 24 || tun.lbvhSortItems == 32)
                            ? tun.lbvhSortItems : (n >= (1 << 23) ? 32 : (n >= (1 << 21) ? 24 : 8));
    const int osTiles = (n + OS_THREADS * osItems - 1) / (OS_THREADS * osItems);
    if (n >= (1 << 27)) return set_error(NTR_ERR_INVALID, "ntr_lbvh_build: at most 2^27 - 1 triangles");

    int spillSize = tun.lbvhSplit;
    if (spillSize < 2) spillSize = 2;
    if (spillSize > 7168) spillSize = 7168;  // 20 bytes of LDS per triangle of a subtree (140 KB), 16-bit positions
    // 0: the whole tree is one hand-over root (scenes of at most `spill` triangles: one subtree workgroup);
    // 2: level-by-level top pass with key probes + subtree workgroups (n <= leafSize: a root over two leaves; leaves of more than
    //    AGG_HALO triangles: the bottom-up path keeps that many neighbours of a tile in LDS);
    // 3: bottom-up emit with ranked indices -- the default
    const int topMode = n <= spillSize ? 0 : ((n <= leafSize || leafSize > AGG_HALO) ? 2 : 3);
    const bool bottomUp = topMode == 3;
    const bool topDown = topMode != 3;

    // workspace: only the slices of the path that runs are reserved (bottom-up: about 175 B per triangle, top-down: about 100 B)
    Carver cv;
    auto takeIf = [&](bool cond, size_t bytes) { return cv.take(cond ? bytes : 0); };
    const size_t oKeysA = cv.take((size_t)n * 4), oKeysB = cv.take((size_t)n * 4);
    const size_t oIdxA = cv.take((size_t)n * 4), oIdxB = cv.take((size_t)n * 4);
    const size_t oWoop = takeIf(topDown, (size_t)n * 24);  // top-down path: box terms in mesh order
    const size_t oQ0 = takeIf(topDown, ((size_t)n + 2) * 16), oQ1 = takeIf(topDown, ((size_t)n + 2) * 16);
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
    const size_t oOsState = cv.take((size_t)osTiles * 256 * 8);
    const size_t oSubList = takeIf(topDown, ((size_t)n / 2 + 2) * 16);
    const size_t oTopLst = takeIf(topDown, ((size_t)n + 2) * 4);
    const size_t oTriBox = takeIf(topDown, (size_t)n * 24), oTriOut = takeIf(topDown, (size_t)n * 4);
    // bottom-up emit: slots of the border meetings, exported roots, vertex records, parent indices, runs, leaf / run marks and their counts
    const size_t oSlot = takeIf(bottomUp, ((size_t)n + 1) * 96);
    const size_t oExports = takeIf(bottomUp, (size_t)aggTiles * AGG_EXPORT_CAP * sizeof(AggExport));
    const size_t oTriVerts = takeIf(bottomUp, (size_t)n * 36);
    const size_t oRunDepth = takeIf(bottomUp, (size_t)n + 1);
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
    {
        const int mortonKeys = tun.lbvhMortonKeys > 0 ? tun.lbvhMortonKeys : 4;
        const int mortonThreads = tun.lbvhMortonThreads > 0 ? tun.lbvhMortonThreads : 1024;   // 2.8 M triangles: 82 -> 65 us against 256-thread workgroups (a quarter of the histogram flushes)
        int mb = (n + mortonThreads * mortonKeys - 1) / (mortonThreads * mortonKeys);
        if (mb > MORTON_SLOTS / mortonThreads) mb = MORTON_SLOTS / mortonThreads;
#define NTR_MORTON_LAUNCH(T)                                                                                                                  \
        hipLaunchKernelGGL(lbvh_morton_hist_kernel<T>, dim3(mb), dim3(T), 0, s, n, d_triVtxIndex, d_vtxPos, lo, step, epsilon, kIn, (int*)nullptr,     \
                           bottomUp ? (float2*)nullptr : (float2*)(ws + oWoop), bottomUp ? (TriVerts*)(ws + oTriVerts) : (TriVerts*)nullptr, osHist, \
                           (unsigned long long*)(ws + oOsState), osTiles * 256)
        if (mortonThreads == 1024) NTR_MORTON_LAUNCH(1024); else if (mortonThreads == 512) NTR_MORTON_LAUNCH(512); else NTR_MORTON_LAUNCH(256);
#undef NTR_MORTON_LAUNCH
    }
    pe.mark(1);

    // L2: stable radix sort by key, 4 passes of 8 bits (the 30-bit code fits)
    for (int pass = 0; pass < 4; pass++) {
        const int shift = pass * 8;
        const unsigned int* dt = osHist + pass * 256;
        unsigned long long* st = (unsigned long long*)(ws + oOsState);
#define NTR_OS_LAUNCH(ITEMS)                                                                                                                \
        onesweep_launch<ITEMS, 0, false>(s, osTiles, n, (const unsigned int*)kIn, pass == 0 ? (const int*)nullptr : (const int*)vIn, kOut, vOut, 1, shift, pass, dt, st,   \
                                         osMisc + pass, osMisc + 4)
        if (osItems == 32) NTR_OS_LAUNCH(32);
        else if (osItems == 24) NTR_OS_LAUNCH(24);
        else if (osItems == 16) NTR_OS_LAUNCH(16);
        else NTR_OS_LAUNCH(8);
#undef NTR_OS_LAUNCH
        unsigned int* tk = kIn; kIn = kOut; kOut = tk;
        int* tv = vIn; vIn = vOut; vOut = tv;
    }
    pe.mark(2);
    const unsigned int* keys = kIn;  // after 4 passes the sorted data is back in the A buffers
    const int* triSorted = vIn;

    // L4 (top-down path only): the per-triangle box terms in sorted order; its Woop rows are produced by lbvh_place_kernel once the leaves
    // have their slots.  The bottom-up path writes Woop rows inside the agglomerate kernel.
    if (topMode != 3)
        hipLaunchKernelGGL(lbvh_gather_box_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, triSorted, (const float2*)(ws + oWoop), (float2*)(ws + oTriBox));
    pe.mark(3);

    // L3 + L5: emit and refit
    const unsigned int nodeCap = (unsigned int)(nodesCapacity / 64);
    LbvhState h;
    {
        EmitCtx c;
        c.st = state; c.keys = keys; c.triBox = (const float2*)(ws + oTriBox); c.triOut = (int*)(ws + oTriOut);
        c.nodes = (int*)d_nodes; c.nodeCap = nodeCap; c.outWoop = (float4*)d_triWoop; c.outIdx = d_triIndex;
        c.leafSize = leafSize; c.subList = (int4*)(ws + oSubList);
        // ranges of at most `spill` triangles become one workgroup's subtree: about 1.5 n / spill of them
        // as large as a workgroup's LDS entry list allows: the LDS levels of a subtree are cheaper than the top pass's
        // global ones (sweep: scripts/studies/lbvh_split_sweep.sh)
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
            a.arrive = (unsigned int*)(ws + oArrive); a.runDepth = (const unsigned char*)(ws + oRunDepth);
            a.runs = (int4*)(ws + oRuns); a.runCount = aggMisc; a.st = state; a.useLds = tun.lbvhAggLds;
            a.exports = (AggExport*)(ws + oExports); a.exportCount = (unsigned int*)(ws + oExportCount); a.slotG = (AggSlotG*)(ws + oSlot);
            a.abortFlag = osMisc + 4;
            // leaf starts and their prefix counts first: everything after it writes to final places
            // small builds are latency chains: one position per thread (262 k triangles: leaf marks 21 -> 16 us); large ones are
            // throughput: four positions per thread in a quarter of the threads (10 M: 70 against 82 us)
            if (tun.lbvhMarkThreads == 256 || (tun.lbvhMarkThreads != 1024 && n >= (1 << 21)))
                hipLaunchKernelGGL(lbvh_leafmark_kernel<256>, dim3(cntTiles), dim3(256), 0, s, n, leafSize, keys, (unsigned long long*)(ws + oLeafBits),
                                   (unsigned long long*)(ws + oRunBits), (unsigned int*)(ws + oTileCount), (unsigned int*)(ws + oSubBase), (unsigned char*)(ws + oRunDepth), state);
            else
                hipLaunchKernelGGL(lbvh_leafmark_kernel<1024>, dim3(cntTiles), dim3(1024), 0, s, n, leafSize, keys, (unsigned long long*)(ws + oLeafBits),
                                   (unsigned long long*)(ws + oRunBits), (unsigned int*)(ws + oTileCount), (unsigned int*)(ws + oSubBase), (unsigned char*)(ws + oRunDepth), state);
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
            hipLaunchKernelGGL(lbvh_runs_kernel, dim3(512), dim3(64), 0, s, a);
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
    {   // builder state and the sort's error word in ONE read-back (they lie in the same cleared block: state, digit histograms, misc)
        static_assert(sizeof(LbvhState) <= 1024, "read-back buffer");
        unsigned char back[1024 + 4 * 256 * 4 + 256 + 64];
        const size_t span = oOsMisc + 64 - oState;
        if (span > sizeof(back)) return set_error(NTR_ERR_INVALID, "ntr_lbvh_build: internal read-back span");
        NTR_HIP(hipMemcpyAsync(back, ws + oState, span, hipMemcpyDeviceToHost, s));
        NTR_HIP(hipStreamSynchronize(s));
        memcpy(&h, back, sizeof(h));
        memcpy(&sortErr, back + (oOsMisc - oState) + 4 * sizeof(unsigned int), sizeof(sortErr));
    }
    result->mortonMs = pe.ms(0, 1);
    result->sortMs = pe.ms(1, 2);
    result->woopMs = pe.ms(2, 3);
    result->emitMs = pe.ms(3, 4);
    result->refitMs = pe.ms(4, 6);
    result->seconds = pe.ms(0, 6) * 1e-3f;
    if (h.overflow & 4u) return set_error(NTR_ERR_HIP, "ntr_lbvh_build: the leaf marks disagree with the depth of a run of equal Morton codes (internal error)");
    if (h.overflow) return set_error(NTR_ERR_OVERFLOW, "ntr_lbvh_build: node buffer overflow");
    if (sortErr) return set_error(NTR_ERR_HIP, "ntr_lbvh_build: a chained scan timed out waiting for a predecessor tile (status %u)", sortErr);
    const unsigned int numNodes = h.nodeCount;
    const int numLevels = (int)h.maxLevel;

    // Compact child references are S32 byte offsets below the sentinel 0x76543210 (CudaBVH.hpp:42-46): a tree with more nodes than that
    // cannot be expressed (the buffers were sized for it, so nothing was written out of bounds; the references are what overflowed)
    if ((unsigned long long)numNodes * 64ull > 0x76543200ull)
        return set_error(NTR_ERR_OVERFLOW, "ntr_lbvh_build: %u nodes exceed what BVHLayout_Compact's 32-bit child offsets address", numNodes);
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
