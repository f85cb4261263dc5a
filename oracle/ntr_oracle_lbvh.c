/*
 * ntr_oracle_lbvh.c -- CPU ORACLE (test infrastructure only; see ntr_oracle.h).
 *
 * Sequential restatement of the reference's GPU LBVH builder (HLBVHBuilder::buildLBVH,
 * src/rt/bvh/HLBVH/HLBVHBuilder.cpp:451-593):
 *   calcMorton      emitTreeKernel.cu:655-691   (host step = (max-min)/1024, HLBVHBuilder.cpp:76-81)
 *   radixSortCuda   radixSort.cu:22-50          (thrust::sort_by_key = stable ascending sort by uint key;
 *                                                Thrust is a third-party dependency of unpinned version --
 *                                                a pure integer stable sort, uniquely determined)
 *   calcWoop        emitTreeKernel.cu:574-635
 *   emitTreeKernel  emitTreeKernel.cu:233-381, createLeaf :170-231, host loop HLBVHBuilder.cpp:337-361
 *   calcAABB        emitTreeKernel.cu:417-562, calcLeaf :383-408, host loop HLBVHBuilder.cpp:427-439
 *
 * The reference kernels are built with -use_fast_math (and nvcc's default FMA contraction), so
 * their float bits are not reproducible off that toolchain.  The canonical semantics used here
 * are the strict IEEE-754 evaluation of the source expressions in source order: binary32
 * operations, no contraction, true division; `1.0/(float expr)` is a binary64 divide narrowed
 * to binary32 exactly as written (emitTreeKernel.cu:589).
 *
 * Node numbering and leaf placement in the reference depend on atomic ordering
 * (emitTreeKernel.cu:176,303).  This restatement is deterministic: nodes are numbered level by
 * level in queue order, leaves are placed in creation order (node order, child 0 before child 1).
 * Comparisons against a GPU build go through orc_bvh_canonical_hash() (order independent).
 *
 * PARITY UNPINNED by the reference (no goldens, reference unbuildable here).
 */
#include "ntr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* CUDA fminf/fmaxf are the PTX min.f32/max.f32: a NaN operand yields the other operand and, for zeros
 * of opposite sign, min returns -0 and max returns +0 (PTX ISA, "min"/"max": -0.0 < +0.0).  C's fminf
 * leaves the zero case to the implementation, so it is spelled out here (it matters only for epsilon = 0,
 * where a box plane can be a signed zero). */
static inline float cu_fminf(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) & 0x80000000u) ? a : b;
    return a < b ? a : b;
}
static inline float cu_fmaxf(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) & 0x80000000u) ? b : a;
    return a > b ? a : b;
}

/* emitTreeKernel.cu:647-653 */
static inline uint32_t spread(uint32_t n)
{
    n &= 0x3ff;
    n = (n ^ (n << 16)) & 0xff0000ff;
    n = (n ^ (n << 8)) & 0x0300f00f;
    n = (n ^ (n << 4)) & 0x030c30c3;
    return (n ^ (n << 2)) & 0x09249249;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* emitTreeKernel.cu:655-691 */
void orc_lbvh_morton(int32_t numTris, const int32_t* tri, const float* pos,
                     const float sceneMin[3], const float sceneMax[3], uint32_t* keys, int32_t* idx)
{
    float step[3];
    for (int k = 0; k < 3; k++) step[k] = (sceneMax[k] - sceneMin[k]) / 1024.0f; /* HLBVHBuilder.cpp:76-81 */
    for (int32_t t = 0; t < numTris; t++) {
        const float* a = pos + 3 * (size_t)tri[3 * t + 0];
        const float* b = pos + 3 * (size_t)tri[3 * t + 1];
        const float* c = pos + 3 * (size_t)tri[3 * t + 2];
        int cell[3];
        for (int k = 0; k < 3; k++) {
            float lo = cu_fminf(a[k], cu_fminf(b[k], c[k]));
            float hi = cu_fmaxf(a[k], cu_fmaxf(b[k], c[k]));
            float mid = lo + (hi - lo) / 2.0f;
            float q = (mid - sceneMin[k]) / step[k];
            cell[k] = clampi((int)floorf(q), 0, 1023);
        }
        keys[t] = spread((uint32_t)cell[0]) | (spread((uint32_t)cell[1]) << 1) | (spread((uint32_t)cell[2]) << 2);
        idx[t] = t;
    }
}

/* emitTreeKernel.cu:574-635 */
void orc_lbvh_woop(int32_t numTris, const int32_t* tri, const float* pos, float* out)
{
    for (int32_t t = 0; t < numTris; t++) {
        const float* v0 = pos + 3 * (size_t)tri[3 * t + 0];
        const float* v1 = pos + 3 * (size_t)tri[3 * t + 1];
        const float* v2 = pos + 3 * (size_t)tri[3 * t + 2];
        float c0x = v0[0] - v2[0], c0y = v0[1] - v2[1], c0z = v0[2] - v2[2];
        float c1x = v1[0] - v2[0], c1y = v1[1] - v2[1], c1z = v1[2] - v2[2];
        /* fcross(c0,c1) */
        float c2x = c0y * c1z - c0z * c1y, c2y = c0z * c1x - c0x * c1z, c2z = c0x * c1y - c0y * c1x;

        float den = c0x * (c2z * c1y - c1z * c2y) - c0y * (c2z * c1x - c1z * c2x) + c0z * (c2y * c1x - c1y * c2x);
        float det = (float)(1.0 / (double)den);

        float i0x = (c2z * c1y - c1z * c2y) * det;
        float i0y = -(c2z * c1x - c1z * c2x) * det;
        float i0z = (c2y * c1x - c1y * c2x) * det;
        float i1x = -(c2z * c0y - c0z * c2y) * det;
        float i1y = (c2z * c0x - c0z * c2x) * det;
        float i1z = -(c2y * c0x - c0y * c2x) * det;
        float i2x = (c1z * c0y - c0z * c1y) * det;
        float i2y = -(c1z * c0x - c0z * c1x) * det;
        float i2z = (c1y * c0x - c0y * c1x) * det;

        /* fdot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z */
        float o0w = -((-i2x) * v2[0] + (-i2y) * v2[1] + (-i2z) * v2[2]);
        float o1w = (-i0x) * v2[0] + (-i0y) * v2[1] + (-i0z) * v2[2];
        float o2w = (-i1x) * v2[0] + (-i1y) * v2[1] + (-i1z) * v2[2];

        float* o = out + 12 * (size_t)t;
        o[0] = i2x; o[1] = i2y; o[2] = i2z; o[3] = o0w;
        o[4] = i0x; o[5] = i0y; o[6] = i0z; o[7] = o1w;
        o[8] = i1x; o[9] = i1y; o[10] = i1z; o[11] = o2w;
        if (o[0] == 0.0f) o[0] = 0.0f; /* -0 would alias the terminator (:621-622) */
    }
}

/* thrust::sort_by_key: stable ascending by key (LSD radix, 4 x 8 bits) */
static void stable_sort_by_key(int32_t n, uint32_t* keys, int32_t* vals)
{
    uint32_t* k2 = (uint32_t*)malloc((size_t)(n > 0 ? n : 1) * 4);
    int32_t* v2 = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * 4);
    for (int pass = 0; pass < 4; pass++) {
        size_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        int sh = pass * 8;
        for (int32_t i = 0; i < n; i++) cnt[((keys[i] >> sh) & 255) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (int32_t i = 0; i < n; i++) {
            size_t p = cnt[(keys[i] >> sh) & 255]++;
            k2[p] = keys[i];
            v2[p] = vals[i];
        }
        memcpy(keys, k2, (size_t)n * 4);
        memcpy(vals, v2, (size_t)n * 4);
    }
    free(k2);
    free(v2);
}

typedef struct Leafs {
    float* woopOut;     /* float4 units */
    int32_t* idxOut;
    const float* inWoop;
    const int32_t* triSorted;
    int64_t numLeafs, allTris;
} Leafs;

/* createLeaf (emitTreeKernel.cu:170-231), COMPACT_LAYOUT + WOOP_TRIANGLES */
static int32_t create_leaf(Leafs* L, int32_t start, int32_t end)
{
    int32_t numTris = end - start;
    int64_t out = L->allTris * 3 + L->numLeafs; /* float4 index */
    for (int32_t i = 0; i < numTris; i++) {
        int32_t t = L->triSorted[start + i];
        memcpy(L->woopOut + 4 * (out + 3 * (int64_t)i), L->inWoop + 12 * (size_t)t, 48);
        L->idxOut[out + 3 * (int64_t)i + 0] = t;
        L->idxOut[out + 3 * (int64_t)i + 1] = 0;
        L->idxOut[out + 3 * (int64_t)i + 2] = 0;
    }
    uint32_t term[4] = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};
    memcpy(L->woopOut + 4 * (out + 3 * (int64_t)numTris), term, 16);
    L->idxOut[out + 3 * (int64_t)numTris] = 0;
    L->allTris += numTris;
    L->numLeafs += 1;
    return (int32_t)~out;
}

/* calcLeaf (emitTreeKernel.cu:383-408) */
static void calc_leaf(const int32_t* tri, const float* pos, const int32_t* triSorted, int32_t start, int32_t end,
                      float eps, float lo[3], float hi[3])
{
    for (int32_t i = start; i < end; i++) {
        int32_t t = triSorted[i];
        const float* a = pos + 3 * (size_t)tri[3 * t + 0];
        const float* b = pos + 3 * (size_t)tri[3 * t + 1];
        const float* c = pos + 3 * (size_t)tri[3 * t + 2];
        for (int k = 0; k < 3; k++) {
            lo[k] = cu_fminf(lo[k], cu_fminf(a[k], cu_fminf(b[k], c[k])) - eps);
            hi[k] = cu_fmaxf(hi[k], cu_fmaxf(a[k], cu_fmaxf(b[k], c[k])) + eps);
        }
    }
}

int orc_lbvh_build(int32_t n, const int32_t* tri, int32_t numVerts, const float* pos,
                   const float sceneMin[3], const float sceneMax[3], int32_t leafSize, float epsilon, OrcLbvh* out)
{
    (void)numVerts;
    memset(out, 0, sizeof(*out));
    if (n < 1 || leafSize < 1) return -1;
    uint32_t* keys = (uint32_t*)malloc((size_t)n * 4);
    int32_t* sorted = (int32_t*)malloc((size_t)n * 4);
    float* inWoop = (float*)malloc((size_t)n * 48);
    orc_lbvh_morton(n, tri, pos, sceneMin, sceneMax, keys, sorted);
    stable_sort_by_key(n, keys, sorted);
    orc_lbvh_woop(n, tri, pos, inWoop);

    size_t cap = (size_t)n + 2;
    int32_t* nodes = (int32_t*)calloc(cap * 16, 4);
    float* woop = (float*)calloc(((size_t)n * 4 + 4) * 4, 4);
    int32_t* tidx = (int32_t*)calloc((size_t)n * 4 + 4, 4);
    int32_t* q0 = (int32_t*)malloc(cap * 12);
    int32_t* q1 = (int32_t*)malloc(cap * 12);
    Leafs L = {woop, tidx, inWoop, sorted, 0, 0};

    /* HLBVHBuilder.cpp:548-567 + buildBottomLevel :319-361 */
    int32_t lvlNodes[64];
    int numLvls = 0;
    lvlNodes[numLvls++] = 1;
    int64_t nodeWritten = 1, nodeCreated = 1;
    q0[0] = 0; q0[1] = 0; q0[2] = n;
    const int n_bits = 30;
    int level = 0;
    while (level < n_bits && nodeCreated > 0) {
        const int kernelLevel = n_bits - (level + 1);
        int64_t outCount = 0;
        for (int64_t e = 0; e < nodeCreated; e++) {
            /* emitTreeKernel (emitTreeKernel.cu:233-381) for queue entry e */
            int32_t nIdx = q0[3 * e], nStart = q0[3 * e + 1], nEnd = q0[3 * e + 2];
            int lv = kernelLevel;
            const int oldLevel = lv;
            int32_t split;
            while (lv >= 0 && (((keys[nStart] >> lv) & 1) == ((keys[nEnd - 1] >> lv) & 1))) lv--;
            if (lv >= 0) {
                uint32_t startBit = (keys[nStart] >> lv) & 1;
                int32_t a = nStart, b = nEnd;
                for (;;) {
                    split = (a + b) >> 1;
                    uint32_t splitBit = (keys[split] >> lv) & 1;
                    if (((keys[split - 1] >> lv) & 1) != splitBit) break;
                    if (splitBit == startBit) a = split; else b = split;
                }
            } else {
                split = (nStart + nEnd) >> 1;
            }
            int32_t* nd = nodes + (size_t)nIdx * 16;
            int32_t c0, c1;
            if ((split - nStart) <= leafSize || oldLevel == 0) {
                c0 = create_leaf(&L, nStart, split);
                nd[0] = nStart; nd[1] = split;
            } else {
                int64_t outIdx = nodeWritten + outCount;
                q1[3 * outCount] = (int32_t)outIdx; q1[3 * outCount + 1] = nStart; q1[3 * outCount + 2] = split;
                c0 = (int32_t)(outIdx * 64);
                outCount++;
            }
            if ((nEnd - split) <= leafSize || oldLevel == 0) {
                c1 = create_leaf(&L, split, nEnd);
                nd[4] = split; nd[5] = nEnd;
            } else {
                int64_t outIdx = nodeWritten + outCount;
                q1[3 * outCount] = (int32_t)outIdx; q1[3 * outCount + 1] = split; q1[3 * outCount + 2] = nEnd;
                c1 = (int32_t)(outIdx * 64);
                outCount++;
            }
            nd[12] = c0; nd[13] = c1; nd[14] = lv % 3; nd[15] = 0;
        }
        nodeCreated = outCount;
        if (nodeCreated > 0) lvlNodes[numLvls++] = (int32_t)nodeCreated;
        nodeWritten += nodeCreated;
        int32_t* t = q0; q0 = q1; q1 = t;
        level++;
    }

    /* calcAABB (HLBVHBuilder.cpp:408-449, emitTreeKernel.cu:417-562): deepest level first */
    int64_t nw = nodeWritten;
    for (int lvl = numLvls - 1; lvl >= 0; lvl--) {
        nw -= lvlNodes[lvl];
        for (int64_t qid = 0; qid < lvlNodes[lvl]; qid++) {
            int32_t* ni = nodes + (size_t)(nw + qid) * 16;
            float* nf = (float*)ni;
            int32_t ch[2] = {ni[12], ni[13]};
            float box[2][6]; /* lo.x hi.x lo.y hi.y lo.z hi.z */
            for (int k = 0; k < 2; k++) {
                if (ch[k] < 0) {
                    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                    calc_leaf(tri, pos, sorted, ni[4 * k + 0], ni[4 * k + 1], epsilon, lo, hi);
                    box[k][0] = lo[0]; box[k][1] = hi[0]; box[k][2] = lo[1]; box[k][3] = hi[1]; box[k][4] = lo[2]; box[k][5] = hi[2];
                } else {
                    const float* cn = (const float*)(nodes + (size_t)(ch[k] / 64) * 16);
                    /* minmax2(childNode[0], childNode[1]) ; min/max of childNode[2] pairs */
                    box[k][0] = cu_fminf(cn[0], cn[4]); box[k][1] = cu_fmaxf(cn[1], cn[5]);
                    box[k][2] = cu_fminf(cn[2], cn[6]); box[k][3] = cu_fmaxf(cn[3], cn[7]);
                    box[k][4] = cu_fminf(cn[8], cn[10]); box[k][5] = cu_fmaxf(cn[9], cn[11]);
                }
            }
            nf[0] = box[0][0]; nf[1] = box[0][1]; nf[2] = box[0][2]; nf[3] = box[0][3];
            nf[4] = box[1][0]; nf[5] = box[1][1]; nf[6] = box[1][2]; nf[7] = box[1][3];
            nf[8] = box[0][4]; nf[9] = box[0][5]; nf[10] = box[1][4]; nf[11] = box[1][5];
        }
    }

    out->nodes = nodes; out->nodesBytes = nodeWritten * 64;
    out->woop = woop; out->woopBytes = ((int64_t)n * 3 + L.numLeafs) * 16;
    out->triIndex = tidx; out->triIndexBytes = ((int64_t)n * 3 + L.numLeafs) * 4;
    out->mortonSorted = keys; out->triSorted = sorted;
    out->numInner = (int32_t)nodeWritten; out->numLeaves = (int32_t)L.numLeafs; out->numLevels = numLvls;
    free(inWoop); free(q0); free(q1);
    return 0;
}

void orc_lbvh_free(OrcLbvh* b)
{
    free(b->nodes); free(b->woop); free(b->triIndex); free(b->mortonSorted); free(b->triSorted);
    memset(b, 0, sizeof(*b));
}

/* ---- canonical (numbering-independent) signature of a Compact BVH ------------------------------ */
static inline uint64_t mix64(uint64_t h, uint64_t v)
{
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    return h ^ (h >> 29);
}

/* NaN payload and sign are properties of the machine that produced the NaN (x86 SSE, NVIDIA and AMD
 * each have their own default NaN), not of the algorithm: every NaN hashes alike.  NaNs appear in Woop
 * rows of degenerate triangles and where intermediate products overflow or underflow. */
static inline uint32_t canon_nan(uint32_t u)
{
    return ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) ? 0x7FC00000u : u;
}

typedef struct HashFrame { int32_t node; int stage; uint64_t acc; } HashFrame;

static uint64_t hash_leaf(const uint8_t* woop, const int32_t* triIndex, int32_t child, int hashWoop)
{
    uint64_t h = 0x1234567ull;
    for (int64_t a = ~child;; a += 3) {
        const uint32_t* w = (const uint32_t*)(woop + (size_t)a * 16);
        if (w[0] == 0x80000000u) break;
        h = mix64(h, (uint64_t)(uint32_t)triIndex[a]);
        if (hashWoop)
            for (int k = 0; k < 12; k++) h = mix64(h, canon_nan(w[k]));
    }
    return h;
}

uint64_t orc_bvh_canonical_hash(const void* nodesv, int64_t nodesBytes, const void* woopv,
                                const int32_t* triIndex, int32_t hashWoop)
{
    const uint8_t* nodes = (const uint8_t*)nodesv;
    const uint8_t* woop = (const uint8_t*)woopv;
    int64_t cap = nodesBytes / 64 + 2;
    HashFrame* st = (HashFrame*)malloc((size_t)cap * sizeof(HashFrame));
    uint64_t* ret = (uint64_t*)malloc((size_t)cap * sizeof(uint64_t));
    int sp = 0, rp = 0;
    st[sp++] = (HashFrame){0, 0, 0};
    /* post-order: hash(node) = mix(boxes, hash(child0), hash(child1)) */
    while (sp > 0) {
        HashFrame* f = &st[sp - 1];
        const uint32_t* n = (const uint32_t*)(nodes + (size_t)f->node);
        const int32_t c[2] = {(int32_t)n[12], (int32_t)n[13]};
        if (f->stage < 2) {
            int k = f->stage++;
            if (c[k] < 0) ret[rp++] = hash_leaf(woop, triIndex, c[k], hashWoop);
            else st[sp++] = (HashFrame){c[k], 0, 0};
            continue;
        }
        uint64_t h = 0xABCDEFull;
        for (int k = 0; k < 12; k++) h = mix64(h, canon_nan(n[k]));
        h = mix64(h, n[14]); /* split-axis word */
        uint64_t h1 = ret[--rp], h0 = ret[--rp];
        h = mix64(mix64(h, h0), h1);
        ret[rp++] = h;
        sp--;
    }
    uint64_t r = ret[0];
    free(st); free(ret);
    return r;
}
