// lbvh_topdown.h -- the top-down fallback of the LBVH builder: scenes of at most NTR_LBVH_SPLIT (3 072) triangles (one workgroup builds
// the whole tree in LDS), n <= leafSize, and leaves of more than 32 triangles, which the bottom-up path of lbvh_kernels.hip does not
// take.  A level-by-level top pass with key probes (lbvh_top_kernel) hands ranges of at most `spill` triangles to subtree workgroups
// (lbvh_subtree_kernel: emit + refit with workgroup barriers only).  Same split and leaf rules as the reference
// (emitTreeKernel.cu:233-381).  Included by lbvh_kernels.hip.

// After the sort: box terms in sorted order (one 24-byte gather per triangle).
__global__ __launch_bounds__(256) void lbvh_gather_box_kernel(int n, const int* __restrict__ triSorted, const float2* __restrict__ boxMesh,
                                                              float2* __restrict__ triBox)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int t = triSorted[j];
    const float2 a = boxMesh[3 * (size_t)t], b = boxMesh[3 * (size_t)t + 1], c = boxMesh[3 * (size_t)t + 2];
    triBox[3 * (size_t)j] = a; triBox[3 * (size_t)j + 1] = b; triBox[3 * (size_t)j + 2] = c;
}

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

