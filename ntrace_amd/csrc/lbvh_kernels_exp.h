// lbvh_kernels_exp.h -- A/B scaffolding of rounds 1-2, compiled ONLY with -DNTR_EXPERIMENTS into libntrace_amd_exp.so
// (tests/test_lbvh_gpu.py runs these build paths against it; the shipped library has none of it).  Included by lbvh_kernels.hip
// at the places the sections depend on, one section per inclusion (NTR_LBVH_EXP_SECTION):
//   1  lbvh_morton_kernel         round-1 Morton pass (per-level path)
//   2  lbvh_woop_kernel           round-1 Woop pass in mesh order
//   3  lbvh_emit_kernel / lbvh_refit_kernel: one launch per level, as the reference (emitTreeKernel.cu:233-381, 417-562)
//   4  lbvh_top_cells_kernel      top of the tree from a 14-bit cell table (measured slower than the bottom-up emit)
//   5  lbvh_top_cells_refit_kernel
//   6  host side of the per-level path (lbvh_levelsync_emit_refit)
// No include guard: every inclusion selects one section.

#if NTR_LBVH_EXP_SECTION == 1
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

#endif  // section 1

#if NTR_LBVH_EXP_SECTION == 2
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

#endif  // section 2

#if NTR_LBVH_EXP_SECTION == 3
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


#endif  // section 3

#if NTR_LBVH_EXP_SECTION == 4
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

#endif  // section 4

#if NTR_LBVH_EXP_SECTION == 5
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

#endif  // section 5
#if NTR_LBVH_EXP_SECTION == 6
// Host side of the per-level path: one emit launch per level, one read-back, one refit launch per level (HLBVHBuilder.cpp:337-361, 427-439).
static int lbvh_levelsync_emit_refit(hipStream_t s, int n, int leafSize, float epsilon, LbvhState* state, LbvhState& h, const unsigned int* keys,
                                     const int* triSorted, char* ws, size_t oWoop, size_t oQ0, size_t oQ1, void* d_nodes, unsigned int nodeCap,
                                     void* d_triWoop, int32_t* d_triIndex, const int32_t* d_triVtxIndex, const float* d_vtxPos, PhaseEvents& pe)
{
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
    return NTR_OK;
}
#endif  // section 6
