// radix_sort.h -- hand-written LSD radix sort building blocks for gfx950 (8 bits per pass), shared by
// the LBVH builder (sort_by_key of Morton codes, radixSort.cu:22-50) and the ray sort (192-bit keys,
// RayBuffer::mortonSort, src/rt/ray/RayBuffer.cpp:103-165).
//   sort_hist_kernel     per-tile digit histogram in LDS -> hist[digit][tile]
//   sort_scan_rows_kernel exclusive scan of the histogram array, one workgroup per digit
//   sort_scatter_kernel  stable scatter: each wave ranks 64 keys per round with 8 ballots
//                        (match-any) + prefix popcount; rounds chain through per-wave LDS counters
// INDEXED = false: keys[i] is the key of element i, keys and values both move.
// INDEXED = true : the key word of element i is keyBase[vals[i] * stride] (multi-word keys stay
//                  in place, only the index array moves).
#pragma once
#include <hip/hip_runtime.h>

namespace ntr {

static constexpr int SORT_THREADS = 256;
static constexpr int SORT_ITEMS = 8;
static constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // keys per workgroup


}  // namespace ntr

// ------------------------------------------------------------------------------------------------------------
// One-sweep LSD pass ("onesweep": chained scan with decoupled look-back): ONE launch per 8-bit digit instead of
// histogram + scan + scatter.  The digit histograms of ALL passes are taken in one earlier read of the keys
// (lbvh_morton_kernel accumulates them while it produces the keys), so a pass reads every key once and writes it
// once.
//
//   * workgroups take tiles in ticket order (one returning atomic), so every predecessor of a tile is resident
//     or finished: waiting on predecessors cannot deadlock;
//   * per tile and digit ONE 32-bit word carries status and count together (4 status bits tagged with the pass,
//     28 count bits), published and polled with agent-scope atomics: no ordering against other memory is needed,
//     and the array is cleared once per sort, not per pass;
//   * ranking inside the tile is the stable wave64 match-any ranking of sort_scatter_kernel; keys are then staged
//     in LDS in tile-sorted order and written out by consecutive threads, so that every digit's run leaves as
//     one contiguous store stream;
//   * every spin is bounded (a poll that never succeeds raises *errFlag instead of hanging the device).
// Stable; n < 2^28.
// ------------------------------------------------------------------------------------------------------------
static constexpr int OS_THREADS = 256;
static constexpr unsigned int OS_COUNT_MASK = 0x0FFFFFFFu;
static constexpr unsigned int OS_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ unsigned int os_status(int pass, bool inclusive) { return (unsigned int)(pass * 2 + (inclusive ? 2 : 1)) << 28; }

// exclusive prefix over the 256 threads of a workgroup (one value each): wave scans by lane shuffles, one barrier
__device__ __forceinline__ unsigned int os_excl_scan_256(unsigned int v, unsigned int* s_waveTotals /* [4] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int u = (unsigned int)__shfl_up((int)incl, off);
        if (lane >= off) incl += u;
    }
    if (lane == 63) s_waveTotals[wave] = incl;
    __syncthreads();
    unsigned int before = 0;
#pragma unroll
    for (int w = 0; w < 3; w++)
        if (w < wave) before += s_waveTotals[w];
    return before + incl - v;
}

// MODE 0: keysIn[i] is the key of element i; keys and values both move.
// MODE 1: the key word of element i is keysIn[valsIn[i] * stride] (multi-word keys stay in place, only the index array moves).
// MODE 2: the key word is fetched as in MODE 1 and moves with the value from here on (first pass over a word of a multi-word key;
//         the following passes over that word run in MODE 0 on the dense (word, value) pairs).
// SKIP_TRIVIAL (the ray sort's instantiations): a pass whose digit is the same in EVERY key -- the upper digits of a 192-bit ray key over a
// batch that fills a corner of the scene: 8-11 of its 19 passes -- is the identity permutation; its tiles then only copy (and, MODE 2, fetch
// the new key word): no ranking, no chained scan.
template <int ITEMS, int MODE, bool SKIP_TRIVIAL = false>
__global__ __launch_bounds__(OS_THREADS) static void onesweep_pass_kernel(int n, const unsigned int* __restrict__ keysIn,
                                                                           const int* __restrict__ valsIn,
                                                                           unsigned int* __restrict__ keysOut, int* __restrict__ valsOut,
                                                                           int stride, int shift, int pass,
                                                                           const unsigned int* __restrict__ digitTotals /* [256] of this pass */,
                                                                           unsigned int* tileState /* [tiles][256] */,
                                                                           unsigned int* ticket, unsigned int* errFlag)
{
    constexpr int WAVES = OS_THREADS / 64;
    constexpr int TILE = OS_THREADS * ITEMS;
    __shared__ unsigned int s_cnt[WAVES][256];   // per wave and digit: keys ranked so far; later the wave's offset inside the digit
    __shared__ unsigned int s_wt[2][WAVES];
    __shared__ unsigned int s_tileStart[256];    // first tile-local position of digit d
    __shared__ unsigned int s_dst[256];          // global position of tile-local position 0 of digit d's run, minus s_tileStart[d]
    __shared__ unsigned int s_keys[TILE];
    __shared__ int s_vals[TILE];
    __shared__ unsigned int s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < WAVES * 256; i += OS_THREADS) (&s_cnt[0][0])[i] = 0;
    // global base of digit `tid`: exclusive scan of the digit totals
    const unsigned int digitBase = os_excl_scan_256(digitTotals[tid], s_wt[0]);   // (its barrier also publishes s_tile and s_cnt)
    const unsigned int tile = s_tile;

    const long long chunk = (long long)tile * TILE + wave * (64 * ITEMS);
    if (SKIP_TRIVIAL && __syncthreads_or(digitTotals[tid] == (unsigned int)n)) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const long long k = chunk + r * 64 + lane;
            if (k < n) {
                const int v = valsIn[k];
                if (MODE != 1) keysOut[k] = MODE != 0 ? keysIn[(size_t)v * stride] : keysIn[k];
                valsOut[k] = v;
            }
        }
        return;
    }
    unsigned int key[ITEMS], rank[ITEMS];
    int val[ITEMS];
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const long long k = chunk + r * 64 + lane;
        const bool valid = k < n;
        val[r] = valid ? valsIn[k] : 0;
        key[r] = valid ? (MODE != 0 ? keysIn[(size_t)val[r] * stride] : keysIn[k]) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const bool valid = (chunk + r * 64 + lane) < n;
        const unsigned int d = (key[r] >> shift) & 255;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const unsigned int before = s_cnt[wave][d];
        rank[r] = before + __popcll(peers & ltMask);
        if (valid && (peers & ltMask) == 0ull) s_cnt[wave][d] = before + __popcll(peers);
    }
    __syncthreads();

    // digit `tid`: the tile's count, the waves' offsets inside the digit
    unsigned int cnt = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
        const unsigned int c = s_cnt[w][tid];
        s_cnt[w][tid] = cnt;
        cnt += c;
    }
    unsigned int* myState = tileState + (size_t)tile * 256 + tid;
    __hip_atomic_store(myState, (tile == 0 ? os_status(pass, true) : os_status(pass, false)) | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // tile-local start of every digit
    const unsigned int tileStart = os_excl_scan_256(cnt, s_wt[1]);
    s_tileStart[tid] = tileStart;

    // decoupled look-back over the predecessors' words of this digit, OS_LOOK of them per round trip (the loads of a round are
    // independent; they are consumed nearest tile first, up to the first inclusive word, or re-read from the first unpublished one)
    unsigned int excl = 0;
    if (tile > 0) {
        constexpr int OS_LOOK = 8;   // (32 words per round trip for the latency-bound 262 k-key sort: 52 -> 64.5 us for the four passes, round 6)
        const unsigned int stAgg = os_status(pass, false) >> 28, stInc = os_status(pass, true) >> 28;
        int t = (int)tile - 1;
        unsigned int spins = 0;
        bool done = false;
        while (!done) {
            unsigned int w[OS_LOOK];
#pragma unroll
            for (int j = 0; j < OS_LOOK; j++)
                w[j] = (t - j) >= 0 ? __hip_atomic_load(tileState + (size_t)(t - j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : (stInc << 28);  // before the first tile: nothing
            int used = 0;
#pragma unroll
            for (int j = 0; j < OS_LOOK; j++) {
                if (done || used != j) continue;
                const unsigned int st = w[j] >> 28;
                if (st != stAgg && st != stInc) continue;   // not published yet: stop consuming here
                excl += w[j] & OS_COUNT_MASK;
                used = j + 1;
                if (st == stInc) done = true;
            }
            t -= used;
            if (!done && used < OS_LOOK) {
                if (++spins > OS_SPIN_LIMIT) { atomicOr(errFlag, 2u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __hip_atomic_store(myState, os_status(pass, true) | ((excl + cnt) & OS_COUNT_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_dst[tid] = digitBase + excl - tileStart;
    __syncthreads();

    // stage in tile-sorted order, then stream out
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if ((chunk + r * 64 + lane) < n) {
            const unsigned int d = (key[r] >> shift) & 255;
            const unsigned int pos = s_tileStart[d] + s_cnt[wave][d] + rank[r];
            s_keys[pos] = key[r];
            s_vals[pos] = val[r];
        }
    }
    __syncthreads();
    const long long tileBeg = (long long)tile * TILE;
    const int tileCount = (int)((n - tileBeg) < (long long)TILE ? (n - tileBeg) : (long long)TILE);
    for (int i = tid; i < tileCount; i += OS_THREADS) {
        const unsigned int k = s_keys[i];
        const unsigned int dst = s_dst[(k >> shift) & 255] + (unsigned int)i;
        if (MODE != 1) keysOut[dst] = k;
        valsOut[dst] = s_vals[i];
    }
}
