// radix_sort.h -- hand-written LSD radix sort pass for gfx950 (8 bits per pass), shared by the LBVH builder (sort_by_key of Morton
// codes, radixSort.cu:22-50) and the ray sort (192-bit keys, RayBuffer::mortonSort, src/rt/ray/RayBuffer.cpp:103-165).
#pragma once
#include <hip/hip_runtime.h>

// ------------------------------------------------------------------------------------------------------------
// One-sweep LSD pass ("onesweep": chained scan with decoupled look-back): ONE launch per 8-bit digit instead of
// histogram + scan + scatter.  The digit histograms of ALL passes are taken in one earlier read of the keys
// (lbvh_morton_kernel accumulates them while it produces the keys), so a pass reads every key once and writes it
// once.
//
//   * workgroups take tiles in ticket order (one returning atomic), so every predecessor of a tile is resident
//     or finished: waiting on predecessors cannot deadlock -- unless the whole grid fits the device at once (onesweep_launch
//     asks the runtime), where every tile becomes resident whatever the dispatch order and workgroup b simply takes tile b: the
//     one counter serves ~88 atomics per microsecond, 2-5 us of a tile's life at 128-512 tiles (scripts/studies/onesweep_timeline.py).
//     (Another kernel holding part of the device only delays that: a grid's workgroups are dispatched in index order, so the tiles
//     that are resident are always a prefix of the pass and wait for nothing that is not; and a wait that never ends raises *errFlag.)
//   * per tile and digit ONE 64-bit word carries status, reach and count together (8 status bits tagged with the pass, 24 bits
//     "lowest tile the count covers", 32 count bits), published and polled with agent-scope atomics: no ordering against other
//     memory is needed, and the array is cleared once per sort, not per pass;
//   * the look-back JUMPS (round 6): a tile that has summed its predecessors back to tile r publishes "count of tiles r..t" as it
//     goes, and whoever reads that word continues at r - 1, so the reach doubles per round trip.  ONE word per round trip then does
//     what eight plain aggregates did (pass times within 2 % at 262 k / 2.8 M / 10 M keys) with an eighth of the polling, and polling
//     is not free: the state words travel at agent scope, and with ~450 tiles in flight every further word per round trip is
//     requests the fabric has to serve (10 M keys: 68 / 70 / 96 / 101 us per pass with 1 / 2 / 4 / 8 words per round trip).
//     What a pass costs is a tile's life times tiles over tiles in flight (scripts/studies/onesweep_timeline.py: 10 M keys, 8 192-key
//     tiles: 16 us = ticket and digit scan 1.6, load and rank 5.9, publish 0.7, look-back 4.5, stage 1.8, write 1.8) -- latency, which
//     neither the jumps nor cheaper ranking shorten much;
//   * ranking inside the tile: the lanes of a 64-key round that hold the same digit find each other through an LDS word per digit
//     (atomic OR of the lane bits, see the kernel); keys are then staged
//     in LDS in tile-sorted order and written out by consecutive threads, so that every digit's run leaves as
//     one contiguous store stream;
//   * every spin is bounded (a poll that never succeeds raises *errFlag instead of hanging the device).
// Stable; n < 2^28.
// ------------------------------------------------------------------------------------------------------------
static constexpr int OS_THREADS = 256;
static constexpr unsigned int OS_SPIN_LIMIT = 1u << 22;
static constexpr int OS_MAX_PASSES = 126;        // 8 status bits: pass p publishes 2p + 1 (partial) and 2p + 2 (inclusive); 0 = nothing yet
static constexpr unsigned int OS_REACH_MASK = 0x00FFFFFFu;

// state word of (tile, digit): status << 56 | reach << 32 | count, where count = keys of this digit in tiles reach .. tile
__device__ __forceinline__ unsigned int os_tag(int pass, bool inclusive) { return (unsigned int)(pass * 2 + (inclusive ? 2 : 1)); }
__device__ __forceinline__ unsigned long long os_word(unsigned int tag, unsigned int reach, unsigned int count)
{
    return ((unsigned long long)tag << 56) | ((unsigned long long)(reach & OS_REACH_MASK) << 32) | (unsigned long long)count;
}

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

// MODE 0: keysIn[i] is the key of element i; keys and values both move (valsIn == nullptr: the value of element i is i -- a sort's first
//         pass needs no index array written and read back).
// MODE 1: the key word of element i is keysIn[valsIn[i] * stride] (multi-word keys stay in place, only the index array moves).
// MODE 2: the key word is fetched as in MODE 1 and moves with the value from here on (first pass over a word of a multi-word key;
//         the following passes over that word run in MODE 0 on the dense (word, value) pairs).
// SKIP_TRIVIAL (the ray sort's instantiations): a pass whose digit is the same in EVERY key -- the upper digits of a 192-bit ray key over a
// batch that fills a corner of the scene: 8-11 of its 19 passes -- is the identity permutation; its tiles then only copy (and, MODE 2, fetch
// the new key word): no ranking, no chained scan.
template <int ITEMS, int MODE, bool SKIP_TRIVIAL = false, int OS_LOOK = 1>
__global__ __launch_bounds__(OS_THREADS) static void onesweep_pass_kernel(int n, const unsigned int* __restrict__ keysIn,
                                                                           const int* __restrict__ valsIn,
                                                                           unsigned int* __restrict__ keysOut, int* __restrict__ valsOut,
                                                                           int stride, int shift, int pass,
                                                                           const unsigned int* __restrict__ digitTotals /* [256] of this pass */,
                                                                           unsigned long long* tileState /* [tiles][256] */,
                                                                           unsigned int* ticket, unsigned int* errFlag)
{
    constexpr int WAVES = OS_THREADS / 64;
    constexpr int TILE = OS_THREADS * ITEMS;
    __shared__ unsigned int s_cnt[WAVES][256];   // per wave and digit: keys ranked so far; later the wave's first tile-local position of the digit
    __shared__ unsigned int s_wt[2][WAVES];
    __shared__ unsigned int s_dst[256];          // global position of tile-local position 0 of digit d's run, minus the run's tile-local start
    __shared__ __attribute__((aligned(8))) unsigned int s_keys[TILE];
    __shared__ int s_vals[TILE];
    __shared__ unsigned int s_tile;
    // per wave and digit: the lanes of the current round that hold the digit (two alternating arrays when the staging area -- unused
    // until the ranks are known -- has room for them)
    constexpr int MATCH_BUFS = ITEMS >= 16 ? 2 : 1;
    static_assert(sizeof(unsigned long long) * WAVES * MATCH_BUFS * 256 <= sizeof(unsigned int) * TILE, "the match words live in the key staging area");
    unsigned long long (*s_match)[MATCH_BUFS][256] = reinterpret_cast<unsigned long long (*)[MATCH_BUFS][256]>(s_keys);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid == 0) s_tile = ticket ? atomicAdd(ticket, 1u) : blockIdx.x;   // ticket == nullptr: every tile of the grid is resident at once
    for (int i = tid; i < WAVES * 256; i += OS_THREADS) (&s_cnt[0][0])[i] = 0;
    for (int i = tid; i < WAVES * MATCH_BUFS * 256; i += OS_THREADS) (&s_match[0][0][0])[i] = 0ull;
    // global base of digit `tid`: exclusive scan of the digit totals
    const unsigned int digitBase = os_excl_scan_256(digitTotals[tid], s_wt[0]);   // (its barrier also publishes s_tile, s_cnt and s_match)
    const unsigned int tile = s_tile;

    const long long chunk = (long long)tile * TILE + wave * (64 * ITEMS);
    if (SKIP_TRIVIAL && __syncthreads_or(digitTotals[tid] == (unsigned int)n)) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const long long k = chunk + r * 64 + lane;
            if (k < n) {
                const int v = (MODE == 0 && !valsIn) ? (int)k : valsIn[k];
                if (MODE != 1) keysOut[k] = MODE != 0 ? keysIn[(size_t)v * stride] : keysIn[k];
                valsOut[k] = v;
            }
        }
        return;
    }
    unsigned int key[ITEMS], rank[ITEMS];
    int val[ITEMS];
    const unsigned long long laneBit = 1ull << lane, ltMask = laneBit - 1ull;
    // how many of the wave's 64-key rounds hold keys at all (the last tile of a pass is ragged)
    const long long left = (long long)n - chunk;
    const int myCount = left >= 64 * ITEMS ? 64 * ITEMS : (left > 0 ? (int)left : 0);
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const bool valid = r * 64 + lane < myCount;
        const long long k = chunk + r * 64 + lane;
        val[r] = valid ? (MODE == 0 && !valsIn ? (int)k : valsIn[k]) : 0;   // MODE 0 without values: element k carries k
        key[r] = valid ? (MODE != 0 ? keysIn[(size_t)val[r] * stride] : keysIn[k]) : 0xFFFFFFFFu;
    }
    // Stable ranking of a round's 64 keys by their digit (round 6): every lane ORs its lane bit into the LDS word of its digit -- an
    // atomic OR commutes, so the word a lane reads back with the wave's next LDS instruction (a wave's LDS operations execute in order)
    // is the set of lanes holding the same digit whatever order the hardware took the lanes in -- and its rank is the digit's count of
    // the earlier rounds plus the set's lanes below it.  The first lane of a set clears the word and advances the count.  (Until round 6
    // the set came from eight ballots and a 64-bit select-and-AND per bit: ~60 vector instructions per round against ~15.)  Two word
    // arrays alternate so that a round's OR never waits for the previous round's clear.
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const bool valid = r * 64 + lane < myCount;
        const unsigned int d = (key[r] >> shift) & 255;
        unsigned long long* m = &s_match[wave][r % MATCH_BUFS][d];
        if (valid) __hip_atomic_fetch_or(m, laneBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned long long peers = __hip_atomic_load(m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned int before = s_cnt[wave][d];
        rank[r] = before + __popcll(peers & ltMask);
        if (valid && (peers & ltMask) == 0ull) {
            s_cnt[wave][d] = before + __popcll(peers);
            __hip_atomic_store(m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();

    // digit `tid`: the tile's count
    unsigned int cnt = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) cnt += s_cnt[w][tid];
    unsigned long long* myState = tileState + (size_t)tile * 256 + tid;
    const unsigned int tagPart = os_tag(pass, false), tagInc = os_tag(pass, true);
    __hip_atomic_store(myState, os_word(tile == 0 ? tagInc : tagPart, tile, cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // tile-local start of every digit
    const unsigned int tileStart = os_excl_scan_256(cnt, s_wt[1]);
    {   // the waves' first tile-local positions of the digit
        unsigned int acc = tileStart;
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            const unsigned int c = s_cnt[w][tid];
            s_cnt[w][tid] = acc;
            acc += c;
        }
    }

    // stage in tile-sorted order -- BEFORE the look-back: the positions inside the tile are known, and while this tile moves its keys
    // its predecessors get on with their own look-backs (what it then reads from them reaches further back)
    __syncthreads();   // the waves' digit positions (s_cnt) are complete; the match words in the staging area are dead
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (r * 64 + lane < myCount) {
            const unsigned int d = (key[r] >> shift) & 255;
            const unsigned int pos = s_cnt[wave][d] + rank[r];
            s_keys[pos] = key[r];
            s_vals[pos] = val[r];
        }
    }

    // decoupled look-back over the predecessors' words of this digit: OS_LOOK words per round trip (independent loads), consumed nearest
    // tile first.  A word says how many keys of the digit tiles reach .. t' hold: it is added and the walk continues at reach - 1 (inside
    // the loaded window, or with the next round trip); an inclusive word ends it, an unpublished one is read again.  After every round
    // trip that made progress the tile publishes what IT covers by now, so that its successors jump over all of it.
    unsigned int excl = 0;
    if (tile > 0) {
        int cur = (int)tile - 1;
        unsigned int spins = 0;
        bool done = false;
        while (!done && cur >= 0) {
            unsigned long long w[OS_LOOK];
#pragma unroll
            for (int j = 0; j < OS_LOOK; j++)
                w[j] = (cur - j) >= 0 ? __hip_atomic_load(tileState + (size_t)(cur - j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            int c = cur;
#pragma unroll
            for (int j = 0; j < OS_LOOK; j++) {
                if (done || (cur - j) != c || c < 0) continue;
                const unsigned int tag = (unsigned int)(w[j] >> 56);
                if (tag != tagPart && tag != tagInc) continue;   // not published yet: nothing behind it can be consumed either
                excl += (unsigned int)w[j];
                if (tag == tagInc) done = true;
                else c = (int)((unsigned int)(w[j] >> 32) & OS_REACH_MASK) - 1;
            }
            if (c == cur) {
                if (++spins > OS_SPIN_LIMIT) { atomicOr(errFlag, 2u); break; }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            cur = c;
            if (!done && cur >= 0)
                __hip_atomic_store(myState, os_word(tagPart, (unsigned int)(cur + 1), excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(myState, os_word(tagInc, 0u, excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_dst[tid] = digitBase + excl - tileStart;
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

// One pass on `s`.  `ticket` is only used when the grid does not fit the device at once (see the kernel's header).
template <int ITEMS, int MODE, bool SKIP_TRIVIAL, int OS_LOOK = 1>
static inline void onesweep_launch(hipStream_t s, int tiles, int n, const unsigned int* keysIn, const int* valsIn, unsigned int* keysOut, int* valsOut,
                                   int stride, int shift, int pass, const unsigned int* digitTotals, unsigned long long* tileState,
                                   unsigned int* ticket, unsigned int* errFlag)
{
    // tiles the device holds at once: per instantiation, asked once (one device type per process; the initialisation is thread-safe)
    static const int residentTiles = [] {
        int perCU = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, onesweep_pass_kernel<ITEMS, MODE, SKIP_TRIVIAL, OS_LOOK>, OS_THREADS, 0) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return 0;
        return perCU * cus;
    }();
    hipLaunchKernelGGL((onesweep_pass_kernel<ITEMS, MODE, SKIP_TRIVIAL, OS_LOOK>), dim3(tiles), dim3(OS_THREADS), 0, s, n, keysIn, valsIn, keysOut, valsOut, stride,
                       shift, pass, digitTotals, tileState, tiles <= residentTiles ? (unsigned int*)nullptr : ticket, errFlag);
}
