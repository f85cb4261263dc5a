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

template <bool INDEXED>
__device__ __forceinline__ unsigned int sort_key(const unsigned int* __restrict__ keys, const int* __restrict__ vals, int stride, int k)
{
    return INDEXED ? keys[(size_t)vals[k] * stride] : keys[k];
}

template <bool INDEXED>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(int n, const unsigned int* __restrict__ keys,
                                                                 const int* __restrict__ vals, int stride, int shift,
                                                                 unsigned int* __restrict__ hist, int numBlocks)
{
    __shared__ unsigned int s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * SORT_TILE;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; i++) {
        const int k = base + i * SORT_THREADS + threadIdx.x;
        if (k < n) atomicAdd(&s_hist[(sort_key<INDEXED>(keys, vals, stride, k) >> shift) & 255], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * numBlocks + blockIdx.x] = s_hist[threadIdx.x];  // digit-major
}

// Scan of the digit-major histogram hist[256][numBlocks] in two small parallel steps:
//   sort_scan_rows_kernel    one workgroup per digit: exclusive prefix over the tiles, row total out
// The exclusive scan of the 256 row totals (global base of every digit) is recomputed by every scatter
// workgroup in LDS -- 256 words, cheaper than a launch of its own.
__global__ __launch_bounds__(256) static void sort_scan_rows_kernel(unsigned int* __restrict__ hist, int numBlocks,
                                                                    unsigned int* __restrict__ rowTotal)
{
    __shared__ unsigned int s_part[256];
    unsigned int* row = hist + (size_t)blockIdx.x * numBlocks;
    const int per = (numBlocks + 255) / 256;
    const int beg = min((int)threadIdx.x * per, numBlocks), end = min(beg + per, numBlocks);
    unsigned int sum = 0;
    for (int i = beg; i < end; i++) sum += row[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned int v = (threadIdx.x >= (unsigned)off) ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned int run = s_part[threadIdx.x] - sum;
    for (int i = beg; i < end; i++) {
        const unsigned int v = row[i];
        row[i] = run;
        run += v;
    }
    if (threadIdx.x == 255) rowTotal[blockIdx.x] = s_part[255];
}

template <bool INDEXED>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(int n, const unsigned int* __restrict__ keysIn,
                                                                    const int* __restrict__ valsIn,
                                                                    unsigned int* __restrict__ keysOut, int* __restrict__ valsOut,
                                                                    int stride, int shift, const unsigned int* __restrict__ hist,
                                                                    const unsigned int* __restrict__ rowTotal, int numBlocks)
{
    constexpr int WAVES = SORT_THREADS / 64;
    static_assert(SORT_THREADS == 256, "one thread per digit");
    __shared__ unsigned int s_cnt[WAVES][256];
    __shared__ unsigned int s_base[WAVES][256];
    __shared__ unsigned int s_digit[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < WAVES * 256; i += SORT_THREADS) (&s_cnt[0][0])[i] = 0;
    // exclusive scan of the digit totals: global base of digit threadIdx.x
    const unsigned int myTotal = rowTotal[threadIdx.x];
    s_digit[threadIdx.x] = myTotal;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned int v = (threadIdx.x >= (unsigned)off) ? s_digit[threadIdx.x - off] : 0u;
        __syncthreads();
        s_digit[threadIdx.x] += v;
        __syncthreads();
    }
    const unsigned int digitBase = s_digit[threadIdx.x] - myTotal;

    const int chunk = blockIdx.x * SORT_TILE + wave * (64 * SORT_ITEMS);
    unsigned int key[SORT_ITEMS], rank[SORT_ITEMS];
    int val[SORT_ITEMS];
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const int k = chunk + r * 64 + lane;
        const bool valid = k < n;
        val[r] = valid ? valsIn[k] : 0;
        key[r] = valid ? (INDEXED ? keysIn[(size_t)val[r] * stride] : keysIn[k]) : 0xFFFFFFFFu;
        const unsigned int d = (key[r] >> shift) & 255;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const unsigned int before = s_cnt[wave][d];  // same address for all peers (broadcast)
        rank[r] = before + __popcll(peers & ltMask);
        if (valid && (peers & ltMask) == 0ull) s_cnt[wave][d] = before + __popcll(peers);  // lowest peer lane
    }
    __syncthreads();
    {   // digit threadIdx.x: offsets of the waves and the global base of this tile
        const unsigned int d = threadIdx.x;
        unsigned int run = hist[d * numBlocks + blockIdx.x] + digitBase;
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            s_base[w][d] = run;
            run += s_cnt[w][d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const int k = chunk + r * 64 + lane;
        if (k < n) {
            const unsigned int d = (key[r] >> shift) & 255;
            const unsigned int dst = s_base[wave][d] + rank[r];
            if (!INDEXED) keysOut[dst] = key[r];
            valsOut[dst] = val[r];
        }
    }
}

}  // namespace ntr
