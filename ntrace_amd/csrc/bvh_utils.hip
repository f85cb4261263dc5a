// bvh_utils.hip -- small whole-buffer passes over a Compact BVH.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "ntr_internal.h"

namespace ntr {

// Flags a Compact node buffer whose 12 box floats per node (bytes 0..47 of each
// 64-B node, src/rt/cuda/CudaBVH.hpp:42-46) are all finite with |x| < 2^100
// (plus the FASTDIV / NOTINY ranges documented in include/ntrace_amd.h).
__global__ __launch_bounds__(256) void bvh_validate_kernel(const float4* __restrict__ nodes, int64_t numFloat4,
                                                           unsigned int* __restrict__ bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool notFinite = false, notFast = false, tiny = false, unordered = false;
    unsigned int leaves = 0, maxLeafOfs = 0;
    for (; i < numFloat4; i += stride) {
        const float4 v = nodes[i];
        if ((i & 3) == 3) {  // child words: a negative child is ~(float4 index of the leaf's first Woop row)
            const int c0 = __float_as_int(v.x), c1 = __float_as_int(v.y);
            if (c0 < 0) { leaves++; maxLeafOfs = max(maxLeafOfs, (unsigned int)~c0); }
            if (c1 < 0) { leaves++; maxLeafOfs = max(maxLeafOfs, (unsigned int)~c1); }
            continue;
        }
        const float c[4] = {v.x, v.y, v.z, v.w};
        unordered = unordered || !(v.x <= v.y) || !(v.z <= v.w);   // every box float4 is two (lo, hi) pairs
        for (int k = 0; k < 4; k++) {
            const float a = fabsf(c[k]);
            notFinite = notFinite || !(a < 0x1p100f);  // also true for NaN
            notFast = notFast || !(a < 0x1p55f);
            tiny = tiny || (c[k] != 0.0f && a < 0x1p-93f);
        }
    }
    const unsigned int bits = (__ballot(notFinite) != 0ull ? 1u : 0u) | (__ballot(notFast) != 0ull ? 2u : 0u) |
                              (__ballot(tiny) != 0ull ? 4u : 0u) | (__ballot(unordered) != 0ull ? 8u : 0u);
    if (bits && (threadIdx.x & 63) == 0) atomicOr(bad, bits);
    // leaf statistics (NTR_BVH_WIDE_LEAVES): number of leaves and the last leaf's offset ~ the triWoop extent
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        leaves += (unsigned int)__shfl_xor((int)leaves, off);
        maxLeafOfs = max(maxLeafOfs, (unsigned int)__shfl_xor((int)maxLeafOfs, off));
    }
    // per-workgroup partials, summed on the host (thousands of waves adding to ONE word cost more than the whole pass)
    __shared__ unsigned int s_leaves[4], s_max[4];
    if ((threadIdx.x & 63) == 0) { s_leaves[threadIdx.x >> 6] = leaves; s_max[threadIdx.x >> 6] = maxLeafOfs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bad[4 + 2 * blockIdx.x] = s_leaves[0] + s_leaves[1] + s_leaves[2] + s_leaves[3];
        bad[5 + 2 * blockIdx.x] = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    }
}

// ---- measurement aid: the memory side of a divergent traversal without its arithmetic ----------------------------------------------
// Every active lane walks a dependent chain of random 64-byte records (4 x global_load_dwordx4; the next index is a hash of the bytes
// just loaded) -- what a ray does from node to node.  The records / s of a full grid on a table larger than the Infinity Cache is the
// practical roof of an HBM-resident BVH's node and triangle fetches (scripts/microbench/chase64.hip; bench.py extras.gather_roof).
__global__ __launch_bounds__(256) void gather_fill_kernel(unsigned int* __restrict__ t, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256)
        t[i] = (unsigned int)(i * 2654435761ull) ^ (unsigned int)(i >> 7) * 40503u;
}
__global__ __launch_bounds__(64) void gather_chase_kernel(const uint4* __restrict__ table, unsigned int mask, int steps, int activeLanes, unsigned int* __restrict__ out)
{
    const int tid = blockIdx.x * 64 + threadIdx.x;
    if ((int)threadIdx.x >= activeLanes) return;
    unsigned int rec = ((unsigned)tid * 2654435761u) & mask;
    unsigned int acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4* q = table + (size_t)rec * 4;
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        const unsigned int h = a.x ^ b.y ^ c.z ^ d.w ^ a.w ^ d.x;
        acc += h;
        rec = ((h ^ (unsigned)tid * 0x9E3779B9u) * 2654435761u + (unsigned)s * 40503u) & mask;
    }
    out[tid] = acc;
}

}  // namespace ntr

extern "C" int ntr_selftest_gather_rate(int64_t tableBytes, int32_t waves, int32_t lanesPerWave, int32_t steps, void* stream, float* seconds)
{
    using namespace ntr;
    if (!seconds || tableBytes < 4096 || tableBytes > ((int64_t)1 << 34) || waves < 1 || waves > (1 << 20) || lanesPerWave < 1 || lanesPerWave > 64 || steps < 1)
        return set_error(NTR_ERR_INVALID, "ntr_selftest_gather_rate: bad argument");
    *seconds = 0.0f;
    hipStream_t s = (hipStream_t)stream;
    size_t recs = (size_t)tableBytes / 64, pow2 = 1;
    while (pow2 * 2 <= recs) pow2 *= 2;
    void* d_t = nullptr;
    unsigned int* d_o = nullptr;
    NTR_HIP(hipMalloc(&d_t, (size_t)tableBytes));
    if (hipMalloc((void**)&d_o, (size_t)waves * 64 * sizeof(unsigned int)) != hipSuccess) { (void)hipFree(d_t); return set_error(NTR_ERR_NOMEM, "ntr_selftest_gather_rate: out of device memory"); }
    hipLaunchKernelGGL(gather_fill_kernel, dim3(4096), dim3(256), 0, s, (unsigned int*)d_t, (size_t)tableBytes / 4);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3 && err == hipSuccess; rep++) {
        err = hipEventRecord(e0, s);
        hipLaunchKernelGGL(gather_chase_kernel, dim3(waves), dim3(64), 0, s, (const uint4*)d_t, (unsigned int)(pow2 - 1), steps, lanesPerWave, d_o);
        if (err == hipSuccess) err = hipEventRecord(e1, s);
        if (err == hipSuccess) err = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        if (err == hipSuccess && ms < best) best = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_t);
    (void)hipFree(d_o);
    if (err != hipSuccess) return hip_fail(err, "ntr_selftest_gather_rate");
    *seconds = best * 1e-3f;
    return NTR_OK;
}

extern "C" int ntr_bvh_validate(const void* d_nodes, int64_t nodesBytes, uint32_t* flags, void* stream)
{
    using namespace ntr;
    if (!flags) return set_error(NTR_ERR_INVALID, "ntr_bvh_validate: null flags");
    *flags = 0;
    // the same extent rule as ntr_trace_bvh: child pointers are S32 byte offsets below the sentinel 0x76543210
    if (!d_nodes || nodesBytes < 64 || (nodesBytes % 64) != 0 || nodesBytes > 0x76543200ll)
        return set_error(NTR_ERR_INVALID, "ntr_bvh_validate: node buffer size must be a multiple of 64 in [64, 0x76543200]");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = nodesBytes / 16;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    unsigned int* d_bad = nullptr;   // [0] flag bits, [4 + 2 b], [5 + 2 b]: leaf count / last leaf offset seen by workgroup b
    const size_t words = 4 + 2 * (size_t)blocks;
    NTR_HIP(hipMalloc((void**)&d_bad, words * sizeof(unsigned int)));
    NTR_HIP(hipMemsetAsync(d_bad, 0, 4 * sizeof(unsigned int), s));
    hipLaunchKernelGGL(bvh_validate_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)d_nodes, n4, d_bad);
    NTR_HIP(hipGetLastError());
    std::vector<unsigned int> hv(words, 0u);
    NTR_HIP(hipMemcpyAsync(hv.data(), d_bad, words * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    NTR_HIP(hipStreamSynchronize(s));
    NTR_HIP(hipFree(d_bad));
    const unsigned int bad = hv[0];
    unsigned int h[4] = {bad, 0, 0, 0};
    for (int b = 0; b < blocks; b++) { h[1] += hv[4 + 2 * b]; h[2] = hv[5 + 2 * b] > h[2] ? hv[5 + 2 * b] : h[2]; }
    // the last leaf starts at float4 index h[2]: with L leaves the triWoop buffer holds about h[2] / L float4 per leaf (3 per triangle + 1)
    if (h[1] > 0 && (double)h[2] / (double)h[1] >= 7.0) *flags |= NTR_BVH_WIDE_LEAVES;
    if (!(bad & 1u)) *flags |= NTR_BVH_FINITE;
    if (!(bad & 2u)) *flags |= NTR_BVH_FASTDIV;
    if (!(bad & 4u)) *flags |= NTR_BVH_NOTINY;
    if (!(bad & 8u)) *flags |= NTR_BVH_ORDERED;
    // hosts validate after every (re)build: refresh the top-of-tree table the dispatch-order prediction uses
    return ntr_top_table_refresh(d_nodes, nodesBytes, stream);
}
