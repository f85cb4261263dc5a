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

}  // namespace ntr

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
