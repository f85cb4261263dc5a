// bvh_utils.hip -- small whole-buffer passes over a Compact BVH.
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    for (; i < numFloat4; i += stride) {
        if ((i & 3) == 3) continue;  // child / split words
        const float4 v = nodes[i];
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
    unsigned int* d_bad = nullptr;
    NTR_HIP(hipMalloc((void**)&d_bad, sizeof(unsigned int)));
    NTR_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), s));
    const int64_t n4 = nodesBytes / 16;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bvh_validate_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)d_nodes, n4, d_bad);
    NTR_HIP(hipGetLastError());
    unsigned int bad = 0;
    NTR_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, s));
    NTR_HIP(hipStreamSynchronize(s));
    NTR_HIP(hipFree(d_bad));
    if (!(bad & 1u)) *flags |= NTR_BVH_FINITE;
    if (!(bad & 2u)) *flags |= NTR_BVH_FASTDIV;
    if (!(bad & 4u)) *flags |= NTR_BVH_NOTINY;
    if (!(bad & 8u)) *flags |= NTR_BVH_ORDERED;
    // hosts validate after every (re)build: refresh the top-of-tree table the dispatch-order prediction uses
    return ntr_top_table_refresh(d_nodes, nodesBytes, stream);
}
