// lbvh_compact.h -- hole compaction of the LBVH builder (included once by lbvh_kernels.hip, after lbvh_workspace.h).
//
// Where the reference's depth rule (a node at depth 29 only has leaf children: level bit 0, emitTreeKernel.cu:289-292) makes a leaf of
// more than leafSize equal codes, the bottom-up numbering has set aside node indices and terminator slots INSIDE that leaf; the
// builder zero-fills them and records them in two bitmasks.  This rare pass squeezes them out, so that the buffers' extents equal
// the reference's exact sizes (HLBVHBuilder.cpp:382-386; a bvhcache file then has the reference's size): ranks over the bitmasks
// (chunks of 512 bits), every node moved to (index - holes before it) with both child references remapped, every Woop row / triangle
// index moved to (slot - holes before it).  Out of place, copied back.
#pragma once

namespace ntr {
constexpr int HOLE_CHUNK_WORDS = 8;
__global__ __launch_bounds__(256) void lbvh_hole_count_kernel(const unsigned long long* __restrict__ bits, int numChunks, unsigned int* __restrict__ chunkCount)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= numChunks) return;
    unsigned int n = 0;
#pragma unroll
    for (int w = 0; w < HOLE_CHUNK_WORDS; w++) n += (unsigned int)__popcll(bits[(size_t)c * HOLE_CHUNK_WORDS + w]);
    chunkCount[c] = n;
}
__global__ __launch_bounds__(1024) void lbvh_hole_scan_kernel(unsigned int* __restrict__ chunkCount, int numChunks)   // exclusive, in place, one workgroup
{
    __shared__ unsigned int s_part[1024];
    __shared__ unsigned int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < numChunks; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned int v = i < numChunks ? chunkCount[i] : 0u;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned int t = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        const unsigned int incl = s_part[threadIdx.x], carry = s_carry;
        if (i < numChunks) chunkCount[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + incl;
        __syncthreads();
    }
}
__device__ __forceinline__ unsigned int hole_rank(const unsigned long long* __restrict__ bits, const unsigned int* __restrict__ chunkBase, unsigned int i)
{
    const unsigned int w = i >> 6, c = w / HOLE_CHUNK_WORDS;
    unsigned int r = chunkBase[c];
    for (unsigned int k = c * HOLE_CHUNK_WORDS; k < w; k++) r += (unsigned int)__popcll(bits[k]);
    return r + (unsigned int)__popcll(bits[w] & ((1ull << (i & 63)) - 1ull));
}
__global__ __launch_bounds__(256) void lbvh_hole_move_nodes_kernel(const int4* __restrict__ nodes, unsigned int numNodes, const unsigned long long* __restrict__ nodeBits,
                                                                  const unsigned int* __restrict__ nodeBase, const unsigned long long* __restrict__ slotBits,
                                                                  const unsigned int* __restrict__ slotBase, int4* __restrict__ out)
{
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numNodes || ((nodeBits[i >> 6] >> (i & 63)) & 1ull)) return;
    const unsigned int ni = i - hole_rank(nodeBits, nodeBase, i);
    int4 w3 = nodes[(size_t)i * 4 + 3];
    auto remap = [&](int c) {
        if (c >= 0) { const unsigned int idx = (unsigned int)c >> 6; return (int)((idx - hole_rank(nodeBits, nodeBase, idx)) << 6); }
        const unsigned int o = (unsigned int)~c;
        return ~(int)(o - hole_rank(slotBits, slotBase, o));
    };
    w3.x = remap(w3.x);
    w3.y = remap(w3.y);
    out[(size_t)ni * 4 + 0] = nodes[(size_t)i * 4 + 0];
    out[(size_t)ni * 4 + 1] = nodes[(size_t)i * 4 + 1];
    out[(size_t)ni * 4 + 2] = nodes[(size_t)i * 4 + 2];
    out[(size_t)ni * 4 + 3] = w3;
}
__global__ __launch_bounds__(256) void lbvh_hole_move_slots_kernel(const float4* __restrict__ woop, const int* __restrict__ idx, unsigned int numSlots,
                                                                  const unsigned long long* __restrict__ slotBits, const unsigned int* __restrict__ slotBase,
                                                                  float4* __restrict__ outWoop, int* __restrict__ outIdx)
{
    const unsigned int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= numSlots || ((slotBits[o >> 6] >> (o & 63)) & 1ull)) return;
    const unsigned int no = o - hole_rank(slotBits, slotBase, o);
    outWoop[no] = woop[o];
    outIdx[no] = idx[o];
}

}  // namespace ntr

namespace {
int lbvh_compact_holes(hipStream_t s, int n, unsigned int numNodes, unsigned int leafs, unsigned int holes, const unsigned long long* nb, size_t holeNodeWords,
                       const unsigned long long* sb, size_t holeSlotWords, void* d_nodes, void* d_triWoop, int32_t* d_triIndex)
{
    using namespace ntr;
    const unsigned int numSlots = (unsigned int)n * 3u + leafs;
    const int nodeChunks = (int)(holeNodeWords / HOLE_CHUNK_WORDS), slotChunks = (int)(holeSlotWords / HOLE_CHUNK_WORDS);
    DevMem nodeBase, slotBase, tmpNodes, tmpWoop, tmpIdx;
    NTR_HIP(hipMalloc(&nodeBase.p, (size_t)nodeChunks * 4));
    NTR_HIP(hipMalloc(&slotBase.p, (size_t)slotChunks * 4));
    NTR_HIP(hipMalloc(&tmpNodes.p, (size_t)numNodes * 64));
    NTR_HIP(hipMalloc(&tmpWoop.p, (size_t)numSlots * 16));
    NTR_HIP(hipMalloc(&tmpIdx.p, (size_t)numSlots * 4));
    hipLaunchKernelGGL(lbvh_hole_count_kernel, dim3((nodeChunks + 255) / 256), dim3(256), 0, s, nb, nodeChunks, (unsigned int*)nodeBase.p);
    hipLaunchKernelGGL(lbvh_hole_count_kernel, dim3((slotChunks + 255) / 256), dim3(256), 0, s, sb, slotChunks, (unsigned int*)slotBase.p);
    hipLaunchKernelGGL(lbvh_hole_scan_kernel, dim3(1), dim3(1024), 0, s, (unsigned int*)nodeBase.p, nodeChunks);
    hipLaunchKernelGGL(lbvh_hole_scan_kernel, dim3(1), dim3(1024), 0, s, (unsigned int*)slotBase.p, slotChunks);
    hipLaunchKernelGGL(lbvh_hole_move_nodes_kernel, dim3((numNodes + 255) / 256), dim3(256), 0, s, (const int4*)d_nodes, numNodes, nb,
                       (const unsigned int*)nodeBase.p, sb, (const unsigned int*)slotBase.p, (int4*)tmpNodes.p);
    hipLaunchKernelGGL(lbvh_hole_move_slots_kernel, dim3((numSlots + 255) / 256), dim3(256), 0, s, (const float4*)d_triWoop, (const int*)d_triIndex, numSlots,
                       sb, (const unsigned int*)slotBase.p, (float4*)tmpWoop.p, (int*)tmpIdx.p);
    NTR_HIP(hipGetLastError());
    NTR_HIP(hipMemcpyAsync(d_nodes, tmpNodes.p, (size_t)(numNodes - holes) * 64, hipMemcpyDeviceToDevice, s));
    NTR_HIP(hipMemcpyAsync(d_triWoop, tmpWoop.p, (size_t)(numSlots - holes) * 16, hipMemcpyDeviceToDevice, s));
    NTR_HIP(hipMemcpyAsync(d_triIndex, tmpIdx.p, (size_t)(numSlots - holes) * 4, hipMemcpyDeviceToDevice, s));
    NTR_HIP(hipStreamSynchronize(s));
    return NTR_OK;
}
}  // namespace
