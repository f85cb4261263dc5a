// rayops_kernels.hip -- the callers on either side of the tracer:
//   reconstructKernel    src/rt/cuda/RendererKernels.cu:59-172 (primary / AO / diffuse branches;
//                        textured, path-traced and VPL shading are out of scope)
//   secondary-ray sort   src/rt/ray/RayBuffer.cpp:103-165 + RayBufferKernels.cu:70-197:
//                        findAABB -> 192-bit Morton keys -> sort -> reorder.  The reference sorts the
//                        keys on the CPU (RayBuffer.cpp:149); here the sort is the one-sweep LSD radix sort of
//                        radix_sort.h over an index array: 19 digit passes cover the 150 significant bits, one
//                        launch each, their histograms taken while the keys are produced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <mutex>

#include "ntr_internal.h"
#include "device_scratch.h"
#include "radix_sort.h"

namespace ntr {

// ---- reconstructKernel ---------------------------------------------------------------------------
struct Col { float x, y, z, w; };
__device__ __forceinline__ Col from_abgr(uint32_t c)  // RendererKernels.cu:37-44
{
    return {(float)(c & 0xFF) * (1.0f / 255.0f), (float)((c >> 8) & 0xFF) * (1.0f / 255.0f),
            (float)((c >> 16) & 0xFF) * (1.0f / 255.0f), (float)(c >> 24) * (1.0f / 255.0f)};
}
__device__ __forceinline__ uint32_t to_abgr(Col v)    // RendererKernels.cu:48-55
{
    return (uint32_t)(fminf(fmaxf(v.x, 0.0f), 1.0f) * 255.0f) | ((uint32_t)(fminf(fmaxf(v.y, 0.0f), 1.0f) * 255.0f) << 8) |
           ((uint32_t)(fminf(fmaxf(v.z, 0.0f), 1.0f) * 255.0f) << 16) | ((uint32_t)(fminf(fmaxf(v.w, 0.0f), 1.0f) * 255.0f) << 24);
}

__global__ __launch_bounds__(256) void reconstruct_kernel(int rayType, int numRaysPerPrimary, int firstPrimary, int numPrimary,
                                                          const int* __restrict__ primarySlotToID,
                                                          const NtrRayResult* __restrict__ primaryResults,
                                                          const int* __restrict__ batchIDToSlot,
                                                          const NtrRayResult* __restrict__ batchResults,
                                                          const uint32_t* __restrict__ triMaterialColor,
                                                          const uint32_t* __restrict__ triShadedColor, uint32_t* __restrict__ pixels)
{
    const int taskIdx = blockIdx.x * blockDim.x + threadIdx.x;
    if (taskIdx >= numPrimary) return;
    const bool isPrimary = rayType == 0, isAO = rayType == 1, isDiffuse = rayType == 2;
    const int primarySlot = firstPrimary + taskIdx;
    const int primaryID = primarySlotToID[primarySlot];
    const int primaryTri = primaryResults[primarySlot].id;
    const int* batchSlots = batchIDToSlot + (isPrimary ? primaryID : taskIdx * numRaysPerPrimary);
    const Col bg = {0.2f, 0.4f, 0.8f, 1.0f};

    Col color = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 0; i < numRaysPerPrimary; i++) {
        const int tri = batchResults[batchSlots[i]].id;
        Col add;
        if (tri == -1) add = isPrimary ? bg : Col{1.0f, 1.0f, 1.0f, 1.0f};
        else if (isAO) add = Col{0.0f, 0.0f, 0.0f, 1.0f};
        else add = from_abgr(triShadedColor[tri]);
        color.x += add.x; color.y += add.y; color.z += add.z; color.w += add.w;
    }
    const float s = 1.0f / (float)numRaysPerPrimary;
    color.x *= s; color.y *= s; color.z *= s; color.w *= s;
    if (isAO && primaryTri == -1) color = bg;
    if (isDiffuse) {
        const Col m = (primaryTri == -1) ? bg : from_abgr(triMaterialColor[primaryTri]);
        color.x *= m.x; color.y *= m.y; color.z *= m.z; color.w *= m.w;
    }
    pixels[primaryID] = to_abgr(color);
}

// ---- ray sort --------------------------------------------------------------------------------------
// order-preserving float <-> uint mapping for atomicMin/Max on floats
__device__ __forceinline__ unsigned int f2ord(float f) { unsigned int u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned int u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// findAABBKernel (RayBufferKernels.cu:70-136): box of ray origins and end points origin + direction * tmax.
__global__ __launch_bounds__(256) void ray_aabb_kernel(int n, const NtrRay* __restrict__ rays, unsigned int* __restrict__ box /* lo xyz, hi xyz (ordered uints) */)
{
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 o = reinterpret_cast<const float4*>(rays)[2 * i], d = reinterpret_cast<const float4*>(rays)[2 * i + 1];
        const float p[3] = {o.x, o.y, o.z}, e[3] = {o.x + d.x * d.w, o.y + d.y * d.w, o.z + d.z * d.w};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(fminf(lo[k], p[k]), e[k]);
            hi[k] = fmaxf(fmaxf(hi[k], p[k]), e[k]);
        }
    }
    // one atomic pair per component and WORKGROUP (wave shuffles, then LDS): thousands of waves hammering six addresses
    // serialise at about 70 ns per atomic
    __shared__ float s_lo[4][3], s_hi[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        if (lane == 0) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = s_lo[0][k], b = s_hi[0][k];
        for (int w = 1; w < 4; w++) { a = fminf(a, s_lo[w][k]); b = fmaxf(b, s_hi[w][k]); }
        atomicMin(&box[k], f2ord(a));
        atomicMax(&box[3 + k], f2ord(b));
    }
}

__global__ void ray_aabb_decode_kernel(const unsigned int* __restrict__ box, float* __restrict__ out)
{
    if (threadIdx.x < 6) out[threadIdx.x] = ord2f(box[threadIdx.x]);
}

// genMortonKeysKernel (RayBufferKernels.cu:140-175): 6 x 32 bits interleaved (component c, bit i -> bit c + 6 i).
constexpr int RAY_KEY_DIGITS = 19;   // 8-bit digits of the 150 significant key bits: words 0..3 fully, word 4 bits 0..23
static_assert(RAY_KEY_DIGITS <= OS_MAX_PASSES, "one clearing of the tile state serves all passes: the pass number must fit the status tag");

__global__ __launch_bounds__(256) void ray_keys_kernel(int n, const NtrRay* __restrict__ rays, const float* __restrict__ box,
                                                       unsigned int* __restrict__ keys /* 6 words per ray */, int* __restrict__ idx,
                                                       unsigned int* __restrict__ hist /* [RAY_KEY_DIGITS][256], zeroed */)
{
    __shared__ unsigned int s_hist[RAY_KEY_DIGITS][256];
    for (int i = threadIdx.x; i < RAY_KEY_DIGITS * 256; i += 256) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const float4 o = reinterpret_cast<const float4*>(rays)[2 * t], d = reinterpret_cast<const float4*>(rays)[2 * t + 1];
    const float ax = (o.x - box[0]) / (box[3] - box[0]), ay = (o.y - box[1]) / (box[4] - box[1]), az = (o.z - box[2]) / (box[5] - box[2]);
    // normalize(v) = v * (1 * rcp(length(v)))  (Math.hpp:141-142)
    const float inv = 1.0f * (1.0f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z));
    const float bx = (d.x * inv + 1.0f) * 0.5f, by = (d.y * inv + 1.0f) * 0.5f, bz = (d.z * inv + 1.0f) * 0.5f;
    const unsigned int c[6] = {(unsigned int)(ax * 256.0f * 65536.0f), (unsigned int)(ay * 256.0f * 65536.0f),
                               (unsigned int)(az * 256.0f * 65536.0f), (unsigned int)(bx * 32.0f * 65536.0f),
                               (unsigned int)(by * 32.0f * 65536.0f), (unsigned int)(bz * 32.0f * 65536.0f)};
    unsigned int h[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int pos = k + i * 6;
            h[pos >> 5] |= ((c[k] >> i) & 1u) << (pos & 31);
        }
#pragma unroll
    for (int k = 0; k < 6; k++) keys[6 * t + k] = h[k];
    idx[t] = t;
#pragma unroll
    for (int p = 0; p < RAY_KEY_DIGITS; p++) atomicAdd(&s_hist[p][(h[p >> 2] >> ((p & 3) * 8)) & 255u], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RAY_KEY_DIGITS * 256; i += 256) {
        const unsigned int v = (&s_hist[0][0])[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// reorderRaysKernel (RayBufferKernels.cu:179-197)
__global__ __launch_bounds__(256) void ray_reorder_kernel(int n, const int* __restrict__ order, const NtrRay* __restrict__ inRays,
                                                          const int* __restrict__ inSlotToID, NtrRay* __restrict__ outRays,
                                                          int* __restrict__ outIDToSlot, int* __restrict__ outSlotToID)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int oldSlot = order[t];
    const int id = inSlotToID[oldSlot];
    reinterpret_cast<float4*>(outRays)[2 * t] = reinterpret_cast<const float4*>(inRays)[2 * oldSlot];
    reinterpret_cast<float4*>(outRays)[2 * t + 1] = reinterpret_cast<const float4*>(inRays)[2 * oldSlot + 1];
    outIDToSlot[id] = t;
    outSlotToID[t] = id;
}

}  // namespace ntr

using namespace ntr;

namespace {
// Grow-only scratch of ntr_ray_morton_sort, one per device, kept between calls (device_scratch.h): a renderer sorts sixteen batches per
// frame and must not pay seven hipMalloc / hipFree pairs (each a device synchronisation) for every one of them.
// ntr_lbvh_release_workspace returns it.
ntr::DeviceScratchPool g_sortScratch;
}  // namespace

namespace ntr {
int raysort_scratch_release() { return g_sortScratch.release(); }
}  // namespace ntr

// Framebuffer gather of the multi-GPU path (ntr_dist.cpp): a rank's pixels packed in slot order / scattered back on the root.
__global__ __launch_bounds__(256) void pixels_pack_kernel(const uint32_t* __restrict__ pixels, const int32_t* __restrict__ slotToPixel, int first, int count,
                                                         uint32_t* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = pixels[slotToPixel[first + i]];
}
__global__ __launch_bounds__(256) void pixels_unpack_kernel(const uint32_t* __restrict__ bySlot, const int32_t* __restrict__ slotToPixel, int count,
                                                           uint32_t* __restrict__ pixels)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < count) pixels[slotToPixel[i]] = bySlot[i];
}

extern "C" hipError_t ntr_launch_pixels_pack(const uint32_t* d_pixels, const int32_t* d_slotToPixel, int first, int count, uint32_t* d_out, hipStream_t s)
{
    if (count > 0) hipLaunchKernelGGL(pixels_pack_kernel, dim3((count + 255) / 256), dim3(256), 0, s, d_pixels, d_slotToPixel, first, count, d_out);
    return hipGetLastError();
}
extern "C" hipError_t ntr_launch_pixels_unpack(const uint32_t* d_bySlot, const int32_t* d_slotToPixel, int count, uint32_t* d_pixels, hipStream_t s)
{
    if (count > 0) hipLaunchKernelGGL(pixels_unpack_kernel, dim3((count + 255) / 256), dim3(256), 0, s, d_bySlot, d_slotToPixel, count, d_pixels);
    return hipGetLastError();
}

extern "C" {

int ntr_reconstruct(int32_t rayType, int32_t numRaysPerPrimary, int32_t firstPrimary, int32_t numPrimary,
                    const int32_t* d_primarySlotToID, const NtrRayResult* d_primaryResults, const int32_t* d_batchIDToSlot,
                    const NtrRayResult* d_batchResults, const uint32_t* d_triMaterialColor, const uint32_t* d_triShadedColor,
                    uint32_t* d_pixels, void* stream)
{
    if (rayType < 0 || rayType > 2 || numRaysPerPrimary < 1 || firstPrimary < 0 || numPrimary < 0)
        return set_error(NTR_ERR_INVALID, "ntr_reconstruct: bad argument");
    if (numPrimary == 0) return NTR_OK;
    if (!d_primarySlotToID || !d_primaryResults || !d_batchIDToSlot || !d_batchResults || !d_pixels ||
        (rayType == 2 && (!d_triMaterialColor || !d_triShadedColor)) || (rayType == 0 && !d_triShadedColor))
        return set_error(NTR_ERR_INVALID, "ntr_reconstruct: null buffer");
    hipLaunchKernelGGL(reconstruct_kernel, dim3((numPrimary + 255) / 256), dim3(256), 0, (hipStream_t)stream, rayType,
                       numRaysPerPrimary, firstPrimary, numPrimary, d_primarySlotToID, d_primaryResults, d_batchIDToSlot,
                       d_batchResults, d_triMaterialColor, d_triShadedColor, d_pixels);
    NTR_HIP(hipGetLastError());
    return NTR_OK;
}

int ntr_ray_morton_sort(int32_t numRays, const NtrRay* d_inRays, const int32_t* d_inSlotToID, NtrRay* d_outRays,
                        int32_t* d_outIDToSlot, int32_t* d_outSlotToID, void* stream, float* seconds)
{
    if (seconds) *seconds = 0.0f;
    if (numRays < 0) return set_error(NTR_ERR_INVALID, "ntr_ray_morton_sort: numRays < 0");
    if (numRays == 0) return NTR_OK;
    if (!d_inRays || !d_inSlotToID || !d_outRays || !d_outIDToSlot || !d_outSlotToID || d_inRays == d_outRays)
        return set_error(NTR_ERR_INVALID, "ntr_ray_morton_sort: null or aliased buffer");
    hipStream_t s = (hipStream_t)stream;
    const int n = numRays;
    if (n >= (1 << 28)) return set_error(NTR_ERR_INVALID, "ntr_ray_morton_sort: at most 2^28 - 1 rays");
    constexpr int ITEMS = 8;
    const int tiles = (n + OS_THREADS * ITEMS - 1) / (OS_THREADS * ITEMS);
    // one zeroed block: digit histograms, per-pass tickets, error flag, then the tile state of the chained scans
    const size_t histWords = (size_t)RAY_KEY_DIGITS * 256, miscWords = 32, stateWords = (size_t)tiles * 256 * 2;   // 64-bit state words
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t oKeys = take((size_t)n * 24), oWordA = take((size_t)n * 4), oWordB = take((size_t)n * 4), oIdxA = take((size_t)n * 4), oIdxB = take((size_t)n * 4);
    const size_t oZero = take((histWords + miscWords + stateWords) * 4), oBox = take(64);
    void* base = nullptr;
    {
        const int rc = g_sortScratch.reserve(off, &base);
        if (rc != NTR_OK) return rc;
    }
    char* ws = (char*)base;
    struct { void* p; } keys{ws + oKeys}, wordA{ws + oWordA}, wordB{ws + oWordB}, idxA{ws + oIdxA}, idxB{ws + oIdxB}, scratch{ws + oZero}, box{ws + oBox};
    unsigned int* histp = (unsigned int*)scratch.p;
    unsigned int* misc = histp + histWords;      // [0..18] tickets, [31] error flag
    unsigned long long* state = (unsigned long long*)(misc + miscWords);   // 8-byte aligned: histWords and miscWords are even
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (seconds) { NTR_HIP(hipEventCreate(&e0)); NTR_HIP(hipEventCreate(&e1)); NTR_HIP(hipEventRecord(e0, s)); }
    NTR_HIP(hipMemsetAsync(scratch.p, 0, (histWords + miscWords + stateWords) * 4, s));

    unsigned int* ubox = (unsigned int*)box.p;
    float* fbox = (float*)box.p + 8;
    const unsigned int init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    NTR_HIP(hipMemcpyAsync(ubox, init, sizeof(init), hipMemcpyHostToDevice, s));
    int blocks = (n + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(ray_aabb_kernel, dim3(blocks), dim3(256), 0, s, n, d_inRays, ubox);
    hipLaunchKernelGGL(ray_aabb_decode_kernel, dim3(1), dim3(64), 0, s, (const unsigned int*)ubox, fbox);
    int kblocks = (n + 1023) / 1024;
    if (kblocks > 512) kblocks = 512;
    hipLaunchKernelGGL(ray_keys_kernel, dim3(kblocks), dim3(256), 0, s, n, d_inRays, (const float*)fbox, (unsigned int*)keys.p, (int*)idxA.p, histp);
    // stable LSD sort of the index array by the 192-bit key: words 0..3 fully, word 4 bits 0..23
    // (the highest set bit is 5 + 6*24 = 149: a* < 2^25, b* < 2^22); word 5 is always zero.
    // The tile-state words carry the pass number in their status bits (8 of them: pass p uses 2p+1, 2p+2), so one clearing serves all passes.
    // Word by word: the first pass over a word fetches it through the index array (one random 4-byte read per ray) and from then on
    // the word travels with the index, so the other passes over it are plain streaming passes -- 5 gathers instead of 19.
    int *vIn = (int*)idxA.p, *vOut = (int*)idxB.p;
    unsigned int *kIn = (unsigned int*)wordA.p, *kOut = (unsigned int*)wordB.p;
    for (int p = 0; p < RAY_KEY_DIGITS; p++) {
        const int word = p >> 2, shift = (p & 3) * 8;
        if ((p & 3) == 0)
            onesweep_launch<ITEMS, 2, true>(s, tiles, n, (const unsigned int*)keys.p + word, (const int*)vIn, kOut, vOut, 6, shift, p,
                                            (const unsigned int*)(histp + (size_t)p * 256), state, misc + p, misc + 31);
        else
            onesweep_launch<ITEMS, 0, true>(s, tiles, n, (const unsigned int*)kIn, (const int*)vIn, kOut, vOut, 1, shift, p,
                                            (const unsigned int*)(histp + (size_t)p * 256), state, misc + p, misc + 31);
        int* t = vIn; vIn = vOut; vOut = t;
        unsigned int* tk = kIn; kIn = kOut; kOut = tk;
    }
    hipLaunchKernelGGL(ray_reorder_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, (const int*)vIn, d_inRays, d_inSlotToID,
                       d_outRays, d_outIDToSlot, d_outSlotToID);
    NTR_HIP(hipGetLastError());
    if (seconds) {
        NTR_HIP(hipEventRecord(e1, s));
        NTR_HIP(hipEventSynchronize(e1));
        float ms = 0;
        NTR_HIP(hipEventElapsedTime(&ms, e0, e1));
        *seconds = ms * 1e-3f;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    unsigned int sortErr = 0;
    NTR_HIP(hipMemcpyAsync(&sortErr, misc + 31, sizeof(sortErr), hipMemcpyDeviceToHost, s));
    NTR_HIP(hipStreamSynchronize(s));  // (the error word is read back: a timed-out chained scan must not go unnoticed)
    if (sortErr) return set_error(NTR_ERR_HIP, "ntr_ray_morton_sort: a chained scan timed out waiting for a predecessor tile");
    return NTR_OK;
}

}  // extern "C"
