// raygen_kernels.hip -- ray production on the device:
//   rayGenPrimaryKernel  src/rt/ray/RayGenKernels.cu:77-125
//   rayGenAOKernel       src/rt/ray/RayGenKernels.cu:129-236  (AO and, with maxDist = camera
//                        far + closest hit, the diffuse rays of Renderer.cpp:533-537)
//   PixelTable           src/rt/ray/PixelTable.cpp:57-143      (host, uploaded once)
// The reference compiles these with -use_fast_math, so its own bits are not reproducible
// across GPUs; parity for the tracer is defined on identical ray buffers.  These kernels
// use the precise libm forms and are tested against a numpy restatement to 1e-5.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "ntr_internal.h"

namespace ntr {

__device__ __forceinline__ void jenkins_mix(uint32_t& a, uint32_t& b, uint32_t& c)
{
    a -= b; a -= c; a ^= (c >> 13);
    b -= c; b -= a; b ^= (a << 8);
    c -= a; c -= b; c ^= (b >> 13);
    a -= b; a -= c; a ^= (c >> 12);
    b -= c; b -= a; b ^= (a << 16);
    c -= a; c -= b; c ^= (b >> 5);
    a -= b; a -= c; a ^= (c >> 3);
    b -= c; b -= a; b ^= (a << 10);
    c -= a; c -= b; c ^= (b >> 15);
}

struct Mat4 { float m[16]; };  // row-major: out[r] = sum_c m[4r+c] * in[c]

__global__ __launch_bounds__(256) void raygen_primary_kernel(NtrRay* __restrict__ rays, int32_t* __restrict__ idToSlot,
                                                             int32_t* __restrict__ slotToID,
                                                             const int32_t* __restrict__ indexToPixel, float ox, float oy,
                                                             float oz, Mat4 nscreenToWorld, int w, int h, float maxDist,
                                                             uint32_t randomSeed)
{
    const int taskIdx = blockIdx.x * blockDim.x + threadIdx.x;
    if (taskIdx >= w * h) return;
    const int pixel = indexToPixel[taskIdx];

    float nx = 2.0f * ((float)(pixel % w) + 0.5f) / (float)w - 1.0f;
    float ny = 2.0f * ((float)(pixel / w) + 0.5f) / (float)h - 1.0f;
    if (randomSeed != 0) {
        uint32_t a = randomSeed + (uint32_t)taskIdx, b = 0x9e3779b9u, c = 0x9e3779b9u;
        jenkins_mix(a, b, c);
        jenkins_mix(a, b, c);
        nx += (float)a * 0x1p-32f * 0.005f;
        ny += (float)b * 0x1p-32f * 0.005f;
    }
    const float* m = nscreenToWorld.m;  // nscreenPos = (nx, ny, 0, 1)
    const float wx = m[0] * nx + m[1] * ny + m[3];
    const float wy = m[4] * nx + m[5] * ny + m[7];
    const float wz = m[8] * nx + m[9] * ny + m[11];
    const float ww = m[12] * nx + m[13] * ny + m[15];
    float dx = wx / ww - ox, dy = wy / ww - oy, dz = wz / ww - oz;
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);

    slotToID[taskIdx] = pixel;
    idToSlot[pixel] = taskIdx;
    float4* out = reinterpret_cast<float4*>(rays + taskIdx);
    out[0] = make_float4(ox, oy, oz, 0.0f);
    out[1] = make_float4(dx * inv, dy * inv, dz * inv, maxDist);
}

__global__ __launch_bounds__(256) void raygen_ao_kernel(NtrRay* __restrict__ outRays, int32_t* __restrict__ outIDToSlot,
                                                        int32_t* __restrict__ outSlotToID, const NtrRay* __restrict__ inRays,
                                                        const NtrRayResult* __restrict__ inResults,
                                                        const float* __restrict__ normals, int firstInputSlot,
                                                        int numInputRays, int numSamples, float maxDist, uint32_t randomSeed)
{
    const int taskIdx = blockIdx.x * blockDim.x + threadIdx.x;
    if (taskIdx >= numInputRays) return;
    const int inSlot = taskIdx + firstInputSlot;
    const float4 ro = reinterpret_cast<const float4*>(inRays + inSlot)[0];
    const float4 rd = reinterpret_cast<const float4*>(inRays + inSlot)[1];
    const int4 res = reinterpret_cast<const int4*>(inResults)[inSlot];
    const int outSlot = taskIdx * numSamples;

    // origin, backed off a little (RayGenKernels.cu:150-151)
    const float back = fmaxf(__int_as_float(res.y) - 1.0e-4f, 0.0f);
    const float px = ro.x + rd.x * back, py = ro.y + rd.y * back, pz = ro.z + rd.z * back;

    const int tri = res.x;
    float nx = 1.0f, ny = 0.0f, nz = 0.0f;
    if (tri != -1) { nx = normals[3 * tri + 0]; ny = normals[3 * tri + 1]; nz = normals[3 * tri + 2]; }
    if (nx * rd.x + ny * rd.y + nz * rd.z > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }

    // perpendicular frame (:164-175)
    const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
    const float nm = fmaxf(fmaxf(ax, ay), az);
    float ux = ny, uy = -nx, uz = 0.0f;
    if (nm == az) { ux = 0.0f; uy = nz; uz = -ny; }
    else if (nm == ax) { ux = -nz; uy = 0.0f; uz = nx; }
    const float ul = 1.0f / sqrtf(ux * ux + uy * uy + uz * uz);
    ux *= ul; uy *= ul; uz *= ul;
    const float bx = ny * uz - nz * uy, by = nz * ux - nx * uz, bz = nx * uy - ny * ux;

    // random rotation (:179-190)
    uint32_t ha = randomSeed + (uint32_t)taskIdx, hb = 0x9e3779b9u, hc = 0x9e3779b9u;
    jenkins_mix(ha, hb, hc);
    jenkins_mix(ha, hb, hc);
    const float angle = 2.0f * 3.14159265358979323846f * (float)hc * 0x1p-32f;
    float sa, ca;
    sincosf(angle, &sa, &ca);
    const float t0x = ux * ca + bx * sa, t0y = uy * ca + by * sa, t0z = uz * ca + bz * sa;
    const float t1x = ux * -sa + bx * ca, t1y = uy * -sa + by * ca, t1z = uz * -sa + bz * ca;

    const float tmax = (tri == -1) ? -1.0f : maxDist;
    for (int i = 0; i < numSamples; i++) {
        // Halton(2,3) (:196-218)
        float x = 0.0f, xadd = 1.0f;
        for (unsigned int h2 = i + 1; h2 != 0; h2 >>= 1) {
            xadd *= 0.5f;
            if (h2 & 1) x += xadd;
        }
        float y = 0.0f, yadd = 1.0f;
        for (int h3 = i + 1; h3 != 0; h3 /= 3) {
            yadd *= 1.0f / 3.0f;
            y += (float)(h3 % 3) * yadd;
        }
        // cosine-weighted hemisphere (:222-226)
        const float ang = 2.0f * 3.14159265358979323846f * y;
        const float r = sqrtf(x);
        float s, c;
        sincosf(ang, &s, &c);
        x = r * c;
        y = r * s;
        const float z = sqrtf(1.0f - x * x - y * y);
        float dx = x * t0x + y * t1x + z * nx, dy = x * t0y + y * t1y + z * ny, dz = x * t0z + y * t1z + z * nz;
        const float dl = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        float4* out = reinterpret_cast<float4*>(outRays + outSlot + i);
        out[0] = make_float4(px, py, pz, 0.0f);
        out[1] = make_float4(dx * dl, dy * dl, dz * dl, tmax);
        outIDToSlot[outSlot + i] = outSlot + i;
        outSlotToID[outSlot + i] = outSlot + i;
    }
}

// countHitsKernel (src/rt/cuda/RendererKernels.cu:174-226): number of rays with id != -1.
// rayGenShadowKernel (src/rt/ray/RayGenKernels.cu:240-301): numSamples rays from every input ray's hit point towards points inside a cube of
// half-edge lightRadius around the light -- a (0,2)-sequence / Hammersley point per sample (sobol2D :52-73, hammersley :47-50), shifted per
// input ray by a Jenkins-hashed offset (Cranley-Patterson, :267-286); tmax = the distance to the target, -1 for rays of missed inputs.
__global__ __launch_bounds__(256) void raygen_shadow_kernel(NtrRay* __restrict__ outRays, int32_t* __restrict__ outIDToSlot,
                                                            int32_t* __restrict__ outSlotToID, const NtrRay* __restrict__ inRays,
                                                            const NtrRayResult* __restrict__ inResults, int firstInputSlot, int numInputRays,
                                                            int numSamples, float lx, float ly, float lz, float lightRadius, uint32_t randomSeed)
{
    const int taskIdx = blockIdx.x * blockDim.x + threadIdx.x;
    if (taskIdx >= numInputRays) return;
    const int inSlot = taskIdx + firstInputSlot;
    const float4 ro = reinterpret_cast<const float4*>(inRays + inSlot)[0];
    const float4 rd = reinterpret_cast<const float4*>(inRays + inSlot)[1];
    const int4 res = reinterpret_cast<const int4*>(inResults)[inSlot];
    const int outSlot = taskIdx * numSamples;

    // origin, backed off a little (:259-260: epsilon 1e-2 here, 1e-4 in the AO generator)
    const float back = fmaxf(__int_as_float(res.y) - 1.0e-2f, 0.0f);
    const float px = ro.x + rd.x * back, py = ro.y + rd.y * back, pz = ro.z + rd.z * back;

    uint32_t ha = randomSeed + (uint32_t)taskIdx, hb = 0x9e3779b9u, hc = 0x9e3779b9u;
    jenkins_mix(ha, hb, hc);
    jenkins_mix(ha, hb, hc);
    const float offx = (float)ha * 0x1p-32f, offy = (float)hb * 0x1p-32f, offz = (float)hc * 0x1p-32f;

    const int tri = res.x;
    for (int i = 0; i < numSamples; i++) {
        unsigned int r1 = 0, r2 = 0;
        int k = i;
        for (unsigned int v1 = 1u << 31, v2 = 3u << 30; k; k >>= 1) {
            if (k & 1) { r1 ^= v1; r2 ^= v2 << 1; }
            v1 |= v1 >> 1;
            v2 ^= v2 >> 1;
        }
        float sx = (float)r1 * 0x1p-32f + offx, sy = (float)r2 * 0x1p-32f + offy, sz = ((float)i + 0.5f) / (float)numSamples + offz;
        if (sx >= 1.0f) sx -= 1.0f;
        if (sy >= 1.0f) sy -= 1.0f;
        if (sz >= 1.0f) sz -= 1.0f;
        sx = sx * 2.0f - 1.0f; sy = sy * 2.0f - 1.0f; sz = sz * 2.0f - 1.0f;
        const float dx = lx + lightRadius * sx - px, dy = ly + lightRadius * sy - py, dz = lz + lightRadius * sz - pz;
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float inv = 1.0f / len;
        float4* out = reinterpret_cast<float4*>(outRays + outSlot + i);
        out[0] = make_float4(px, py, pz, 0.0f);
        out[1] = make_float4(dx * inv, dy * inv, dz * inv, (tri == -1) ? -1.0f : len);
        outIDToSlot[outSlot + i] = outSlot + i;
        outSlotToID[outSlot + i] = outSlot + i;
    }
}

__global__ __launch_bounds__(256) void count_hits_kernel(const NtrRayResult* __restrict__ results, int numRays,
                                                         int* __restrict__ count)
{
    int local = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < numRays; i += gridDim.x * blockDim.x)
        local += (reinterpret_cast<const int4*>(results)[i].x != -1);
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

}  // namespace ntr

using namespace ntr;

// PixelTable::recalculate (src/rt/ray/PixelTable.cpp:57-143) -- host code in the reference too.
static void pixel_table_host(int w, int h, std::vector<int32_t>& idxToPos, std::vector<int32_t>& posToIdx)
{
    idxToPos.assign((size_t)w * h, 0);
    posToIdx.assign((size_t)w * h, 0);
    int idx = 0;
    const int bheight = h & ~7, bwidth = w & ~7;
    int maxdim = (bwidth > bheight) ? bwidth : bheight;
    maxdim |= maxdim >> 1; maxdim |= maxdim >> 2; maxdim |= maxdim >> 4; maxdim |= maxdim >> 8; maxdim |= maxdim >> 16;
    maxdim = (maxdim + 1) >> 1;
    const int width8 = bwidth >> 3, height8 = bheight >> 3;
    for (long long i = 0; i < (long long)maxdim * maxdim; i++) {
        int tx = 0, ty = 0, bit = 1;
        for (long long val = i; val; val >>= 2, bit += bit) {
            if (val & 1) tx |= bit;
            if (val & 2) ty |= bit;
        }
        if (tx < width8 && ty < height8)
            for (int inner = 0; inner < 64; inner++) {
                const int ix = ((inner & 1) >> 0) | ((inner & 4) >> 1) | ((inner & 16) >> 2);
                const int iy = ((inner & 2) >> 1) | ((inner & 8) >> 2) | ((inner & 32) >> 3);
                const int pos = (ty * 8 + iy) * w + (tx * 8 + ix);
                posToIdx[pos] = idx;
                idxToPos[idx++] = pos;
            }
    }
    for (int px = 0; px < bwidth; px++)
        for (int py = bheight; py < h; py++) { const int pos = px + py * w; posToIdx[pos] = idx; idxToPos[idx++] = pos; }
    for (int py = 0; py < h; py++)
        for (int px = bwidth; px < w; px++) { const int pos = px + py * w; posToIdx[pos] = idx; idxToPos[idx++] = pos; }
}

extern "C" {

int ntr_pixel_table(int32_t w, int32_t h, int32_t* d_indexToPixel, int32_t* d_pixelToIndex, void* stream)
{
    if (w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return set_error(NTR_ERR_INVALID, "ntr_pixel_table: bad size");
    std::vector<int32_t> a, b;
    pixel_table_host(w, h, a, b);
    hipStream_t s = (hipStream_t)stream;
    if (d_indexToPixel) NTR_HIP(hipMemcpyAsync(d_indexToPixel, a.data(), a.size() * 4, hipMemcpyHostToDevice, s));
    if (d_pixelToIndex) NTR_HIP(hipMemcpyAsync(d_pixelToIndex, b.data(), b.size() * 4, hipMemcpyHostToDevice, s));
    NTR_HIP(hipStreamSynchronize(s));
    return NTR_OK;
}

int ntr_raygen_primary(NtrRay* d_rays, int32_t* d_idToSlot, int32_t* d_slotToID, const int32_t* d_indexToPixel,
                       const float origin[3], const float nscreenToWorld[16], int32_t w, int32_t h, float maxDist,
                       uint32_t kernelSeed, void* stream)
{
    if (!d_rays || !d_idToSlot || !d_slotToID || !d_indexToPixel || !origin || !nscreenToWorld || w <= 0 || h <= 0)
        return set_error(NTR_ERR_INVALID, "ntr_raygen_primary: bad argument");
    Mat4 m;
    for (int i = 0; i < 16; i++) m.m[i] = nscreenToWorld[i];
    const int n = w * h;
    hipLaunchKernelGGL(raygen_primary_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_rays, d_idToSlot,
                       d_slotToID, d_indexToPixel, origin[0], origin[1], origin[2], m, w, h, maxDist, kernelSeed);
    NTR_HIP(hipGetLastError());
    return NTR_OK;
}

int ntr_raygen_ao(NtrRay* d_outRays, int32_t* d_outIDToSlot, int32_t* d_outSlotToID, const NtrRay* d_inRays,
                  const NtrRayResult* d_inResults, const float* d_triNormals, int32_t firstInputSlot,
                  int32_t numInputRays, int32_t numSamples, float maxDist, uint32_t kernelSeed, void* stream)
{
    if (numInputRays < 0 || numSamples < 0 || firstInputSlot < 0) return set_error(NTR_ERR_INVALID, "ntr_raygen_ao: negative count");
    if (numInputRays == 0 || numSamples == 0) return NTR_OK;
    if (!d_outRays || !d_outIDToSlot || !d_outSlotToID || !d_inRays || !d_inResults || !d_triNormals)
        return set_error(NTR_ERR_INVALID, "ntr_raygen_ao: null buffer");
    hipLaunchKernelGGL(raygen_ao_kernel, dim3((numInputRays + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_outRays,
                       d_outIDToSlot, d_outSlotToID, d_inRays, d_inResults, d_triNormals, firstInputSlot, numInputRays,
                       numSamples, maxDist, kernelSeed);
    NTR_HIP(hipGetLastError());
    return NTR_OK;
}

int ntr_raygen_shadow(NtrRay* d_outRays, int32_t* d_outIDToSlot, int32_t* d_outSlotToID, const NtrRay* d_inRays, const NtrRayResult* d_inResults,
                      int32_t firstInputSlot, int32_t numInputRays, int32_t numSamples, const float lightPos[3], float lightRadius,
                      uint32_t kernelSeed, void* stream)
{
    if (numInputRays < 0 || numSamples < 0 || firstInputSlot < 0) return set_error(NTR_ERR_INVALID, "ntr_raygen_shadow: negative count");
    if (numInputRays == 0 || numSamples == 0) return NTR_OK;
    if (!d_outRays || !d_outIDToSlot || !d_outSlotToID || !d_inRays || !d_inResults || !lightPos)
        return set_error(NTR_ERR_INVALID, "ntr_raygen_shadow: null buffer");
    hipLaunchKernelGGL(raygen_shadow_kernel, dim3((numInputRays + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_outRays, d_outIDToSlot,
                       d_outSlotToID, d_inRays, d_inResults, firstInputSlot, numInputRays, numSamples, lightPos[0], lightPos[1], lightPos[2],
                       lightRadius, kernelSeed);
    NTR_HIP(hipGetLastError());
    return NTR_OK;
}

int ntr_count_hits(const NtrRayResult* d_results, int32_t numRays, int32_t* count, void* stream)
{
    if (!count) return set_error(NTR_ERR_INVALID, "ntr_count_hits: null count");
    *count = 0;
    if (numRays <= 0) return NTR_OK;
    if (!d_results) return set_error(NTR_ERR_INVALID, "ntr_count_hits: null results");
    hipStream_t s = (hipStream_t)stream;
    int* d_cnt = nullptr;
    NTR_HIP(hipMalloc((void**)&d_cnt, sizeof(int)));
    NTR_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int), s));
    int blocks = (numRays + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(count_hits_kernel, dim3(blocks), dim3(256), 0, s, d_results, numRays, d_cnt);
    NTR_HIP(hipGetLastError());
    NTR_HIP(hipMemcpyAsync(count, d_cnt, sizeof(int), hipMemcpyDeviceToHost, s));
    NTR_HIP(hipStreamSynchronize(s));
    NTR_HIP(hipFree(d_cnt));
    return NTR_OK;
}

}  // extern "C"
