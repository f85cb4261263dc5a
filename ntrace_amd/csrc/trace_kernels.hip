// trace_kernels.hip -- CDNA4 (gfx950, wave64) BVH traversal kernels.
//
// Replaces the reference's `trace_bvh` kernels (contract TRACE_FUNC_BVH,
// src/rt/kernels/CudaTracerKernels.hpp:99-112) for BVHLayout_Compact:
//   fermi_speculative_while_while.cu:54-263   -> trace_bvh_perray   (one ray per lane)
//   tesla_persistent_while_while.cu:71-316,
//   kepler_dynamic_fetch.cu:61-322            -> trace_bvh_persistent (persistent waves,
//                                                 ballot/mbcnt refill, LDS stack)
//
// ARITHMETIC.  The hit records must be bit-exact against the reference's *CPU*
// tracer (CudaBVH::trace<BVHLayout_Compact>, src/rt/cuda/CudaBVH.cpp:698-784), so
// every decision reproduces its binary32 expressions, not the CUDA kernels':
//   slabs      (lo - o) / d, true IEEE division      (src/rt/Util.cpp:39-40)
//   min / max  selects (a<b)?a:b, folded x,y,z       (Defs.hpp:212-213, Math.hpp:146-147)
//   accept     tmin<=tmax && tmax>=ray.tmin && tmin<=ray.tmax   (CudaBVH.cpp:742-743)
//   order      near child = smaller tmin, ties -> child 0         (CudaBVH.cpp:761)
//   Woop       unfused left-to-right dots incl. the leading 0 and the w term
//              (Util.cpp:106-121, Math.hpp:185), 1.f/x then multiply
// This file is compiled with -ffp-contract=off and without fast-math; hipcc's
// default correctly-rounded f32 divide is relied on (checked by the parity tests).
//
// No MFMA: there is no dense contraction on this path.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "trace_kernels.h"

namespace ntr {

static constexpr int kSentinel = 0x76543210;  // CudaTracerKernels.hpp:38 (EntrypointSentinel)

__device__ __forceinline__ float sel_min(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float sel_max(float a, float b) { return (a > b) ? a : b; }

struct RayRegs {
    float ox, oy, oz, tmin;
    float dx, dy, dz, tmax;  // tmax shrinks to the closest accepted t (CudaBVH.cpp:1215)
};

// Intersect::RayBox for one child (Util.cpp:34-46), exact form.
__device__ __forceinline__ void ray_box(const RayRegs& r, float lox, float hix, float loy, float hiy,
                                        float loz, float hiz, float& tmn, float& tmx)
{
    float t0x = (lox - r.ox) / r.dx, t1x = (hix - r.ox) / r.dx;
    float t0y = (loy - r.oy) / r.dy, t1y = (hiy - r.oy) / r.dy;
    float t0z = (loz - r.oz) / r.dz, t1z = (hiz - r.oz) / r.dz;
    tmn = sel_max(sel_max(sel_min(t0x, t1x), sel_min(t0y, t1y)), sel_min(t0z, t1z));
    tmx = sel_min(sel_min(sel_max(t0x, t1x), sel_max(t0y, t1y)), sel_max(t0z, t1z));
}

// dot(Vec4f a, Vec4f(b,bw)) as Math.hpp:185: r = 0; r += a[i]*b[i].
__device__ __forceinline__ float dot4(float4 a, float bx, float by, float bz, float bw)
{
    float r = 0.0f;
    r += a.x * bx;
    r += a.y * by;
    r += a.z * bz;
    r += a.w * bw;
    return r;
}

// Per-lane traversal stack: the first kLdsDepth entries live in LDS laid out
// [entry][lane] (bank = lane % 32 whatever the per-lane depth -> conflict-free,
// MI355X_MICROARCH LDS table), deeper entries spill to scratch.  The reference CPU
// stack holds 100 entries (CudaBVH.cpp:701); SAH trees are at most 64 deep
// (SAHBVHBuilder.hpp MaxDepth) and LBVH trees at most 30 inner levels.
template <int LDS_DEPTH, int SPILL_DEPTH>
struct LaneStack {
    int* lds;               // &s_stack[wave][0][lane]
    int  spill[SPILL_DEPTH];
    int  sp;

    __device__ __forceinline__ void push(int v, unsigned int* status)
    {
        if (sp < LDS_DEPTH) lds[sp * 64] = v;
        else if (sp < LDS_DEPTH + SPILL_DEPTH) spill[sp - LDS_DEPTH] = v;
        else { atomicOr(status, NTR_STATUS_STACK_OVERFLOW); return; }
        sp++;
    }
    __device__ __forceinline__ int pop()
    {
        sp--;
        return (sp < LDS_DEPTH) ? lds[sp * 64] : spill[sp - LDS_DEPTH];
    }
};

// One inner-node step of trace<BVHLayout_Compact> (CudaBVH.cpp:721-775) for one lane.
struct LaneStats {
    unsigned int inner, tris, leaves;
};

template <class Stack>
__device__ __forceinline__ void inner_step(const char* __restrict__ nodes, const RayRegs& r,
                                           int& node, Stack& st, unsigned int* status)
{
    const float4* n = reinterpret_cast<const float4*>(nodes + (size_t)(unsigned)node);
    const float4 n0 = n[0];  // c0.lo.x c0.hi.x c0.lo.y c0.hi.y
    const float4 n1 = n[1];  // c1.lo.x c1.hi.x c1.lo.y c1.hi.y
    const float4 nz = n[2];  // c0.lo.z c0.hi.z c1.lo.z c1.hi.z
    const int4   nc = reinterpret_cast<const int4*>(n)[3];

    float mn0, mx0, mn1, mx1;
    ray_box(r, n0.x, n0.y, n0.z, n0.w, nz.x, nz.y, mn0, mx0);
    ray_box(r, n1.x, n1.y, n1.z, n1.w, nz.z, nz.w, mn1, mx1);

    const bool i0 = (mn0 <= mx0) && (mx0 >= r.tmin) && (mn0 <= r.tmax);
    const bool i1 = (mn1 <= mx1) && (mx1 >= r.tmin) && (mn1 <= r.tmax);

    int c0 = nc.x, c1 = nc.y;
    if (i0 && i1) {
        if (mn0 > mn1) { int t = c0; c0 = c1; c1 = t; }
        node = c0;
        st.push(c1, status);
    } else if (i0) {
        node = c0;
    } else if (i1) {
        node = c1;
    } else {
        node = st.pop();
    }
}

// intersectTriangles<BVHLayout_Compact> + updateHit (CudaBVH.cpp:1084-1126, 1183-1225).
// Returns true when an any-hit ray terminates.
template <bool STATS = false>
__device__ __forceinline__ bool leaf_step(const float4* __restrict__ woop, RayRegs& r, int leaf,
                                          bool anyHit, int& hitAddr, float& hitU, float& hitV,
                                          LaneStats* ls = nullptr)
{
    for (int triAddr = ~leaf;; triAddr += 3) {
        const float4 z = woop[triAddr];
        if (__float_as_uint(z.x) == 0x80000000u) {  // terminator (CudaBVH.cpp:1091)
            if (STATS) ls->leaves++;
            break;
        }
        if (STATS) ls->tris++;  // numTriangleTests (CudaBVH.cpp:1107-1111)
        const float4 u4 = woop[triAddr + 1];
        const float4 v4 = woop[triAddr + 2];

        // Intersect::RayTriangleWoop (Util.cpp:99-127)
        const float Oz = z.w - r.ox * z.x - r.oy * z.y - r.oz * z.z;
        const float ooDz = 1.0f / dot4(z, r.dx, r.dy, r.dz, 0.0f);
        const float t = Oz * ooDz;
        float tt = FLT_MAX, uu = 0.0f, vv = 0.0f;  // miss -> bary[2] = FW_F32_MAX
        if (t > r.tmin && t < r.tmax) {
            const float u = dot4(u4, r.ox, r.oy, r.oz, 1.0f) + t * dot4(u4, r.dx, r.dy, r.dz, 0.0f);
            if (u >= 0.0f) {
                const float v = dot4(v4, r.ox, r.oy, r.oz, 1.0f) + t * dot4(v4, r.dx, r.dy, r.dz, 0.0f);
                if (v >= 0.0f && (u + v) <= 1.0f) { tt = t; uu = u; vv = v; }
            }
        }
        // updateHit re-tests the returned t, so with tmax = +inf a *missed* test
        // is recorded at t = FLT_MAX exactly like the reference (CudaBVH.cpp:1200).
        if (tt > r.tmin && tt < r.tmax) {
            r.tmax = tt;
            hitAddr = triAddr;
            hitU = uu;
            hitV = vv;
            if (anyHit) return true;
        }
    }
    return false;
}

__device__ __forceinline__ void store_result(NtrRayResult* __restrict__ results, const int* __restrict__ triIndex,
                                             int rayIdx, int hitAddr, float t, float u, float v)
{
    int4 out;
    out.x = (hitAddr < 0) ? -1 : triIndex[hitAddr];
    out.y = __float_as_int(t);
    out.z = (hitAddr < 0) ? 0 : __float_as_int(u);
    out.w = (hitAddr < 0) ? 0 : __float_as_int(v);
    reinterpret_cast<int4*>(results)[rayIdx] = out;
}

// ---------------------------------------------------------------------------------
// Variant 1: one ray per lane, while-while ("fermi_speculative_while_while" slot).
// ---------------------------------------------------------------------------------
template <int WAVES, bool STATS>
__global__ __launch_bounds__(WAVES * 64) void trace_bvh_perray(TraceParams p)
{
    constexpr int LDS_DEPTH = 16, SPILL_DEPTH = 88;
    __shared__ int s_stack[WAVES][LDS_DEPTH][64];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rayIdx = blockIdx.x * (WAVES * 64) + threadIdx.x;
    if (rayIdx >= p.numRays) return;

    const float4 o = reinterpret_cast<const float4*>(p.rays)[rayIdx * 2 + 0];
    const float4 d = reinterpret_cast<const float4*>(p.rays)[rayIdx * 2 + 1];
    RayRegs r = {o.x, o.y, o.z, o.w, d.x, d.y, d.z, d.w};

    LaneStack<LDS_DEPTH, SPILL_DEPTH> st;
    st.lds = &s_stack[wave][0][lane];
    st.sp = 0;
    st.push(kSentinel, p.status);

    int hitAddr = -1;
    float hitU = 0.0f, hitV = 0.0f;
    // No triangle can be accepted unless tmin < tmax (t>tmin && t<tmax), so a
    // degenerate ray (Ray::degenerate, Util.hpp:65) is a miss without traversal.
    int node = (r.tmin < r.tmax) ? 0 : kSentinel;
    const char* nodes = reinterpret_cast<const char*>(p.nodes);
    const float4* woop = reinterpret_cast<const float4*>(p.woop);

    LaneStats ls = {0u, 0u, 0u};
    while (node != kSentinel) {
        while ((unsigned)node < (unsigned)kSentinel) {
            inner_step(nodes, r, node, st, p.status);
            if (STATS) ls.inner++;
        }
        if (node < 0) {
            if (leaf_step<STATS>(woop, r, node, p.anyHit != 0, hitAddr, hitU, hitV, &ls)) break;
            node = st.pop();
        }
    }
    store_result(p.results, p.triIndex, rayIdx, hitAddr, r.tmax, hitU, hitV);
    if (STATS) {
        // diagnostics variant only: plain per-lane atomics
        atomicAdd(&p.stats[0], (unsigned long long)ls.inner);
        atomicAdd(&p.stats[1], (unsigned long long)ls.tris);
        atomicAdd(&p.stats[2], (unsigned long long)ls.leaves);
        atomicAdd(&p.stats[3], (unsigned long long)(hitAddr >= 0));
    }
}

// ---------------------------------------------------------------------------------
// Variant 2: persistent waves.  Each wave owns a chunk [next,end) of the ray index
// space taken from one global counter (one returning atomic per chunk, lane 0);
// terminated lanes are refilled from the chunk by ballot + mbcnt prefix
// (kepler_dynamic_fetch.cu:97-111 on wave64: 64-bit ballot, v_mbcnt_lo/hi).
// ---------------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void trace_bvh_persistent(TraceParams p)
{
    constexpr int LDS_DEPTH = 16, SPILL_DEPTH = 88;
    __shared__ int s_stack[WAVES][LDS_DEPTH][64];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* nodes = reinterpret_cast<const char*>(p.nodes);
    const float4* woop = reinterpret_cast<const float4*>(p.woop);
    const bool anyHit = p.anyHit != 0;

    LaneStack<LDS_DEPTH, SPILL_DEPTH> st;
    st.lds = &s_stack[wave][0][lane];
    st.sp = 0;

    RayRegs r = {0, 0, 0, 0, 0, 0, 0, 0};
    int node = kSentinel, rayIdx = -1, hitAddr = -1;
    float hitU = 0.0f, hitV = 0.0f;
    int chunkNext = 0, chunkEnd = 0;  // wave-uniform
    bool poolEmpty = false;           // wave-uniform

    // Invariant at the top of the loop: a lane either holds a live ray
    // (rayIdx >= 0, node != sentinel) or is empty (rayIdx < 0, node == sentinel).
    for (;;) {
        // ---- refill empty lanes from the wave's chunk ----------------------------
        unsigned long long empty = __ballot(rayIdx < 0);
        while (empty != 0ull && !poolEmpty) {
            if (chunkNext >= chunkEnd) {  // wave-uniform: grab the next chunk
                int base = 0;
                if (lane == 0) base = atomicAdd(p.counter, p.chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= p.numRays) { poolEmpty = true; break; }
                chunkNext = base;
                chunkEnd = min(base + p.chunk, p.numRays);
            }
            // rank of this lane among the empty lanes (wave64 prefix popcount)
            const int prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(empty >> 32),
                               __builtin_amdgcn_mbcnt_lo((unsigned)empty, 0));
            const int avail = chunkEnd - chunkNext;
            if (rayIdx < 0 && prefix < avail) {
                rayIdx = chunkNext + prefix;
                const float4 o = reinterpret_cast<const float4*>(p.rays)[rayIdx * 2 + 0];
                const float4 d = reinterpret_cast<const float4*>(p.rays)[rayIdx * 2 + 1];
                r = {o.x, o.y, o.z, o.w, d.x, d.y, d.z, d.w};
                hitAddr = -1;
                hitU = hitV = 0.0f;
                st.sp = 0;
                st.push(kSentinel, p.status);
                // tmin < tmax is necessary for any accept (t>tmin && t<tmax):
                // degenerate rays (Util.hpp:65) are misses without traversal.
                node = (r.tmin < r.tmax) ? 0 : kSentinel;
            }
            chunkNext += min(__popcll(empty), avail);
            empty = __ballot(rayIdx < 0);
        }

        // ---- while-while traversal ------------------------------------------------
        while (node != kSentinel) {
            while ((unsigned)node < (unsigned)kSentinel)
                inner_step(nodes, r, node, st, p.status);
            if (node < 0) {
                if (leaf_step(woop, r, node, anyHit, hitAddr, hitU, hitV)) node = kSentinel;
                else node = st.pop();
            }
            // dynamic fetch (kepler_dynamic_fetch.cu:310): too few live lanes while
            // rays remain in the pool -> leave the loop and refill the idle lanes.
            if (!poolEmpty && __popcll(__ballot(true)) < p.fetchThreshold) break;
        }

        // ---- retire finished rays ---------------------------------------------------
        if (rayIdx >= 0 && node == kSentinel) {
            store_result(p.results, p.triIndex, rayIdx, hitAddr, r.tmax, hitU, hitV);
            rayIdx = -1;
        }
        if (poolEmpty && __ballot(rayIdx >= 0) == 0ull) break;
    }
}

}  // namespace ntr

// ---- host-side launchers (called from ntr_api.cpp) -----------------------------------
extern "C" hipError_t ntr_launch_trace(int variant, const ntr::TraceParams* p, int numBlocks, hipStream_t stream)
{
    constexpr int WAVES = NTR_TRACE_WAVES_PER_BLOCK;
    switch (variant) {
    case NTR_VARIANT_PERRAY:
        hipLaunchKernelGGL((ntr::trace_bvh_perray<WAVES, false>), dim3(numBlocks), dim3(WAVES * 64), 0, stream, *p);
        break;
    case NTR_VARIANT_PERRAY_STATS:
        hipLaunchKernelGGL((ntr::trace_bvh_perray<WAVES, true>), dim3(numBlocks), dim3(WAVES * 64), 0, stream, *p);
        break;
    case NTR_VARIANT_PERSISTENT:
        hipLaunchKernelGGL(ntr::trace_bvh_persistent<WAVES>, dim3(numBlocks), dim3(WAVES * 64), 0, stream, *p);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
