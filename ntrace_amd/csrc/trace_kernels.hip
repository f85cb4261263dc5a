// trace_kernels.hip -- CDNA4 (gfx950, wave64) BVH traversal kernels.
//
// Replaces the reference's `trace_bvh` kernels (contract TRACE_FUNC_BVH,
// src/rt/kernels/CudaTracerKernels.hpp:99-112) for BVHLayout_Compact:
//   fermi_speculative_while_while.cu:54-263   -> trace_bvh_perray     (one ray per lane)
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
// Compiled with -ffp-contract=off and without fast-math.
//
// Two code paths compute the slab test, both exact:
//   GENERIC  `/` (hipcc's correctly rounded f32 divide: v_div_scale / v_rcp / fma chain /
//            v_div_fmas / v_div_fixup) and select-form min/max.  Valid for every input
//            (zero direction components, NaN, infinities, denormals).
//   FAST     for waves whose rays are all "nice" (see ray_is_nice) over a BVH flagged
//            NTR_BVH_FASTDIV: in that range nothing over- or underflows and v_div_scale never
//            rescales, so a quotient is  r = RN(1/d)  (the IEEE divide, once per ray and axis) and,
//            per quotient,  q0 = x*r;  e = fma(-d,q0,x);  q = fma(e,r,q0)  -- three operations.
//            With the CORRECTLY ROUNDED reciprocal one residual correction gives RN(x/d), the
//            GENERIC path's bits (exact_rcp below: why, and how every quotient that could differ
//            was checked).  Rounds 1-3 used the hardware divide's own chain instead -- v_rcp
//            refined once, which is not always RN(1/d), and therefore TWO corrections: five
//            operations per quotient, sixty of the ~100 vector instructions of an inner-node step.
//            No NaN/inf can arise in the range either, so v_min3/v_max3 equal the select-form
//            folds up to the sign of zero, which no later comparison can observe.
//   ntr_selftest_division() / ntr_selftest_division_hard() check FAST == GENERIC bit for bit on the device.
//
// DATA PATH.  The while-while loop (traverse) fetches nodes and Woop triangles with buffer
// loads through wave-uniform resource descriptors (voffset = the Compact layout's own byte
// offsets, so no 64-bit address arithmetic; out-of-range reads return 0 instead of faulting,
// which lets a leaf fetch its triangle and the following terminator word in one round trip).
// The unified-step loop (traverse_unified: every live lane advances by one node OR one
// triangle per iteration) fetches 64 bytes per lane with ONE group of global loads from the
// lane's own buffer, descriptor loads only for the lanes within 64 bytes of a buffer's end.
// The traversal stack lives in LDS ([entry][lane], conflict-free), spilling to scratch
// beyond LDS_DEPTH.  No MFMA: there is no dense contraction on this path.
//
// SCHEDULING (never a ray's own visiting order, hence never a hit record or a counter):
//   * a wave leaves its inner-node loop early when few lanes still hold an inner node (leafSwitchBelow);
//   * trace_bvh_perray maps workgroup i to ray block order[i] when an order is given -- predicted
//     (sched_kernels.hip, automatic for large closest-hit launches) or learned from the previous launch of
//     the batch (NtrSchedHint: per-block cost recording here, sched_order_kernel below);
//   * the closest-hit instantiation of trace_bvh_perray runs a batch as wave-private mini-pools (minipool_body: a
//     wave owns K 64-ray chunks of the dispatch order and refills its finished lanes from them) when the pool
//     depth K the device derived for the batch (coherence estimate in sched_kernels.hip) is above 1;
//   * every per-launch counter is cleared by a kernel, so an asynchronous launch can be captured in a HIP
//     graph and replayed.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "trace_kernels.h"
#include "trace_arith.h"

namespace ntr {

static constexpr int kSentinel = 0x76543210;  // CudaTracerKernels.hpp:38 (EntrypointSentinel)
static constexpr int LDS_DEPTH = 16;
static constexpr int SPILL_DEPTH = 88;        // 16 + 88 >= the reference CPU stack of 100 (CudaBVH.cpp:701)

typedef __amdgpu_buffer_rsrc_t Rsrc;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ Rsrc make_rsrc(const void* p, unsigned int bytes)
{
    // built from kernel arguments only -> provably wave-uniform (no waterfall loops)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld4(Rsrc r, int byteOfs)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byteOfs, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ unsigned int ld1(Rsrc r, int byteOfs)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, byteOfs, 0, 0);
}


__device__ __forceinline__ float sel_min(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float sel_max(float a, float b) { return (a > b) ? a : b; }

struct RayRegs {
    float ox, oy, oz, tmin;
    float dx, dy, dz, tmax;  // tmax shrinks to the closest accepted t (CudaBVH.cpp:1215)
    float rx, ry, rz;        // FAST path: correctly rounded reciprocals of dx,dy,dz
};

// ---- FAST-path preconditions ---------------------------------------------------------
// FAST-path ranges.  Directions: 2^-40 <= |d| <= 2^20.  Box coordinates: |c| < 2^55 (BVH flag
// NTR_BVH_FASTDIV).  Ray origin components: 2^-36 <= |o| < 2^55 -- then x = c - o is 0 or
// |x| >= 2^-84 for ANY such c (a c much smaller than o leaves x = -o; otherwise both operands
// are >= 2^-61 and a non-zero difference is at least one ulp of that).  An origin component
// that is exactly 0 makes x = c, which is only safe when the BVH has no tiny coordinates
// (flag NTR_BVH_NOTINY: c == 0 or |c| >= 2^-93).  In these ranges |x| < 2^56,
// exponent(x) - exponent(d) < 96, |x| >= 2^-103 and |x/d| >= 2^-113: none of v_div_scale's
// rescaling cases, and every residual of the fma chain is exactly representable.
__device__ __forceinline__ bool nice_dir(float v) { const float a = fabsf(v); return a >= 0x1p-40f && a <= 0x1p20f; }
__device__ __forceinline__ bool nice_pos(float v, bool zeroOk)
{
    const float a = fabsf(v);
    return (a >= 0x1p-36f && a < 0x1p55f) || (zeroOk && v == 0.0f);
}
__device__ __forceinline__ bool ray_is_nice(const RayRegs& r, uint32_t bvhFlags)
{
    const bool zeroOk = (bvhFlags & NTR_BVH_NOTINY) != 0;
    return nice_dir(r.dx) && nice_dir(r.dy) && nice_dir(r.dz) && nice_pos(r.ox, zeroOk) && nice_pos(r.oy, zeroOk) &&
           nice_pos(r.oz, zeroOk);
}
// Intersect::RayBox for BOTH children of a node (Util.cpp:34-46).  The FAST form evaluates
// the twelve quotients stage by stage (all q0, then all e1, ...) so that consecutive
// instructions are independent: a lone wave cannot issue a VALU op that depends on the
// previous one back to back.
// OCT < 8 (FAST only): every live ray of the wave has direction signs OCT (bit 0: dx < 0, bit 1: dy < 0, bit 2: dz < 0) and every box
// has lo <= hi (NTR_BVH_ORDERED).  Rounding is monotone, so (lo - o) / d <= (hi - o) / d for d > 0 and >= for d < 0: the smaller
// quotient of a slab is known without comparing -- the same value min / max would pick, six instructions per child less.
template <bool FAST, int OCT = 8>
__device__ __forceinline__ void ray_box2(const RayRegs& r, const float4& n0, const float4& n1, const float4& nz,
                                         float& mn0, float& mx0, float& mn1, float& mx1)
{
    if (FAST) {
        // x[k] = plane - origin ; axis of slot k: x x y y z z (child 0), x x y y z z (child 1)
        float x[12] = {n0.x - r.ox, n0.y - r.ox, n0.z - r.oy, n0.w - r.oy, nz.x - r.oz, nz.y - r.oz,
                       n1.x - r.ox, n1.y - r.ox, n1.z - r.oy, n1.w - r.oy, nz.z - r.oz, nz.w - r.oz};
        const float d[3] = {r.dx, r.dy, r.dz};
        const float rc[3] = {r.rx, r.ry, r.rz};
        float q[12], e[12];
#pragma unroll
        for (int k = 0; k < 12; k++) q[k] = x[k] * rc[(k % 6) >> 1];
#pragma unroll
        for (int k = 0; k < 12; k++) e[k] = __builtin_fmaf(-d[(k % 6) >> 1], q[k], x[k]);
#pragma unroll
        for (int k = 0; k < 12; k++) q[k] = __builtin_fmaf(e[k], rc[(k % 6) >> 1], q[k]);
        if (OCT < 8) {
            constexpr int sx = OCT & 1, sy = (OCT >> 1) & 1, sz = (OCT >> 2) & 1;   // 1: the hi plane is the near one
            mn0 = fmaxf(fmaxf(q[0 + sx], q[2 + sy]), q[4 + sz]);
            mx0 = fminf(fminf(q[1 - sx], q[3 - sy]), q[5 - sz]);
            mn1 = fmaxf(fmaxf(q[6 + sx], q[8 + sy]), q[10 + sz]);
            mx1 = fminf(fminf(q[7 - sx], q[9 - sy]), q[11 - sz]);
        } else {
            mn0 = fmaxf(fmaxf(fminf(q[0], q[1]), fminf(q[2], q[3])), fminf(q[4], q[5]));
            mx0 = fminf(fminf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])), fmaxf(q[4], q[5]));
            mn1 = fmaxf(fmaxf(fminf(q[6], q[7]), fminf(q[8], q[9])), fminf(q[10], q[11]));
            mx1 = fminf(fminf(fmaxf(q[6], q[7]), fmaxf(q[8], q[9])), fmaxf(q[10], q[11]));
        }
    } else {
        float t0x = (n0.x - r.ox) / r.dx, t1x = (n0.y - r.ox) / r.dx;
        float t0y = (n0.z - r.oy) / r.dy, t1y = (n0.w - r.oy) / r.dy;
        float t0z = (nz.x - r.oz) / r.dz, t1z = (nz.y - r.oz) / r.dz;
        mn0 = sel_max(sel_max(sel_min(t0x, t1x), sel_min(t0y, t1y)), sel_min(t0z, t1z));
        mx0 = sel_min(sel_min(sel_max(t0x, t1x), sel_max(t0y, t1y)), sel_max(t0z, t1z));
        t0x = (n1.x - r.ox) / r.dx; t1x = (n1.y - r.ox) / r.dx;
        t0y = (n1.z - r.oy) / r.dy; t1y = (n1.w - r.oy) / r.dy;
        t0z = (nz.z - r.oz) / r.dz; t1z = (nz.w - r.oz) / r.dz;
        mn1 = sel_max(sel_max(sel_min(t0x, t1x), sel_min(t0y, t1y)), sel_min(t0z, t1z));
        mx1 = sel_min(sel_min(sel_max(t0x, t1x), sel_max(t0y, t1y)), sel_max(t0z, t1z));
    }
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

// Per-lane traversal stack: entries [0, LDS_DEPTH) in LDS laid out [entry][lane] (bank =
// lane % 32 whatever the per-lane depth -> conflict-free), deeper entries in a scratch array
// that only the (rare) overflow branches touch.  `sp` and the LDS base stay in registers.
typedef __attribute__((address_space(3))) int lds_int;

struct LaneStack {
    lds_int* lds;  // &s_stack[wave][0][lane]
    int sp;        // entries held in memory (LDS, then scratch)
    int tos;       // top of the stack, kept in a register: a pop hands out the next node without
                   // waiting for LDS; the entry below it is fetched off the critical path
};

#define NTR_STACK_RESET(st) do { (st).sp = 0; (st).tos = kSentinel; } while (0)

template <int LD = LDS_DEPTH>   // LD: the entries this stack has in LDS
__device__ __forceinline__ void stack_push(LaneStack& st, int (&spill)[SPILL_DEPTH], int v, unsigned int* status)
{
    if (__builtin_expect(st.sp < LD, 1)) st.lds[st.sp * 64] = st.tos;
    else if (st.sp < LD + SPILL_DEPTH) spill[st.sp - LD] = st.tos;
    else { atomicOr(status, NTR_STATUS_STACK_OVERFLOW); return; }
    st.sp++;
    st.tos = v;
}
template <int LD = LDS_DEPTH>
__device__ __forceinline__ int stack_pop(LaneStack& st, int (&spill)[SPILL_DEPTH])
{
    const int r = st.tos;
    if (st.sp > 0) {
        st.sp--;
        st.tos = __builtin_expect(st.sp < LD, 1) ? st.lds[st.sp * 64] : spill[st.sp - LD];
    } else {
        st.tos = kSentinel;
    }
    return r;
}

// Keeps a loaded value live at this point so that hipcc cannot sink its load into a later
// conditional block (which would turn one memory round trip per node into two).
__device__ __forceinline__ void keep(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void keep(unsigned int& v) { asm volatile("" : "+v"(v)); }

struct LaneStats {
    unsigned int inner, tris, leaves;
};

static constexpr int kNoNode = (int)0xFFFFFF00u;  // buffer offset beyond any extent (< 4 GiB)

// One inner-node step of trace<BVHLayout_Compact> (CudaBVH.cpp:721-775).  Executed by the whole
// wave; only lanes whose current node is an inner node (`inner`) update their state.
template <bool FAST, int OCT = 8>
__device__ __forceinline__ void inner_step(Rsrc nodes, bool inner, const RayRegs& r,
                                           int& node, LaneStack& st, int (&spill)[SPILL_DEPTH], unsigned int* status)
{
    // every lane fetches its own node (4 x 16 B; the quad-cooperative LDS-DMA fetch of rounds 1-2 was 1.2-3.7x slower in this loop:
    // scripts/studies/rejected_patches/coop_fetch.patch)
    const int ofs = inner ? node : kNoNode;
    const float4 n0 = ld4(nodes, ofs), n1 = ld4(nodes, ofs + 16), nz = ld4(nodes, ofs + 32);
    float4 nc = ld4(nodes, ofs + 48);   // (an 8-byte load of the two child words alone: 0.9 % slower, profiles/r03_ab_child_load_b64.jsonl)
    keep(nc);

    float mn0, mx0, mn1, mx1;
    ray_box2<FAST, OCT>(r, n0, n1, nz, mn0, mx0, mn1, mx1);

    const bool i0 = (mn0 <= mx0) && (mx0 >= r.tmin) && (mn0 <= r.tmax);
    const bool i1 = (mn1 <= mx1) && (mx1 >= r.tmin) && (mn1 <= r.tmax);

    const int c0 = __float_as_int(nc.x), c1 = __float_as_int(nc.y);
    const bool swp = i1 && (!i0 || mn0 > mn1);  // visit c1 first (ties -> c0, CudaBVH.cpp:761)
    const int nearC = swp ? c1 : c0;
    const int farC = swp ? c0 : c1;
    if (inner) {
        if (i0 && i1) stack_push(st, spill, farC, status);
        node = (i0 || i1) ? nearC : stack_pop(st, spill);
    }
}

// intersectTriangles<BVHLayout_Compact> + updateHit (CudaBVH.cpp:1084-1126, 1183-1225).
// Returns true when an any-hit ray terminates.
template <bool STATS>
__device__ __forceinline__ bool leaf_step(Rsrc woop, RayRegs& r, int leaf, bool anyHit, int& hitAddr,
                                          float& hitU, float& hitV, LaneStats& ls)
{
    for (int ofs = (~leaf) * 16;; ofs += 48) {
        const float4 z = ld4(woop, ofs);
        float4 u4 = ld4(woop, ofs + 16);         // past a terminator these may run off the
        float4 v4 = ld4(woop, ofs + 32);         // buffer: range-checked loads return 0
        unsigned int nextWord = ld1(woop, ofs + 48);
        keep(u4); keep(v4); keep(nextWord);      // one round trip per triangle, not four
        if (__float_as_uint(z.x) == 0x80000000u) {  // terminator (CudaBVH.cpp:1091)
            if (STATS) ls.leaves++;
            break;
        }
        if (STATS) ls.tris++;  // numTriangleTests (CudaBVH.cpp:1107-1111)

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
            hitAddr = ofs >> 4;
            hitU = uu;
            hitV = vv;
            if (anyHit) return true;
        }
        if (nextWord == 0x80000000u) {  // the terminator was fetched with this triangle
            if (STATS) ls.leaves++;
            break;
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

__device__ __forceinline__ void load_ray(const NtrRay* __restrict__ rays, int rayIdx, RayRegs& r)
{
    const float4 o = reinterpret_cast<const float4*>(rays)[rayIdx * 2 + 0];
    const float4 d = reinterpret_cast<const float4*>(rays)[rayIdx * 2 + 1];
    r.ox = o.x; r.oy = o.y; r.oz = o.z; r.tmin = o.w;
    r.dx = d.x; r.dy = d.y; r.dz = d.z; r.tmax = d.w;
    r.rx = exact_rcp(d.x); r.ry = exact_rcp(d.y); r.rz = exact_rcp(d.z);
}

// While-while traversal of the lanes' current rays until every lane is done (or, in the
// persistent kernel, until too few lanes are live).  Both loops are wave-uniform (ballots); per-ray
// visiting order is exactly the CPU tracer's depth-first order, whatever the other lanes do.
// SLICED (persistent kernels): the loop also ends after `slice` rounds of it (`slice` counts down; the caller looks at the wave -- posts the
// dequeue of its next chunk -- and calls again).
template <bool FAST, bool STATS, bool DYNAMIC_FETCH, int OCT = 8, bool SLICED = false>
__device__ __forceinline__ void traverse(Rsrc nodes, Rsrc woop, RayRegs& r, int& node,
                                         LaneStack& st, int (&spill)[SPILL_DEPTH], bool anyHit,
                                         int& hitAddr, float& hitU, float& hitV, LaneStats& ls, unsigned int* status,
                                         bool poolEmpty, int fetchThreshold, int leafSwitchBelow, int& slice)
{
    unsigned long long live = __ballot(node != kSentinel);
    while (live != 0ull) {
        if (SLICED && --slice < 0) break;
        for (;;) {
            const bool inner = (unsigned)node < (unsigned)kSentinel;
            const unsigned long long innerMask = __ballot(inner);
            if (innerMask == 0ull) break;
            // Phase switch policy (affects scheduling only, never a ray's own visiting order): when few
            // lanes still hold an inner node while others already wait at a leaf, serve the leaves first
            // instead of letting a handful of stragglers stall the wave.
            if (__popcll(innerMask) < leafSwitchBelow && __ballot(node < 0) != 0ull) break;
            inner_step<FAST, OCT>(nodes, inner, r, node, st, spill, status);
            if (STATS && inner) ls.inner++;
        }
        if (node < 0) {
            if (leaf_step<STATS>(woop, r, node, anyHit, hitAddr, hitU, hitV, ls)) node = kSentinel;
            else node = stack_pop(st, spill);
        }
        live = __ballot(node != kSentinel);
        // dynamic fetch (kepler_dynamic_fetch.cu:310): too few live lanes while rays remain
        // in the pool -> leave the loop so that the idle lanes are refilled.
        if (DYNAMIC_FETCH && !poolEmpty && __popcll(live) < fetchThreshold) break;
    }
}
template <bool FAST, bool STATS, bool DYNAMIC_FETCH, int OCT = 8>
__device__ __forceinline__ void traverse(Rsrc nodes, Rsrc woop, RayRegs& r, int& node,
                                         LaneStack& st, int (&spill)[SPILL_DEPTH], bool anyHit,
                                         int& hitAddr, float& hitU, float& hitV, LaneStats& ls, unsigned int* status,
                                         bool poolEmpty, int fetchThreshold, int leafSwitchBelow)
{
    int never = 0;
    traverse<FAST, STATS, DYNAMIC_FETCH, OCT, false>(nodes, woop, r, node, st, spill, anyHit, hitAddr, hitU, hitV, ls, status, poolEmpty, fetchThreshold,
                                                     leafSwitchBelow, never);
}

// Unified-step traversal (the dynamic-fetch kernel's loop).  The while-while loop above lets a wave alternate between an
// inner-node phase and a leaf phase, and a leaf of k triangles costs k dependent round trips during which the lanes that hold
// inner nodes idle: on divergent batches (diffuse / incoherent rays in LBVH trees with 6-triangle leaves) only 9-20 % of the
// lane slots of a wave iteration do work (profiles/r03_divergence_*).  Here EVERY live lane advances by one step per iteration,
// whatever it holds: an inner node (64 B from `nodes`) or the next triangle of its leaf (48 B + the following word from
// `woop`: also 64 contiguous bytes).  Both fetches are issued before the wave waits, so an iteration is ONE memory round trip.
// A lane's own visiting order -- and with it every hit record -- is exactly that of traverse(): only the interleaving of the
// lanes changes.  `node` < 0 doubles as the triangle cursor: the lane's next triangle is at float4 index ~node (a leaf
// reference IS the index of its first triangle; advancing one triangle subtracts 3).
// Buffer resource descriptor as four scalar words (what make_rsrc builds): base, base_hi (stride 0), extent in bytes, flags.
__device__ __forceinline__ u32x4 rsrc_words(const void* p, unsigned int bytes)
{
    const unsigned long long a = (unsigned long long)p;
    u32x4 w;
    w.x = __builtin_amdgcn_readfirstlane((unsigned int)a);
    w.y = __builtin_amdgcn_readfirstlane((unsigned int)(a >> 32) & 0xFFFFu);
    w.z = __builtin_amdgcn_readfirstlane(bytes);
    w.w = 0x00020000u;
    return w;
}

// Lanes of maskA fetch 64 B at byte offset `ofs` of buffer A, lanes of maskB at `ofs` of buffer B (range-checked: beyond the extent
// a load returns 0 and touches no memory); the other lanes fetch nothing and their a..d are undefined.
__device__ __forceinline__ void fetch64_two_buffers(u32x4 rsrcA, u32x4 rsrcB, int ofs, unsigned long long maskA,
                                                    unsigned long long maskB, float4& a, float4& b, float4& c, float4& d)
{
    u32x4 va, vb, vc, vd;
    unsigned long long sav;
    asm volatile(
        "s_mov_b64 %[sav], exec\n\t"
        "s_and_b64 exec, %[sav], %[ma]\n\t"
        "buffer_load_dwordx4 %[a], %[ofs], %[ra], 0 offen\n\t"
        "buffer_load_dwordx4 %[b], %[ofs], %[ra], 0 offen offset:16\n\t"
        "buffer_load_dwordx4 %[c], %[ofs], %[ra], 0 offen offset:32\n\t"
        "buffer_load_dwordx4 %[d], %[ofs], %[ra], 0 offen offset:48\n\t"
        "s_and_b64 exec, %[sav], %[mb]\n\t"
        "buffer_load_dwordx4 %[a], %[ofs], %[rb], 0 offen\n\t"
        "buffer_load_dwordx4 %[b], %[ofs], %[rb], 0 offen offset:16\n\t"
        "buffer_load_dwordx4 %[c], %[ofs], %[rb], 0 offen offset:32\n\t"
        "buffer_load_dwordx4 %[d], %[ofs], %[rb], 0 offen offset:48\n\t"
        "s_mov_b64 exec, %[sav]\n\t"
        "s_waitcnt vmcnt(0)"
        : [a] "=&v"(va), [b] "=&v"(vb), [c] "=&v"(vc), [d] "=&v"(vd), [sav] "=&s"(sav)
        : [ofs] "v"(ofs), [ra] "s"(rsrcA), [rb] "s"(rsrcB), [ma] "s"(maskA), [mb] "s"(maskB)
        : "memory", "scc");   // (s_and_b64 writes SCC)
    a = make_float4(__uint_as_float(va.x), __uint_as_float(va.y), __uint_as_float(va.z), __uint_as_float(va.w));
    b = make_float4(__uint_as_float(vb.x), __uint_as_float(vb.y), __uint_as_float(vb.z), __uint_as_float(vb.w));
    c = make_float4(__uint_as_float(vc.x), __uint_as_float(vc.y), __uint_as_float(vc.z), __uint_as_float(vc.w));
    d = make_float4(__uint_as_float(vd.x), __uint_as_float(vd.y), __uint_as_float(vd.z), __uint_as_float(vd.w));
}

// The same loads INTO registers that already hold other lanes' data (read-write operands: lanes outside both masks keep theirs).
__device__ __forceinline__ void fetch64_two_buffers_into(u32x4 rsrcA, u32x4 rsrcB, int ofs, unsigned long long maskA,
                                                         unsigned long long maskB, float4& a, float4& b, float4& c, float4& d)
{
    u32x4 va = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)};
    u32x4 vb = {__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w)};
    u32x4 vc = {__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), __float_as_uint(c.w)};
    u32x4 vd = {__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), __float_as_uint(d.w)};
    unsigned long long sav;
    asm volatile(
        "s_mov_b64 %[sav], exec\n\t"
        "s_and_b64 exec, %[sav], %[ma]\n\t"
        "buffer_load_dwordx4 %[a], %[ofs], %[ra], 0 offen\n\t"
        "buffer_load_dwordx4 %[b], %[ofs], %[ra], 0 offen offset:16\n\t"
        "buffer_load_dwordx4 %[c], %[ofs], %[ra], 0 offen offset:32\n\t"
        "buffer_load_dwordx4 %[d], %[ofs], %[ra], 0 offen offset:48\n\t"
        "s_and_b64 exec, %[sav], %[mb]\n\t"
        "buffer_load_dwordx4 %[a], %[ofs], %[rb], 0 offen\n\t"
        "buffer_load_dwordx4 %[b], %[ofs], %[rb], 0 offen offset:16\n\t"
        "buffer_load_dwordx4 %[c], %[ofs], %[rb], 0 offen offset:32\n\t"
        "buffer_load_dwordx4 %[d], %[ofs], %[rb], 0 offen offset:48\n\t"
        "s_mov_b64 exec, %[sav]\n\t"
        "s_waitcnt vmcnt(0)"
        : [a] "+v"(va), [b] "+v"(vb), [c] "+v"(vc), [d] "+v"(vd), [sav] "=&s"(sav)
        : [ofs] "v"(ofs), [ra] "s"(rsrcA), [rb] "s"(rsrcB), [ma] "s"(maskA), [mb] "s"(maskB)
        : "memory", "scc");   // (s_and_b64 writes SCC)
    a = make_float4(__uint_as_float(va.x), __uint_as_float(va.y), __uint_as_float(va.z), __uint_as_float(va.w));
    b = make_float4(__uint_as_float(vb.x), __uint_as_float(vb.y), __uint_as_float(vb.z), __uint_as_float(vb.w));
    c = make_float4(__uint_as_float(vc.x), __uint_as_float(vc.y), __uint_as_float(vc.z), __uint_as_float(vc.w));
    d = make_float4(__uint_as_float(vd.x), __uint_as_float(vd.y), __uint_as_float(vd.z), __uint_as_float(vd.w));
}

// The two buffers of a unified fetch, as plain pointers + extents (FLAT = true: one set of four global loads for all live lanes) and as
// descriptor words (the range-checked two-buffer form: lanes whose 64 bytes would cross the end of their buffer, FLAT = false).
struct UnifiedBufs {
    const char* nodes; const char* woop;
    unsigned int nodesBytes, woopBytes;
    u32x4 rNodes, rWoop;
    bool uniformPrologue;   // per-ray kernels: scalar fetches while the wave's lanes all hold the same inner node (TraceParams::uniformPrologue)
    // FLAT fetch: both buffers lie inside one 4 GiB window (the host checks it before it selects the flat fetch), so a lane's 64 bytes
    // are base + a 32-bit offset -- the global load takes the scalar base and the lane's offset as they are, where two unrelated 64-bit
    // pointers cost every iteration a per-lane 64-bit select and add (round 5)
    const char* base;       // the lower of the two buffers
    unsigned int dN, dW;    // nodes - base, woop - base
    unsigned int limNode;   // largest inner-node offset whose 64 bytes lie inside the node buffer (below the sentinel: `node <= limNode` implies inner)
    int limTri;             // smallest (most negative) triangle cursor ~index whose 64 bytes lie inside triWoop
};
__device__ __forceinline__ UnifiedBufs unified_bufs(const TraceParams& p)
{
    UnifiedBufs u;
    u.nodes = (const char*)p.nodes; u.woop = (const char*)p.woop;
    u.nodesBytes = p.nodesBytes; u.woopBytes = p.woopBytes;
    u.rNodes = rsrc_words(p.nodes, p.nodesBytes); u.rWoop = rsrc_words(p.woop, p.woopBytes);
    u.uniformPrologue = p.uniformPrologue != 0;
    const unsigned long long an = (unsigned long long)p.nodes, aw = (unsigned long long)p.woop;
    const unsigned long long lo = an < aw ? an : aw;
    u.base = (const char*)lo;
    u.dN = (unsigned int)(an - lo); u.dW = (unsigned int)(aw - lo);
    u.limNode = p.nodesBytes - 64u;
    u.limTri = ~(int)((p.woopBytes - 64u) >> 4);
    return u;
}

// FLAT: the texture-address unit charges a wave-level load instruction about 16 cycles whatever its exec mask, so the two masked
// groups of fetch64_two_buffers cost 128 TA cycles per iteration and made the unified loop TA-bound (0.6-0.87 busy, profiles/r03v_*).
// With FLAT every live lane forms the 64-bit address of its own 64 bytes and ONE group of four global loads serves nodes and triangles
// alike (64 TA cycles).  Global loads are not range-checked: a lane whose 64 bytes would end beyond its buffer (an empty leaf's
// terminator in the last 48 bytes of triWoop; a malformed child offset) takes the descriptor path instead, which reads zeros there.
// One unified step, in two halves (the two-rays-per-lane experiment of round 5 stepped two rays per iteration with them: 24 % slower --
// 85 VGPRs, five waves per SIMD; scripts/studies/rejected_patches/two_rays_per_lane.patch, EXPERIMENTS.md).
// unified_fetch: one 64-byte fetch per lane from its own buffer -- the node of a lane at an inner node, the triangle (48 B + the following
// word) of a lane at a leaf.  Issues the loads and, apart from the rare end-of-buffer lanes, does not wait for them.
template <bool FLAT>
__device__ __forceinline__ void unified_fetch(const UnifiedBufs& ub, int node, float4& a, float4& b, float4& c, float4& d)
{
    const bool inner = (unsigned)node < (unsigned)kSentinel;
    const bool atTri = node < 0;
    // (Written as `inner ? ld4(nodes, ..) : ld4(woop, ..)` hipcc selects the descriptor per lane and wraps every load in a waterfall loop.)
    if (FLAT) {
        asm volatile("" : "=v"(a.x), "=v"(a.y), "=v"(a.z), "=v"(a.w), "=v"(b.x), "=v"(b.y), "=v"(b.z), "=v"(b.w));   // defined, whatever the lane
        asm volatile("" : "=v"(c.x), "=v"(c.y), "=v"(c.z), "=v"(c.w), "=v"(d.x), "=v"(d.y), "=v"(d.z), "=v"(d.w));
        const bool okNode = (unsigned)node <= ub.limNode, okTri = atTri && node >= ub.limTri;   // (extents are >= 64 here)
        const bool flatOk = okNode || okTri;
        const unsigned int cofs = okNode ? ub.dN + (unsigned)node : ub.dW + ((unsigned)(~node) << 4);   // from the scalar base: a 32-bit offset
        if (flatOk) {   // (global address space spelled out: the base comes out of integer arithmetic, and a generic pointer would be a flat_load)
            typedef const __attribute__((address_space(1))) u32x4* global_u4_ptr;
            const global_u4_ptr q = (global_u4_ptr)((const __attribute__((address_space(1))) char*)ub.base + cofs);
            const u32x4 qa = q[0], qb = q[1], qc = q[2], qd = q[3];
            a = make_float4(__uint_as_float(qa.x), __uint_as_float(qa.y), __uint_as_float(qa.z), __uint_as_float(qa.w));
            b = make_float4(__uint_as_float(qb.x), __uint_as_float(qb.y), __uint_as_float(qb.z), __uint_as_float(qb.w));
            c = make_float4(__uint_as_float(qc.x), __uint_as_float(qc.y), __uint_as_float(qc.z), __uint_as_float(qc.w));
            d = make_float4(__uint_as_float(qd.x), __uint_as_float(qd.y), __uint_as_float(qd.z), __uint_as_float(qd.w));
        }
        const unsigned long long odd = __ballot((inner || atTri) && !flatOk);
        if (odd != 0ull)   // rare: range-checked descriptor loads, into the same registers, for the lanes at the very end of a buffer
            fetch64_two_buffers_into(ub.rNodes, ub.rWoop, inner ? node : (~node) * 16, __ballot(inner && !flatOk), __ballot(atTri && !flatOk), a, b, c, d);
    } else {
        const int ofs = inner ? node : (~node) * 16;   // four loads under the inner lanes' mask and four under the triangle lanes' mask into the SAME registers, one wait
        fetch64_two_buffers(ub.rNodes, ub.rWoop, ofs, __ballot(inner), __ballot(atTri), a, b, c, d);
    }
}

// unified_advance: the lane's ray takes the step its 64 bytes allow -- one inner node (trace<BVHLayout_Compact>, CudaBVH.cpp:721-775) or one
// triangle (intersectTriangles + updateHit, CudaBVH.cpp:1084-1126, 1183-1225).
// one inner node of trace<BVHLayout_Compact> (CudaBVH.cpp:721-775): both child boxes, nearer child first (ties -> child 0), the other pushed
template <bool FAST, int OCT, int LD = LDS_DEPTH>
__device__ __forceinline__ void inner_advance(const float4& a, const float4& b, const float4& c, const float4& d, const RayRegs& r, int& node,
                                              LaneStack& st, int (&spill)[SPILL_DEPTH], unsigned int* status)
{
    float mn0, mx0, mn1, mx1;
    ray_box2<FAST, OCT>(r, a, b, c, mn0, mx0, mn1, mx1);
    const bool i0 = (mn0 <= mx0) && (mx0 >= r.tmin) && (mn0 <= r.tmax);
    const bool i1 = (mn1 <= mx1) && (mx1 >= r.tmin) && (mn1 <= r.tmax);
    const int c0 = __float_as_int(d.x), c1 = __float_as_int(d.y);
    const bool swp = i1 && (!i0 || mn0 > mn1);
    const int nearC = swp ? c1 : c0, farC = swp ? c0 : c1;
    if (i0 && i1) stack_push<LD>(st, spill, farC, status);
    node = (i0 || i1) ? nearC : stack_pop<LD>(st, spill);
}

template <bool FAST, int OCT, int LD = LDS_DEPTH>
__device__ __forceinline__ void unified_advance(const float4& a, const float4& b, const float4& c, const float4& d, RayRegs& r, int& node,
                                                LaneStack& st, int (&spill)[SPILL_DEPTH], bool anyHit, int& hitAddr, float& hitU, float& hitV,
                                                unsigned int* status)
{
    const bool inner = (unsigned)node < (unsigned)kSentinel;
    const bool atTri = node < 0;
    if (inner) {
        inner_advance<FAST, OCT, LD>(a, b, c, d, r, node, st, spill, status);
    } else if (atTri) {
        bool leafDone = __float_as_uint(a.x) == 0x80000000u;   // terminator: an empty leaf
        if (!leafDone) {
            const float Oz = a.w - r.ox * a.x - r.oy * a.y - r.oz * a.z;
            const float ooDz = 1.0f / dot4(a, r.dx, r.dy, r.dz, 0.0f);
            const float t = Oz * ooDz;
            float tt = FLT_MAX, uu = 0.0f, vv = 0.0f;
            if (t > r.tmin && t < r.tmax) {
                const float u = dot4(b, r.ox, r.oy, r.oz, 1.0f) + t * dot4(b, r.dx, r.dy, r.dz, 0.0f);
                if (u >= 0.0f) {
                    const float v = dot4(c, r.ox, r.oy, r.oz, 1.0f) + t * dot4(c, r.dx, r.dy, r.dz, 0.0f);
                    if (v >= 0.0f && (u + v) <= 1.0f) { tt = t; uu = u; vv = v; }
                }
            }
            bool terminated = false;
            if (tt > r.tmin && tt < r.tmax) {
                r.tmax = tt;
                hitAddr = ~node;
                hitU = uu;
                hitV = vv;
                terminated = anyHit;
            }
            if (terminated) node = kSentinel;
            else if (__float_as_uint(d.x) == 0x80000000u) leafDone = true;   // the terminator came with this triangle
            else node -= 3;
        }
        if (leafDone) node = stack_pop<LD>(st, spill);
    }
}

// Wave-uniform prologue (round 5).  The rays of a fresh wave all start at the root, and the rays of one wave -- an 8 x 8 pixel tile, or
// the AO samples of eight neighbouring pixels -- take the same way down the top of the tree: while every live lane holds the SAME inner
// node, that node is fetched ONCE through the scalar cache (s_load, no texture-path cycles: the per-lane fetch costs the TA 64 cycles per
// wave and iteration whatever the lanes hold) and the planes are scalar operands of the same arithmetic.  The loop ends for good at the
// first iteration in which the lanes disagree, or hold a leaf: the test (one v_readlane, one compare) is paid only while it succeeds --
// run on EVERY iteration it cost more than the fetches it saved (round 2), and looking again every 2 / 4 / 8 / 16 iterations of the
// general loop loses 1-4 % (profiles/r05_uniform_recheck_knob.txt): once apart, the lanes of a wave rarely all meet again.  Measured
// and left out as well: the same for a triangle every lane stands at (no gain, and 2.5 % lost to the larger loop:
// profiles/r05_uniform_prologue_levels_knob.txt), and the prologue after a persistent wave's refill (nothing).  Per-ray arithmetic, visiting order and
// stack are untouched: hit records cannot change.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f32x4* const_f32x4_ptr;   // constant address space: a wave-uniform load becomes s_load

template <bool FAST, int OCT>
__device__ __forceinline__ void uniform_prologue(const UnifiedBufs& ub, const RayRegs& r, int& node, LaneStack& st, int (&spill)[SPILL_DEPTH],
                                                 unsigned int* status)
{
    if ((reinterpret_cast<unsigned long long>(ub.nodes) & 63ull) != 0ull) return;   // (s_load_dwordx16 wants the record 64-byte aligned)
    for (;;) {
        const bool live = node != kSentinel;
        const unsigned long long liveMask = __ballot(live);
        if (liveMask == 0ull) return;
        const int unode = __builtin_amdgcn_readlane(node, (int)__builtin_ctzll(liveMask));   // the first live lane's node: a scalar
        if (__ballot(live && node != unode) != 0ull) return;                                  // the lanes disagree: the general loop from here on
        if ((unsigned)unode >= (unsigned)kSentinel || (unsigned)unode > ub.nodesBytes - 64u) return;   // a leaf (or a malformed offset): likewise
        const const_f32x4_ptr q = (const_f32x4_ptr)(ub.nodes + (unsigned)unode);
        const f32x4 A = q[0], B = q[1], C = q[2], D = q[3];
        if (live)
            inner_advance<FAST, OCT>(make_float4(A.x, A.y, A.z, A.w), make_float4(B.x, B.y, B.z, B.w), make_float4(C.x, C.y, C.z, C.w),
                                     make_float4(D.x, D.y, D.z, D.w), r, node, st, spill, status);
    }
}

// SLICED (persistent kernels): the loop also ends after `slice` iterations (`slice` counts down: the caller posts the dequeue of the wave's
// next chunk, or -- drain phase -- looks at the wave's lanes again: split_settle / split_donate, trace_split.h).
template <bool FAST, bool FLAT, int OCT, bool PROLOGUE, bool SLICED>
__device__ __forceinline__ void traverse_unified(const UnifiedBufs& ub, RayRegs& r, int& node, LaneStack& st,
                                                 int (&spill)[SPILL_DEPTH], bool anyHit, int& hitAddr, float& hitU, float& hitV,
                                                 unsigned int* status, bool poolEmpty, int fetchThreshold, int& slice)
{
    if (PROLOGUE && ub.uniformPrologue) uniform_prologue<FAST, OCT>(ub, r, node, st, spill, status);
    for (;;) {
        const unsigned long long live = __ballot(node != kSentinel);
        if (live == 0ull) break;
        // dynamic fetch (kepler_dynamic_fetch.cu:310): too few live lanes while rays remain in the pool -> refill
        if (!poolEmpty && __popcll(live) < fetchThreshold) break;
        if (SLICED && --slice < 0) break;
        float4 a, b, c, d;
        unified_fetch<FLAT>(ub, node, a, b, c, d);
        if (FLAT) { keep(a); keep(b); keep(c); keep(d); }
        unified_advance<FAST, OCT>(a, b, c, d, r, node, st, spill, anyHit, hitAddr, hitU, hitV, status);
    }
}
template <bool FAST, bool FLAT, int OCT = 8, bool PROLOGUE = false>
__device__ __forceinline__ void traverse_unified(const UnifiedBufs& ub, RayRegs& r, int& node, LaneStack& st,
                                                 int (&spill)[SPILL_DEPTH], bool anyHit, int& hitAddr, float& hitU, float& hitV,
                                                 unsigned int* status, bool poolEmpty, int fetchThreshold)
{
    int never = 0;
    traverse_unified<FAST, FLAT, OCT, PROLOGUE, false>(ub, r, node, st, spill, anyHit, hitAddr, hitU, hitV, status, poolEmpty, fetchThreshold, never);
}

}  // namespace ntr
#include "trace_split.h"   // drain phase of the persistent waves: idle lanes take over parts of the wave's long rays
namespace ntr {

// ---------------------------------------------------------------------------------
// Variant 1: one ray per lane, while-while ("fermi_speculative_while_while" slot).
// ---------------------------------------------------------------------------------
// UNIFIED: the unified-step loop (traverse_unified) -- for trees whose leaves hold several triangles (the device LBVH).
// MINI: the launch may run as the wave-private mini-pool instead (minipool_body below), decided on the device per batch.
template <bool FLATF>
__device__ __forceinline__ void minipool_body(const TraceParams& p, unsigned int K, lds_int* stackBase);

#define NTR_PERRAY_BOUNDS(W) __launch_bounds__((W) * 64, NTR_TRACE_MIN_WAVES_PER_SIMD)
template <int WAVES, bool STATS, bool UNIFIED = false, bool FLATF = true, bool MINI = false>
__device__ __forceinline__ void perray_body(const TraceParams& p)
{
    __shared__ int s_stack[WAVES][LDS_DEPTH][64];  // [wave][entry][lane]
    if constexpr (MINI) {
        static_assert(WAVES == 1 && UNIFIED && !STATS, "the mini-pool shares the one-wave unified-step launch");
        unsigned int K = (unsigned int)p.poolKConst;
        if (p.poolK) {               // wave-uniform (scalar load)
            const unsigned int word = *p.poolK;
            if (p.routeSkip == NTR_ROUTE_SKIP_INCOHERENT && NTR_BATCH_WORD_INCOHERENT(word)) return;   // the persistent body behind this launch traces the batch
            K = NTR_BATCH_WORD_K(word);
        }
        bool pooled = K >= 2u && K <= (unsigned int)NTR_MINIPOOL_MAX_K;
        if (pooled) {
            minipool_body<FLATF>(p, K, (lds_int*)&s_stack[0][0][threadIdx.x]);
            return;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // A "block" is 256 consecutive rays (the unit of the dispatch order and of the cost feedback) whatever the workgroup size: a
    // workgroup of WAVES waves traces one of its 4 / WAVES parts.
    constexpr int PARTS = 4 / WAVES;
    const unsigned int g = blockIdx.x / PARTS, part = blockIdx.x % PARTS;
    const unsigned int block = p.order ? p.order[g] : g;
    const int rayIdx = block * 256 + part * (WAVES * 64) + threadIdx.x;
    const bool valid = rayIdx < p.numRays;
    const Rsrc nodes = make_rsrc(p.nodes, p.nodesBytes), woop = make_rsrc(p.woop, p.woopBytes);

    unsigned long long tl0 = 0;
    if (p.cost) tl0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz; scheduling feedback

    RayRegs r;
    load_ray(p.rays, valid ? rayIdx : 0, r);
    LaneStack st;
    int spill[SPILL_DEPTH];
    st.lds = (lds_int*)&s_stack[wave][0][lane];
    NTR_STACK_RESET(st);

    int hitAddr = -1;
    float hitU = 0.0f, hitV = 0.0f;
    // No triangle can be accepted unless tmin < tmax (t>tmin && t<tmax), so a
    // degenerate ray (Ray::degenerate, Util.hpp:65) is a miss without traversal.
    int node = (valid && r.tmin < r.tmax) ? 0 : kSentinel;
    LaneStats ls = {0u, 0u, 0u};

    const bool fastWave = (p.bvhFlags & NTR_BVH_FASTDIV) && __ballot(node != kSentinel && !ray_is_nice(r, p.bvhFlags)) == 0ull;
    // direction signs shared by every live ray of the wave (a primary wave is an 8 x 8 pixel tile): the octant's own slab test
    int oct = 8;
    if (!STATS && fastWave && p.octant && (p.bvhFlags & NTR_BVH_ORDERED)) {
        const unsigned long long liveMask = __ballot(node != kSentinel);
        const unsigned long long sx = __ballot(node != kSentinel && r.dx < 0.0f), sy = __ballot(node != kSentinel && r.dy < 0.0f),
                                 sz = __ballot(node != kSentinel && r.dz < 0.0f);
        if ((sx == 0ull || sx == liveMask) && (sy == 0ull || sy == liveMask) && (sz == 0ull || sz == liveMask))
            oct = (sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0);
    }
    if (UNIFIED) {
        const UnifiedBufs ub = unified_bufs(p);
#define NTR_UNIFIED_OCT(O) traverse_unified<true, FLATF, O, true>(ub, r, node, st, spill, p.anyHit != 0, hitAddr, hitU, hitV, p.status, true, 0)
        if (oct < 8) {
            switch (oct) {
                case 0: NTR_UNIFIED_OCT(0); break;
                case 1: NTR_UNIFIED_OCT(1); break;
                case 2: NTR_UNIFIED_OCT(2); break;
                case 3: NTR_UNIFIED_OCT(3); break;
                case 4: NTR_UNIFIED_OCT(4); break;
                case 5: NTR_UNIFIED_OCT(5); break;
                case 6: NTR_UNIFIED_OCT(6); break;
                default: NTR_UNIFIED_OCT(7); break;
            }
        }
#undef NTR_UNIFIED_OCT
        else if (fastWave) traverse_unified<true, FLATF, 8, true>(ub, r, node, st, spill, p.anyHit != 0, hitAddr, hitU, hitV, p.status, true, 0);
        else traverse_unified<false, FLATF, 8, true>(ub, r, node, st, spill, p.anyHit != 0, hitAddr, hitU, hitV, p.status, true, 0);
    } else
#define NTR_TRAVERSE_OCT(O) traverse<true, STATS, false, O>(nodes, woop, r, node, st, spill, p.anyHit != 0, hitAddr, hitU, hitV, ls, p.status, true, 0, p.leafSwitchBelow)
    if (!STATS && oct < 8) {
        switch (oct) {
            case 0: NTR_TRAVERSE_OCT(0); break;
            case 1: NTR_TRAVERSE_OCT(1); break;
            case 2: NTR_TRAVERSE_OCT(2); break;
            case 3: NTR_TRAVERSE_OCT(3); break;
            case 4: NTR_TRAVERSE_OCT(4); break;
            case 5: NTR_TRAVERSE_OCT(5); break;
            case 6: NTR_TRAVERSE_OCT(6); break;
            default: NTR_TRAVERSE_OCT(7); break;
        }
    }
#undef NTR_TRAVERSE_OCT
    else if (fastWave) traverse<true, STATS, false>(nodes, woop, r, node, st, spill, p.anyHit != 0, hitAddr, hitU, hitV, ls, p.status, true, 0, p.leafSwitchBelow);
    else traverse<false, STATS, false>(nodes, woop, r, node, st, spill, p.anyHit != 0, hitAddr, hitU, hitV, ls, p.status, true, 0, p.leafSwitchBelow);

    if (p.cost && lane == 0)  // scheduling feedback: a block's cost is the lifetime of its longest wave
        atomicMax(&p.cost[block], (unsigned int)(__builtin_amdgcn_s_memrealtime() - tl0));
    if (!valid) return;
    store_result(p.results, p.triIndex, rayIdx, hitAddr, r.tmax, hitU, hitV);
    if (STATS) {  // diagnostics variant only: plain per-lane atomics
        atomicAdd(&p.stats[0], (unsigned long long)ls.inner);
        atomicAdd(&p.stats[1], (unsigned long long)ls.tris);
        atomicAdd(&p.stats[2], (unsigned long long)ls.leaves);
        atomicAdd(&p.stats[3], (unsigned long long)(hitAddr >= 0));
    }
}

template <int WAVES, bool STATS, bool UNIFIED = false, bool FLATF = true>
__global__ NTR_PERRAY_BOUNDS(WAVES) void trace_bvh_perray(TraceParams p)
{
    perray_body<WAVES, STATS, UNIFIED, FLATF, false>(p);
}
// The one-wave unified-step launch that may run as mini-pools (K decided on the device).
#define NTR_MINI_BOUNDS __launch_bounds__(64)
__global__ NTR_MINI_BOUNDS void trace_bvh_perray_mini(TraceParams p)
{
    perray_body<1, false, true, true, true>(p);
}
// ... with the two-descriptor fetch, for a BVH whose node and triangle buffers do not lie inside one 4 GiB window (the flat fetch's
// condition): such a batch keeps its ray pools -- without them an incoherent batch is 60 % slower (hairball box rays 3.6 -> 5.8 ms)
__global__ NTR_MINI_BOUNDS void trace_bvh_perray_mini_desc(TraceParams p)
{
    perray_body<1, false, true, false, true>(p);
}

// ---------------------------------------------------------------------------------
// Variant 2: persistent waves.  Each wave owns a chunk [next,end) of the ray index
// space; empty lanes are refilled from the chunk by ballot + mbcnt prefix
// (kepler_dynamic_fetch.cu:97-111 on wave64: 64-bit ballot, v_mbcnt_lo/hi).
// Pool: the index space is cut into numHeads contiguous ranges, one head (counter) each.  A returning atomic on ONE
// address is served at about 88 per us (MI355X_MICROARCH price list, "dequeue"); with 6 144 waves asking at once --
// at launch, and again whenever equally long rays (AO) end together -- eight heads made a dequeue wait 8-12 us
// (profiles/r02b_persistent_vs_perray_timelines.jsonl: 32 % of a wave's life on AO; r02k: 13 % with 64-256 heads, AO batch
// 208 -> 134 us; 512 heads and more lose again to end-of-pool probing).  So: (1) a wave's FIRST chunk is
// assigned statically, no atomic; (2) 128 heads, a block works on head blockIdx % numHeads (blocks are dealt round-robin
// to the XCDs, so head h stays on XCD h % 8 and an XCD's heads cover one contiguous screen region); (3) a wave whose
// head ran dry reads all heads with one 64-lane load and moves to the next one that still has rays, instead of
// paying an atomic round trip per dry head.
// ---------------------------------------------------------------------------------
// UNIFIED: the unified-step loop (traverse_unified) instead of the while-while loop -- what kepler_dynamic_fetch launches.
// Round 6: what a persistent wave does between two chunks used to be 40 % of its life on coherent batches (an AO batch of the headline
// frame: 7 us per refill -- the returning atomic on the pool head, then the ray load, one dependent round trip after the other -- of
// a 20 us chunk; profiles/r05_persist_ao_timeline.jsonl), and a refilled wave ran the general loop where a fresh wave of the per-ray
// kernel runs the octant-specialised one behind the uniform prologue.  Now:
//   * WHOLE-WAVE refills are fresh waves: a wave that was empty and took its rays from one chunk walks the top of the tree through the
//     scalar cache exactly like a wave of the per-ray kernel (uniform_prologue; the octant-specialised slab test is left to the
//     per-ray kernel: eight more loop bodies cost these kernels 6-7 VGPRs, i.e. a wave per SIMD, for +3 %);
//   * the NEXT chunk is dequeued while the current one is traced: `prefetchAfter` iterations into a chunk lane 0 posts the atomic for the
//     wave's next chunk, and the value is there when the wave comes back (not at the start of the chunk: that would commit every wave
//     to two chunks at launch -- 16 384 chunks of an AO batch over 8 192 waves -- and give up the dynamic balancing a pool is for);
//   * kepler_dynamic_fetch refills single lanes (ballot / mbcnt, fetchThreshold) only on batches the device's coherence estimate (poolK)
//     found incoherent -- dynamic fetch pays exactly where the rays of one chunk differ in length, and costs ~10 % where they do not
//     (refilled lanes de-cohere a wave's node fetches) -- and refills whole waves otherwise.  (A per-wave switch decided by how busy a
//     chunk kept its lanes was built first: changing the policy inside the wave's main loop costs 13 VGPRs, i.e. a wave per SIMD.)
// None of this touches a ray's own visiting order: records cannot change.
template <int WAVES, bool UNIFIED = false, bool FLATF = true>
__global__ __launch_bounds__(WAVES * 64, NTR_TRACE_PERSISTENT_MIN_WAVES_PER_SIMD) void trace_bvh_persistent(TraceParams p)
{
    __shared__ int s_stack[WAVES][LDS_DEPTH][64];  // [wave][entry][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Rsrc nodes = make_rsrc(p.nodes, p.nodesBytes), woop = make_rsrc(p.woop, p.woopBytes);
    const bool anyHit = p.anyHit != 0;
    const bool bvhFast = (p.bvhFlags & NTR_BVH_FASTDIV) != 0;
    // How many rays should be in flight?  Beyond the L2 the chip serves ~56 G requests/s from ~64 k requests in flight on; more in flight
    // only adds queueing delay, which every ray -- the batch's longest included -- pays per step (EXPERIMENTS.md, gather roof).  A batch the
    // device found incoherent (pool word > 1: scattered origins, every step a cache miss) is therefore traced by HALF the grid:
    // courtyard-10M box rays 5.12 -> 4.60 ms, hairball 4.16 -> 3.43 ms; coherent batches keep the full grid (atrium primary 0.57 against
    // 0.67 ms with half).  Workgroups beyond the effective grid leave at once; everything below counts with the effective grid.
    int numBlocksEff = p.numBlocks;   // wave-uniform
    const unsigned int batchWord = p.poolK ? *p.poolK : 1u;
    const bool incoherentBatch = NTR_BATCH_WORD_INCOHERENT(batchWord);   // scattered origins, or long rays that point apart: single-lane refills
    if (p.routeSkip == NTR_ROUTE_SKIP_COHERENT && !incoherentBatch) return;   // the per-ray body beside this launch traces the batch
    // (round 6: a diffuse batch -- rays that start together and wander apart -- gains from fewer rays in flight too: one hairball batch 2.03 ms with
    // eight workgroups per CU, 1.57 with four, 1.63-1.74 with three; scattered origins -- box rays -- are best with three: 2.86 against 3.4 ms with four)
    const int gridIncoherent = NTR_BATCH_WORD_K(batchWord) > 1u ? p.numBlocksIncoherent : p.numBlocksDivergent;
    if (incoherentBatch && gridIncoherent > 0 && gridIncoherent < numBlocksEff) numBlocksEff = gridIncoherent;
    if ((int)blockIdx.x >= numBlocksEff) return;

    LaneStack st;
    int spill[SPILL_DEPTH];
    st.lds = (lds_int*)&s_stack[wave][0][lane];
    st.sp = 0;
    st.tos = kSentinel;

    RayRegs r = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int node = kSentinel, rayIdx = -1, hitAddr = -1;
    float hitU = 0.0f, hitV = 0.0f;
    bool nice = true;                 // this lane's current ray qualifies for the FAST path
    int chunkNext = 0, chunkEnd = 0;  // wave-uniform
    bool poolEmpty = false;           // wave-uniform
    const int numHeads = p.numHeads;
    int shard = (int)(blockIdx.x % (unsigned)numHeads);  // wave-uniform
    bool firstChunk = true;
    // Pool positions: head h owns positions [h * shardRays, (h + 1) * shardRays).  Buffer order (p.order == null): the k-th chunk of head h
    // is chunk k * numHeads + h of the batch -- the heads interleave, so every head holds an even sample of the batch and all of them
    // run dry together.  (Until round 6 a head owned a CONTIGUOUS range of the batch, "so that an XCD's L2 sees one screen region": the
    // heads of cheap screen regions then ran dry early, their waves all moved to the same next head -- a returning atomic on one address
    // is served at ~88 per us -- and a coherent launch spent most of its time in that queue whatever its occupancy: AO batches 145 us
    // with 6, 7 or 8 workgroups per CU alike, against 52 us for the per-ray kernel, whose workgroups interleave over the XCDs the same
    // way.)  Predicted-cost order (p.order: the 256-ray blocks heaviest class first, sched_kernels.hip): the positions of head h are the
    // blocks order[h], order[h + numHeads], order[h + 2 numHeads], ... -- every head hands its blocks out from heavy to light, so the
    // long-lived rays of the batch start first instead of forming the tail of the launch.  Either way the pool spans numHeads * shardRays
    // positions and a position may lie beyond the batch: it is skipped.
    const unsigned int* const order = p.order;
    const int poolEnd = numHeads * p.shardRays;
    auto range_beg = [&](int h) { return h * p.shardRays; };
    int chunkDelta = 0;               // wave-uniform: ray index - pool position inside the current chunk (a chunk never spans two 256-ray blocks)
    // the chunk at position `base` of head h: where its rays are in the batch.  One scalar load per chunk in the ordered pool (until round 6
    // every lane looked its block up in order[] itself: a dependent vector load on the refill's critical path)
    auto chunk_delta = [&](int base, int h) {
        const int off = base - h * p.shardRays;
        if (!order) return ((off / p.chunk) * numHeads + h) * p.chunk - base;
        const int q = (off >> 8) * numHeads + h;                  // the block's place in the predicted / learned order
        if (q >= p.orderBlocks) return 0x3FFFFFFF - base;          // beyond the batch: every position of the chunk maps past numRays
        return (int)order[q] * 256 + (off & 255) - base;
    };
    // chunks of head h handed out statically: one per wave of every block with blockIdx % numHeads == h
    auto static_rays = [&](int h) { return ((numBlocksEff - h + numHeads - 1) / numHeads) * WAVES * p.chunk; };
    LaneStats ls = {0u, 0u, 0u};
    // drain phase (unified-step loop): once the pool is dry, idle lanes take over parts of the wave's rays (trace_split.h)
    SplitState split;
    split_reset(split);
    bool splitOn = false;             // wave-uniform
    const int splitSlice = UNIFIED ? p.splitSlice : 0;

    // refill policy (wave-uniform): whole-wave (a wave takes rays only when it holds none) or dynamic fetch (fewer than fetchThreshold
    // lanes live -> the idle lanes take rays; kepler_dynamic_fetch.cu:310)
    const bool dynamicFetch = p.fetchThreshold > 0 && (!UNIFIED || p.wholeWave == 0 || incoherentBatch);
    // dequeue-ahead: the atomic of the wave's next chunk, posted `prefetchAfter` iterations into the current one
    int prefetched = 0;               // lane 0: what the atomic returned
    bool havePrefetch = false;        // wave-uniform
    int prefetchHead = 0;             // wave-uniform: the head it was posted on
    int prefetchIn = -1;              // wave-uniform: iterations until the prefetch is posted (< 0: none pending)
    // scheduling feedback (a hint's refresh launch, whole-wave mode): a 256-ray block's cost is the life of the longest chunk taken from it
    int costBlock = -1;               // wave-uniform: the block of the chunk in flight (-1: none / not recorded)
    unsigned long long costT0 = 0;

    // The wave's next chunk: its statically assigned first one, the one it dequeued ahead, or an atomic on its head -- and when that head
    // is dry, on the next head that still has rays.  Sets chunkNext / chunkEnd / chunkHead / chunkDelta; false = the pool is dry.
    // The ray index space is dealt to numHeads pool heads (a single head saturates near 88 dequeues/us, MI355X_MICROARCH price list "dequeue").
    auto grab = [&]() -> bool {
        bool got = false;
        auto take = [&](int base, int h) {
            const int rangeBeg = range_beg(h);
            const int rangeEnd = min(rangeBeg + p.shardRays, poolEnd);
            if (base >= rangeEnd) return false;
            chunkNext = base;
            chunkEnd = min(base + p.chunk, rangeEnd);
            chunkDelta = chunk_delta(base, h);
            return true;
        };
        if (firstChunk) {  // static: the (blockIdx / numHeads * WAVES + wave)-th chunk of the block's head
            firstChunk = false;
            got = take(range_beg(shard) + ((int)(blockIdx.x / (unsigned)numHeads) * WAVES + __builtin_amdgcn_readfirstlane(wave)) * p.chunk, shard);
        } else if (havePrefetch) {   // the dequeue posted while the previous chunk was traced
            havePrefetch = false;    // (a head that ran dry meanwhile: the search below starts on it and moves on)
            got = take(__builtin_amdgcn_readfirstlane(prefetched) + static_rays(prefetchHead) + range_beg(prefetchHead), prefetchHead);
        }
        while (!got) {
            int base = 0;
            if (lane == 0) base = atomicAdd(p.counter + shard * 16, p.chunk);
            if (take(__builtin_amdgcn_readfirstlane(base) + static_rays(shard) + range_beg(shard), shard)) { got = true; break; }
            // dry: lane l looks at head (shard + 1 + l) % numHeads; counters only grow, so a head seen dry stays dry
            unsigned long long live = 0ull;
            int ofs = 1;
            for (; ofs < numHeads && live == 0ull; ofs += 64) {
                bool has = false;
                if (ofs + lane < numHeads) {
                    const int h = (shard + ofs + lane) % numHeads;
                    const int hb = range_beg(h);
                    const int he = min(hb + p.shardRays, poolEnd);
                    const int taken = __hip_atomic_load(p.counter + h * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    has = hb + static_rays(h) + taken < he;
                }
                live = __ballot(has);
            }
            if (live == 0ull) break;
            shard = (shard + ofs - 64 + (int)__builtin_ctzll(live)) % numHeads;
        }
        if (got) prefetchIn = p.prefetchAfter;   // (negative: no dequeue-ahead)
        return got;
    };
    // scheduling feedback: the 256-ray block of the chunk just grabbed
    auto cost_open = [&]() {
        costBlock = (chunkEnd - p.chunk + chunkDelta) >> 8;
        if ((unsigned)costBlock >= (unsigned)((p.numRays + 255) >> 8)) costBlock = -1;
        costT0 = __builtin_amdgcn_s_memrealtime();
    };

    // Invariant at the top of the loop: a lane either holds a live ray
    // (rayIdx >= 0, node != sentinel) or is empty (rayIdx < 0, node == sentinel).
    for (;;) {
        // ---- refill empty lanes from the wave's chunk ----------------------------
        unsigned long long empty = __ballot(rayIdx < 0);
        const bool wholeWave = empty == ~0ull;
        const bool refill = !poolEmpty && (wholeWave || (dynamicFetch && 64 - __popcll(empty) < p.fetchThreshold));
        while (refill && empty != 0ull && !poolEmpty) {
            if (chunkNext >= chunkEnd && !grab()) { poolEmpty = true; break; }
            // rank of this lane among the empty lanes (wave64 prefix popcount)
            const int prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(empty >> 32),
                               __builtin_amdgcn_mbcnt_lo((unsigned)empty, 0));
            const int avail = chunkEnd - chunkNext;
            const int pos = chunkNext + prefix + chunkDelta;   // pool position -> ray index (beyond the batch: skipped)
            if (rayIdx < 0 && prefix < avail && pos < p.numRays) {
                rayIdx = pos;
                load_ray(p.rays, rayIdx, r);
                hitAddr = -1;
                hitU = hitV = 0.0f;
                NTR_STACK_RESET(st);
                // tmin < tmax is necessary for any accept (t>tmin && t<tmax):
                // degenerate rays (Util.hpp:65) are misses without traversal.
                node = (r.tmin < r.tmax) ? 0 : kSentinel;
                nice = ray_is_nice(r, p.bvhFlags);
            }
            chunkNext += min(__popcll(empty), avail);
            empty = __ballot(rayIdx < 0);
            if (wholeWave && !dynamicFetch) break;   // whole-wave mode takes rays ONCE per refill (a short last chunk stays short)
        }
        const bool fresh = refill && wholeWave;   // every ray the wave holds starts at the root now
        if (p.cost && refill && wholeWave && !dynamicFetch && !poolEmpty) cost_open();

        // ---- drain phase: lanes without a ray take over stack entries of the wave's live rays (unified-step loop) ------------------
        if (UNIFIED && poolEmpty && splitSlice > 0) {
            if (!splitOn) { splitOn = true; split_reset(split); }
            split_donate(split, r, node, st, rayIdx, hitAddr, hitU, hitV, nice);
        }
        const bool fastWave = bvhFast && __ballot(node != kSentinel && !nice) == 0ull;
        // dequeue-ahead: the traversal comes back after `prefetchIn` iterations, the atomic is posted, the traversal goes on
        int slice = 0x7FFFFFFF;
        if (splitOn) slice = splitSlice;   // drain phase with splitting: the lanes are looked at again every `slice` steps
        else if (prefetchIn >= 0 && !poolEmpty) slice = prefetchIn;
        const int fetchBelow = dynamicFetch ? p.fetchThreshold : 0;
        if (UNIFIED) {
            const UnifiedBufs ub = unified_bufs(p);
            if (fresh && p.uniformPrologue) {   // the top of the tree through the scalar cache while the wave's rays agree (uniform_prologue)
                if (fastWave) uniform_prologue<true, 8>(ub, r, node, st, spill, p.status);
                else uniform_prologue<false, 8>(ub, r, node, st, spill, p.status);
            }
            if (fastWave) traverse_unified<true, FLATF, 8, false, true>(ub, r, node, st, spill, anyHit, hitAddr, hitU, hitV, p.status, poolEmpty, fetchBelow, slice);
            else traverse_unified<false, FLATF, 8, false, true>(ub, r, node, st, spill, anyHit, hitAddr, hitU, hitV, p.status, poolEmpty, fetchBelow, slice);
            if (splitOn) split_settle(split, r, node, st, hitAddr, hitU, hitV, anyHit);
        } else {
            if (fresh && p.uniformPrologue) {   // the top of the tree through the scalar cache while the wave's rays agree (uniform_prologue)
                const UnifiedBufs ub = unified_bufs(p);
                if (fastWave) uniform_prologue<true, 8>(ub, r, node, st, spill, p.status);
                else uniform_prologue<false, 8>(ub, r, node, st, spill, p.status);
            }
            if (fastWave) traverse<true, false, true, 8, true>(nodes, woop, r, node, st, spill, anyHit, hitAddr, hitU, hitV, ls, p.status, poolEmpty, fetchBelow, p.leafSwitchBelow, slice);
            else traverse<false, false, true, 8, true>(nodes, woop, r, node, st, spill, anyHit, hitAddr, hitU, hitV, ls, p.status, poolEmpty, fetchBelow, p.leafSwitchBelow, slice);
        }
        // ---- dequeue-ahead: post the atomic of the wave's next chunk now (its latency hides behind the rest of this chunk) -------------
        if (!splitOn && prefetchIn >= 0 && !poolEmpty) {
            prefetchIn = -1;
            if (!havePrefetch && __ballot(node != kSentinel) != 0ull) {   // (a wave that is done already dequeues in the refill above)
                if (lane == 0) prefetched = atomicAdd(p.counter + shard * 16, p.chunk);
                havePrefetch = true;
                prefetchHead = shard;
            }
        }

        // ---- retire finished rays (an owner whose helpers are still out waits for their reports) ------
        if (rayIdx >= 0 && node == kSentinel && (!splitOn || split.base == 0)) {
            store_result(p.results, p.triIndex, rayIdx, hitAddr, r.tmax, hitU, hitV);
            rayIdx = -1;
        }
        if (costBlock >= 0 && __ballot(rayIdx >= 0) == 0ull) {   // the chunk is done
            if (lane == 0) atomicMax(&p.cost[costBlock], (unsigned int)(__builtin_amdgcn_s_memrealtime() - costT0));
            costBlock = -1;
        }
        if (poolEmpty && __ballot(rayIdx >= 0) == 0ull) break;
    }
}

// ---------------------------------------------------------------------------------
// Variant 3: per-ray kernel with a wave-private mini-pool (round 3).  A hardware-scheduled 64-thread workgroup owns K x 64 consecutive
// rays instead of 64: its lanes start on the first 64, and a lane that finishes takes the wave's next unstarted ray (ballot + mbcnt
// prefix over the wave's OWN range: no atomic, no shared head).  On divergent batches the per-ray kernel's waves live as long as their
// longest ray while most lanes idle (lane utilisation 0.22-0.48 on the LBVH scenes, scripts/studies/divergence_study.py); list scheduling K x 64
// rays on 64 lanes lifts that to 0.33-0.63 (K = 2) / 0.49-0.77 (K = 4) by the per-ray step counts, at the price of a longer critical
// path per wave -- which is why the pool stays small and private: the global pool of the persistent kernels keeps every lane busy
// until it runs dry, and then 6 144 waves each hold a few long rays (a tail of 60-70 % of their launch, profiles/r03_divergence_timelines.jsonl).
// Unified-step loop, flat fetch; 256-ray blocks keep their role as the unit of the dispatch order and of the cost feedback.
// ---------------------------------------------------------------------------------
template <bool FLATF>
__device__ __forceinline__ void minipool_body(const TraceParams& p, unsigned int K, lds_int* stackBase)
{
    // The launch has one wave per 64-ray chunk (the per-ray kernel's grid); K consecutive chunks of the dispatch order form a pool, and one
    // workgroup of every K owns it, the others exit at once.  WHICH one must not follow a regular pattern: workgroups are dealt round-robin
    // to XCDs, shader engines and CUs, and "every 4th workgroup" put all live waves on a quarter of the chip (measured 2.7x slower, with the
    // XCD bits excluded just the same).  So the live member of a group is picked by a golden-ratio hash of the group's position.
    const int lane = threadIdx.x;
    const unsigned int numChunks = gridDim.x;                                    // 4 per 256-ray block of the order, the last block's empty ones included
    const unsigned int q = blockIdx.x / K;                                       // wave-uniform (one software division per wave)
    const unsigned int members = min(K, numChunks - q * K);                      // (the last group may be short)
    if (blockIdx.x - q * K != ((((q * 0x9E3779B1u) >> 16) * members) >> 16)) return;
    unsigned int chunk = q * K;
    const unsigned int chunkEnd = min(chunk + K, numChunks);
    if (chunk >= chunkEnd) return;
    // chunk c of the order = quarter (c & 3) of block order[c >> 2]
    unsigned int block = p.order ? p.order[chunk >> 2] : (chunk >> 2);
    int poolNext = (int)(block * 256u + (chunk & 3u) * 64u);                     // wave-uniform: the unstarted rays of the current chunk
    int poolEnd = min(poolNext + 64, p.numRays);
    const bool anyHit = p.anyHit != 0;
    const bool bvhFast = (p.bvhFlags & NTR_BVH_FASTDIV) != 0;
    const UnifiedBufs ub = unified_bufs(p);

    unsigned long long tl0 = 0;
    if (p.cost) tl0 = __builtin_amdgcn_s_memrealtime();

    LaneStack st;
    int spill[SPILL_DEPTH];
    st.lds = stackBase;
    NTR_STACK_RESET(st);
    RayRegs r = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int node = kSentinel, rayIdx = -1, hitAddr = -1;
    float hitU = 0.0f, hitV = 0.0f;
    bool nice = true;

    for (;;) {
        // ---- start the wave's next rays on its empty lanes (from the current chunk; what it cannot fill is filled next time round) --
        while (poolNext >= poolEnd && chunk + 1u < chunkEnd) {
            chunk++;
            block = p.order ? p.order[chunk >> 2] : (chunk >> 2);
            poolNext = (int)(block * 256u + (chunk & 3u) * 64u);
            poolEnd = min(poolNext + 64, p.numRays);
        }
        const unsigned long long empty = __ballot(rayIdx < 0);
        if (empty != 0ull && poolNext < poolEnd) {
            const int prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(empty >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)empty, 0));
            const int avail = poolEnd - poolNext;
            if (rayIdx < 0 && prefix < avail) {
                rayIdx = poolNext + prefix;
                load_ray(p.rays, rayIdx, r);
                hitAddr = -1;
                hitU = hitV = 0.0f;
                NTR_STACK_RESET(st);
                node = (r.tmin < r.tmax) ? 0 : kSentinel;   // degenerate rays (Util.hpp:65) are misses without traversal
                nice = ray_is_nice(r, p.bvhFlags);
            }
            poolNext += min(__popcll(empty), avail);
        }
        const bool poolEmpty = poolNext >= poolEnd && chunk + 1u >= chunkEnd;
        // ---- unified-step traversal until every lane is done, or (rays left in the pool) until enough lanes are free to be worth a refill --
        const bool fastWave = bvhFast && __ballot(node != kSentinel && !nice) == 0ull;
        if (fastWave) traverse_unified<true, FLATF, 8, false>(ub, r, node, st, spill, anyHit, hitAddr, hitU, hitV, p.status, poolEmpty, p.fetchThreshold);
        else traverse_unified<false, FLATF, 8, false>(ub, r, node, st, spill, anyHit, hitAddr, hitU, hitV, p.status, poolEmpty, p.fetchThreshold);
        // ---- retire finished rays -------------------------------------------------------------------------------------------------
        if (rayIdx >= 0 && node == kSentinel) {
            store_result(p.results, p.triIndex, rayIdx, hitAddr, r.tmax, hitU, hitV);
            rayIdx = -1;
        }
        if (poolEmpty && __ballot(rayIdx >= 0) == 0ull) break;
    }
    if (p.cost && lane == 0) {  // scheduling feedback: a block's cost is the lifetime of the longest wave that traced a part of it
        const unsigned int life = (unsigned int)(__builtin_amdgcn_s_memrealtime() - tl0);
        for (unsigned int c = q * K; c < chunkEnd; c += 4u - (c & 3u)) atomicMax(&p.cost[p.order ? p.order[c >> 2] : (c >> 2)], life);
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
    case NTR_VARIANT_PERRAY_W2:   // smaller workgroups for short any-hit rays (numBlocks counts 128-ray blocks)
        hipLaunchKernelGGL((ntr::trace_bvh_perray<2, false>), dim3(numBlocks), dim3(128), 0, stream, *p);
        break;
    case NTR_VARIANT_PERRAY_W1:
        hipLaunchKernelGGL((ntr::trace_bvh_perray<1, false>), dim3(numBlocks), dim3(64), 0, stream, *p);
        break;
    case NTR_VARIANT_PERRAY_UNIFIED_W1:   // flatFetch 0: the two-group descriptor fetch (A/B; extents below 64 bytes)
        if (p->flatFetch) hipLaunchKernelGGL((ntr::trace_bvh_perray<1, false, true, true>), dim3(numBlocks), dim3(64), 0, stream, *p);
        else hipLaunchKernelGGL((ntr::trace_bvh_perray<1, false, true, false>), dim3(numBlocks), dim3(64), 0, stream, *p);
        break;
    case NTR_VARIANT_PERRAY_UNIFIED_MINI:   // numBlocks counts waves of 64 rays
        if (p->flatFetch) hipLaunchKernelGGL(ntr::trace_bvh_perray_mini, dim3(numBlocks), dim3(64), 0, stream, *p);
        else hipLaunchKernelGGL(ntr::trace_bvh_perray_mini_desc, dim3(numBlocks), dim3(64), 0, stream, *p);
        break;
    case NTR_VARIANT_PERRAY_STATS:
        hipLaunchKernelGGL((ntr::trace_bvh_perray<WAVES, true>), dim3(numBlocks), dim3(WAVES * 64), 0, stream, *p);
        break;
    case NTR_VARIANT_PERSISTENT_UNIFIED:
        if (p->flatFetch) hipLaunchKernelGGL((ntr::trace_bvh_persistent<WAVES, true, true>), dim3(numBlocks), dim3(WAVES * 64), 0, stream, *p);
        else hipLaunchKernelGGL((ntr::trace_bvh_persistent<WAVES, true, false>), dim3(numBlocks), dim3(WAVES * 64), 0, stream, *p);
        break;
    case NTR_VARIANT_PERSISTENT:
        hipLaunchKernelGGL((ntr::trace_bvh_persistent<WAVES, false>), dim3(numBlocks), dim3(WAVES * 64), 0, stream, *p);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
