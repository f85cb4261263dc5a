// CudaBVHTrace.cpp -- CudaAS::trace(RayBuffer&, Buffer& visibility): the reference's HOST tracer
// (src/rt/cuda/CudaBVH.cpp:213-302 -> trace<BVHLayout_Compact> :698-784 -> intersectTriangles :1084-1126
// -> updateHit :1183-1225; Intersect::RayBox / RayTriangleWoop, src/rt/Util.cpp:34-46, 99-127).
//
// It is part of the reference's CudaAS interface (CPURenderer and BASELINE configuration 1, "Cornell box:
// CPU SAH-BVH build + CPU primary-ray trace via the src/rt host path", use it), so the mirror has it too.
// It is NOT a fallback of the device tracer: CudaBVHTracer::traceBatch / ntr_trace_bvh never come here and
// fail without a HIP device.  Single-threaded like the reference; binary32 arithmetic in the reference's
// source order (this file is compiled with -ffp-contract=off, no fast-math), which is also what the HIP
// kernels reproduce -- hit records (id, t) of the two are identical.
#include <cfloat>
#include <cstring>

#include "CudaBVH.hpp"

namespace FW {

namespace {

// generic FW::min / FW::max are selects (src/framework/base/Defs.hpp:212-213): NaN and signed-zero behaviour
// differs from fminf / fmaxf and decides which child a grazing ray enters
inline F32 pick_lo(F32 a, F32 b) { return (a < b) ? a : b; }
inline F32 pick_hi(F32 a, F32 b) { return (a > b) ? a : b; }

struct Span { F32 enter, leave; };

struct CompactTree {
    const U8* nodes;     // 64 B per inner node (CudaBVH.hpp:42-46)
    const U8* woop;      // 16 B per float4
    const S32* index;    // parallel to woop float4s
};

// Intersect::RayBox: (plane - origin) / direction per component, true divisions, x-y-z folds
inline Span slab(const F32 x[2], const F32 y[2], const F32 z[2], const Ray& r)
{
    const F32 ax = (x[0] - r.origin.x) / r.direction.x, bx = (x[1] - r.origin.x) / r.direction.x;
    const F32 ay = (y[0] - r.origin.y) / r.direction.y, by = (y[1] - r.origin.y) / r.direction.y;
    const F32 az = (z[0] - r.origin.z) / r.direction.z, bz = (z[1] - r.origin.z) / r.direction.z;
    Span s;
    s.enter = pick_hi(pick_hi(pick_lo(ax, bx), pick_lo(ay, by)), pick_lo(az, bz));
    s.leave = pick_lo(pick_lo(pick_hi(ax, bx), pick_hi(ay, by)), pick_hi(az, bz));
    return s;
}

// dot(Vec4f, Vec4f) accumulates from 0 over the four components (src/framework/base/Math.hpp:185)
inline F32 dot_from_zero(const F32* p, F32 x, F32 y, F32 z, F32 w)
{
    F32 acc = 0.0f;
    acc += p[0] * x;
    acc += p[1] * y;
    acc += p[2] * z;
    acc += p[3] * w;
    return acc;
}

// Intersect::RayTriangleWoop: t of the hit, or FW_F32_MAX
inline F32 woop_hit(const F32* zp, const F32* up, const F32* vp, const Ray& r)
{
    const F32 oz = zp[3] - r.origin.x * zp[0] - r.origin.y * zp[1] - r.origin.z * zp[2];
    const F32 inv = 1.0f / dot_from_zero(zp, r.direction.x, r.direction.y, r.direction.z, 0.0f);
    const F32 t = oz * inv;
    if (!(t > r.tmin && t < r.tmax)) return FLT_MAX;
    const F32 u = dot_from_zero(up, r.origin.x, r.origin.y, r.origin.z, 1.0f) + t * dot_from_zero(up, r.direction.x, r.direction.y, r.direction.z, 0.0f);
    if (!(u >= 0.0f)) return FLT_MAX;
    const F32 v = dot_from_zero(vp, r.origin.x, r.origin.y, r.origin.z, 1.0f) + t * dot_from_zero(vp, r.direction.x, r.direction.y, r.direction.z, 0.0f);
    if (!(v >= 0.0f && (u + v) <= 1.0f)) return FLT_MAX;
    return t;
}

template <bool ANY_HIT>
void walk(const CompactTree& tree, Ray r, RayResult& out, RayStats* stats)
{
    enum { Depth = 100 };  // CudaBVH.cpp:701
    S32 pending[Depth];
    int top = 0;
    S32 at = 0;  // byte offset of the root
    for (;;) {
        if (at >= 0) {  // inner node: test both child boxes, descend into the nearer one
            const F32* nd = reinterpret_cast<const F32*>(tree.nodes + at);
            const S32* link = reinterpret_cast<const S32*>(tree.nodes + at + 48);
            const F32 x0[2] = {nd[0], nd[1]}, y0[2] = {nd[2], nd[3]}, z0[2] = {nd[8], nd[9]};
            const F32 x1[2] = {nd[4], nd[5]}, y1[2] = {nd[6], nd[7]}, z1[2] = {nd[10], nd[11]};
            const Span s0 = slab(x0, y0, z0, r), s1 = slab(x1, y1, z1, r);
            const bool in0 = s0.enter <= s0.leave && s0.leave >= r.tmin && s0.enter <= r.tmax;   // :742-743
            const bool in1 = s1.enter <= s1.leave && s1.leave >= r.tmin && s1.enter <= r.tmax;
            if (stats) stats->numNodeTests += 2;
            if (in0 && in1) {
                const bool secondFirst = s0.enter > s1.enter;  // ties keep child 0 first (:761)
                if (top == Depth) fail("CudaBVH::trace: traversal stack overflow");
                pending[top++] = secondFirst ? link[0] : link[1];
                at = secondFirst ? link[1] : link[0];
                continue;
            }
            if (in0) { at = link[0]; continue; }
            if (in1) { at = link[1]; continue; }
        } else {  // leaf: triangles in stored order until the terminator (:1084-1126)
            for (S32 a = ~at;; a += 3) {
                const F32* zp = reinterpret_cast<const F32*>(tree.woop + (size_t)a * 16);
                if (floatToBits(zp[0]) == 0x80000000u) break;
                if (stats) stats->numTriangleTests++;
                const F32 t = woop_hit(zp, zp + 4, zp + 8, r);
                // updateHit re-tests what came back, FW_F32_MAX included: with tmax = +inf a missed test is
                // recorded at t = FLT_MAX (:1200)
                if (t > r.tmin && t < r.tmax) {
                    r.tmax = t;
                    out.t = t;
                    out.id = tree.index[a];
                    if (ANY_HIT) return;
                }
            }
        }
        if (top == 0) return;
        at = pending[--top];
    }
}

}  // namespace

void CudaBVH::trace(RayBuffer& rays, Buffer& visibility, RayStats* stats)
{
    if (m_layout != BVHLayout_Compact) fail("CudaBVH::trace: only BVHLayout_Compact is supported");
    const S32 n = rays.getSize();
    if (n == 0) return;
    if (m_nodes.getSize() < 64) fail("CudaBVH::trace: No BVH!");
    CompactTree tree;
    tree.nodes = m_nodes.getPtr();
    tree.woop = m_triWoop.getPtr();
    tree.index = reinterpret_cast<const S32*>(m_triIndex.getPtr());
    const bool anyHit = !rays.getNeedClosestHit();
    S32* visib = visibility.getSize() > 0 ? reinterpret_cast<S32*>(visibility.getMutablePtr()) : NULL;
    const S64 visCount = visibility.getSize() / (S64)sizeof(S32);
    for (S32 slot = 0; slot < n; slot++) {
        const Ray ray = rays.getRayForSlot(slot);
        RayResult& res = rays.getMutableResultForSlot(slot);
        res.clear();          // :273-274: a miss is (-1, ray.tmax)
        res.t = ray.tmax;
        if (stats) stats->numRays++;
        if (anyHit) walk<true>(tree, ray, res, stats);
        else walk<false>(tree, ray, res, stats);
        if (visib && res.hit() && res.id < visCount) visib[res.id] = 1;   // :296-297
    }
}

}  // namespace FW
