/*
 * ntr_oracle_trace.c -- CPU ORACLE (test infrastructure only; see ntr_oracle.h).
 *
 * Restates the reference CPU tracer for BVHLayout_Compact:
 *   CudaBVH::trace                         src/rt/cuda/CudaBVH.cpp:213-302
 *   CudaBVH::trace<BVHLayout_Compact>      src/rt/cuda/CudaBVH.cpp:698-784
 *   intersectTriangles<BVHLayout_Compact>  src/rt/cuda/CudaBVH.cpp:1084-1126
 *   CudaBVH::updateHit                     src/rt/cuda/CudaBVH.cpp:1183-1225
 *   getNodeTemplate<BVHLayout_Compact>     src/rt/cuda/CudaBVH.cpp:1250-1265
 *   Intersect::RayBox                      src/rt/Util.cpp:34-46
 *   Intersect::RayTriangleWoop             src/rt/Util.cpp:99-127
 *   FW::min/max (select form)              src/framework/base/Defs.hpp:212-213
 *   VectorBase::min()/max()/dot()          src/framework/base/Math.hpp:146-147,185
 *
 * PARITY UNPINNED by the reference (no goldens, reference unbuildable here);
 * cross-checked by tests/np_tracer.py and brute force.
 *
 * Must be compiled with -ffp-contract=off -fno-fast-math (see Makefile).
 */
#include "ntr_oracle.h"

#include <float.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_STACK_SIZE 100 /* CudaBVH.cpp:701 */

/* Defs.hpp:212-213: generic FW::min / FW::max are selects, not fminf/fmaxf. */
static inline float fw_min(float a, float b) { return (a < b) ? a : b; }
static inline float fw_max(float a, float b) { return (a > b) ? a : b; }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* Util.cpp:34-46.  Vec3f operator- and operator/ are component-wise
 * (Math.hpp), min(t0,t1)/max(t0,t1) component-wise selects (Math.hpp:186-187),
 * .max()/.min() fold left from component 0 (Math.hpp:146-147). */
void orc_ray_box(const float lo[3], const float hi[3], const OrcRay* ray, float out[2])
{
    float t0x = (lo[0] - ray->ox) / ray->dx;
    float t0y = (lo[1] - ray->oy) / ray->dy;
    float t0z = (lo[2] - ray->oz) / ray->dz;
    float t1x = (hi[0] - ray->ox) / ray->dx;
    float t1y = (hi[1] - ray->oy) / ray->dy;
    float t1z = (hi[2] - ray->oz) / ray->dz;

    float mnx = fw_min(t0x, t1x), mny = fw_min(t0y, t1y), mnz = fw_min(t0z, t1z);
    float mxx = fw_max(t0x, t1x), mxy = fw_max(t0y, t1y), mxz = fw_max(t0z, t1z);

    float tmin = mnx; tmin = fw_max(tmin, mny); tmin = fw_max(tmin, mnz);
    float tmax = mxx; tmax = fw_min(tmax, mxy); tmax = fw_min(tmax, mxz);
    out[0] = tmin;
    out[1] = tmax;
}

/* Math.hpp:185: r = 0; r += a[i]*b[i] for i = 0..3 (left to right, no FMA). */
static inline float dot4(const float a[4], float bx, float by, float bz, float bw)
{
    float r = 0.0f;
    r += a[0] * bx;
    r += a[1] * by;
    r += a[2] * bz;
    r += a[3] * bw;
    return r;
}

/* Util.cpp:99-127. */
float orc_ray_triangle_woop(const float z[4], const float u4[4], const float v4[4],
                            const OrcRay* ray, float* uv)
{
    /* orig = (origin, 1), dir = (direction, 0)   Util.cpp:103-104 */
    float Oz = z[3] - ray->ox * z[0] - ray->oy * z[1] - ray->oz * z[2];
    /* dot(dir, zpleq): this = dir, v = zpleq  -> r += dir[i]*zpleq[i] */
    float dz = 0.0f;
    dz += ray->dx * z[0];
    dz += ray->dy * z[1];
    dz += ray->dz * z[2];
    dz += 0.0f * z[3];
    float ooDz = 1.0f / dz;
    float t = Oz * ooDz;
    if (t > ray->tmin && t < ray->tmax) {
        float Ou = dot4(u4, ray->ox, ray->oy, ray->oz, 1.0f);
        float Du = dot4(u4, ray->dx, ray->dy, ray->dz, 0.0f);
        float u = Ou + t * Du;
        if (u >= 0) {
            float Ov = dot4(v4, ray->ox, ray->oy, ray->oz, 1.0f);
            float Dv = dot4(v4, ray->dx, ray->dy, ray->dz, 0.0f);
            float v = Ov + t * Dv;
            if (v >= 0 && (u + v) <= 1.0f) {
                if (uv) { uv[0] = u; uv[1] = v; }
                return t;
            }
        }
    }
    return FLT_MAX; /* miss = Vec3f(FW_F32_MAX...) ; caller reads bary[2] */
}

typedef struct TraceCtx {
    const uint8_t* nodes;
    const uint8_t* woop;
    const int32_t* triIndex;
    int anyHit;
    OrcTraceStats st;
    int overflow;
} TraceCtx;

/* CudaBVH.cpp:1183-1225 (VISIBLE_* variants are compiled out). */
static inline int update_hit(TraceCtx* c, OrcRay* ray, OrcResult* res, float t, int32_t index)
{
    if (t > ray->tmin && t < ray->tmax) {
        ray->tmax = t;
        res->t = t;
        res->id = index;
        if (c->anyHit) /* !m_needClosestHit */
            return 1;
    }
    return 0;
}

/* CudaBVH.cpp:1084-1126 (MASK_TRACE_EMPTY is defined at :41, so the empty-leaf
 * branch is compiled out). */
static int intersect_triangles(TraceCtx* c, int32_t node, OrcRay* ray, OrcResult* res)
{
    for (int32_t triAddr = (-node - 1);; triAddr += 3) {
        const float* w = (const float*)(c->woop + (size_t)triAddr * 16);
        if (f2u(w[0]) == 0x80000000u) {
            c->st.numLeafVisits++;
            break;
        }
        c->st.numTriTests++;
        int32_t index = c->triIndex[triAddr];
        float t = orc_ray_triangle_woop(w + 0, w + 4, w + 8, ray, NULL);
        if (update_hit(c, ray, res, t, index))
            return 1;
    }
    return 0;
}

/* CudaBVH.cpp:698-784 + :1250-1265. */
static void trace_one(TraceCtx* c, int32_t node, OrcRay* ray, OrcResult* res)
{
    int32_t stack[ORC_STACK_SIZE];
    int stackIndex = 1;
    stack[0] = 0;

    while (stackIndex > 0) {
        for (;;) {
            if (node < 0) {
                if (intersect_triangles(c, node, ray, res))
                    return;
                break;
            } else {
                const float* n = (const float*)(c->nodes + (size_t)node);
                const int32_t* ni = (const int32_t*)n;
                /* getNodeTemplate<Compact>: byte offsets 0,8,32 / 4,12,36 / 16,24,40 / 20,28,44 */
                float c0lo[3] = { n[0], n[2], n[8] };
                float c0hi[3] = { n[1], n[3], n[9] };
                float c1lo[3] = { n[4], n[6], n[10] };
                float c1hi[3] = { n[5], n[7], n[11] };
                int32_t child0 = ni[12], child1 = ni[13];

                float s0[2], s1[2];
                orc_ray_box(c0lo, c0hi, ray, s0);
                orc_ray_box(c1lo, c1hi, ray, s1);
                c->st.numInnerVisits++;

                int i0 = (s0[0] <= s0[1]) && (s0[1] >= ray->tmin) && (s0[0] <= ray->tmax);
                int i1 = (s1[0] <= s1[1]) && (s1[1] >= ray->tmin) && (s1[0] <= ray->tmax);

                if (i0 && i1) {
                    if (s0[0] > s1[0]) { /* swap(tspan), swap(childAddr)  :761-765 */
                        int32_t tmp = child0; child0 = child1; child1 = tmp;
                    }
                    node = child0;
                    if (stackIndex >= ORC_STACK_SIZE) { c->overflow = 1; return; }
                    stack[stackIndex++] = child1;
                    if (stackIndex > c->st.maxStackDepth) c->st.maxStackDepth = stackIndex;
                } else if (i0)
                    node = child0;
                else if (i1)
                    node = child1;
                else
                    break;
            }
        }
        stackIndex--;
        node = stack[stackIndex];
    }
}

static void trace_range(TraceCtx* c, const OrcRay* rays, OrcResult* results, int32_t begin, int32_t end)
{
    for (int32_t i = begin; i < end; i++) {
        OrcRay ray = rays[i];          /* local copy, CudaBVH.cpp:270 */
        OrcResult* res = &results[i];
        res->id = -1;                  /* result.clear()        :273 */
        res->t = ray.tmax;             /* result.t = ray.tmax   :274 */
        c->st.numRays++;
        trace_one(c, 0, &ray, res);    /* trace<Compact>(0,...) :288 */
        if (res->id != -1) c->st.numHits++;
        if (c->overflow) return;
    }
}

static void stats_add(OrcTraceStats* a, const OrcTraceStats* b)
{
    a->numRays += b->numRays;
    a->numInnerVisits += b->numInnerVisits;
    a->numTriTests += b->numTriTests;
    a->numLeafVisits += b->numLeafVisits;
    a->numHits += b->numHits;
    if (b->maxStackDepth > a->maxStackDepth) a->maxStackDepth = b->maxStackDepth;
}

int orc_trace_compact(const void* nodes, const void* woop, const int32_t* triIndex,
                      const OrcRay* rays, OrcResult* results, int32_t numRays,
                      int32_t anyHit, OrcTraceStats* stats)
{
    TraceCtx c;
    memset(&c, 0, sizeof(c));
    c.nodes = (const uint8_t*)nodes;
    c.woop = (const uint8_t*)woop;
    c.triIndex = triIndex;
    c.anyHit = anyHit;
    trace_range(&c, rays, results, 0, numRays);
    if (stats) { memset(stats, 0, sizeof(*stats)); stats_add(stats, &c.st); }
    return c.overflow ? -1 : 0;
}

typedef struct MtJob {
    TraceCtx c;
    const OrcRay* rays;
    OrcResult* results;
    int32_t begin, end;
} MtJob;

static void* mt_entry(void* p)
{
    MtJob* j = (MtJob*)p;
    trace_range(&j->c, j->rays, j->results, j->begin, j->end);
    return NULL;
}

int orc_trace_compact_mt(const void* nodes, const void* woop, const int32_t* triIndex,
                         const OrcRay* rays, OrcResult* results, int32_t numRays,
                         int32_t anyHit, int32_t nthreads, OrcTraceStats* stats)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    MtJob* jobs = (MtJob*)calloc((size_t)nthreads, sizeof(MtJob));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    int64_t per = ((int64_t)numRays + nthreads - 1) / nthreads;
    for (int i = 0; i < nthreads; i++) {
        MtJob* j = &jobs[i];
        j->c.nodes = (const uint8_t*)nodes;
        j->c.woop = (const uint8_t*)woop;
        j->c.triIndex = triIndex;
        j->c.anyHit = anyHit;
        j->rays = rays;
        j->results = results;
        int64_t b = per * i, e = b + per;
        if (b > numRays) b = numRays;
        if (e > numRays) e = numRays;
        j->begin = (int32_t)b;
        j->end = (int32_t)e;
        pthread_create(&th[i], NULL, mt_entry, j);
    }
    int overflow = 0;
    OrcTraceStats tot;
    memset(&tot, 0, sizeof(tot));
    for (int i = 0; i < nthreads; i++) {
        pthread_join(th[i], NULL);
        stats_add(&tot, &jobs[i].c.st);
        overflow |= jobs[i].c.overflow;
    }
    if (stats) *stats = tot;
    free(jobs);
    free(th);
    return overflow ? -1 : 0;
}

/* Per-ray visit counts (inner nodes, triangle tests) for workload analysis. */
int orc_trace_compact_counts(const void* nodes, const void* woop, const int32_t* triIndex,
                             const OrcRay* rays, OrcResult* results, int32_t numRays, int32_t anyHit,
                             int32_t* perRayInner, int32_t* perRayTris)
{
    TraceCtx c;
    memset(&c, 0, sizeof(c));
    c.nodes = (const uint8_t*)nodes;
    c.woop = (const uint8_t*)woop;
    c.triIndex = triIndex;
    c.anyHit = anyHit;
    for (int32_t i = 0; i < numRays; i++) {
        int64_t i0 = c.st.numInnerVisits, t0 = c.st.numTriTests;
        trace_range(&c, rays, results, i, i + 1);
        perRayInner[i] = (int32_t)(c.st.numInnerVisits - i0);
        perRayTris[i] = (int32_t)(c.st.numTriTests - t0);
        if (c.overflow) return -1;
    }
    return 0;
}

void orc_bruteforce_closest(const void* woop, const int32_t* triIndex, int32_t numFloat4,
                            const OrcRay* rays, OrcResult* results, int32_t numRays)
{
    const uint8_t* wb = (const uint8_t*)woop;
    for (int32_t i = 0; i < numRays; i++) {
        OrcRay ray = rays[i];
        OrcResult* res = &results[i];
        res->id = -1;
        res->t = ray.tmax;
        int32_t a = 0;
        while (a < numFloat4) {
            const float* w = (const float*)(wb + (size_t)a * 16);
            if (f2u(w[0]) == 0x80000000u) { a += 1; continue; }
            float t = orc_ray_triangle_woop(w, w + 4, w + 8, &ray, NULL);
            if (t > ray.tmin && t < ray.tmax) {
                ray.tmax = t;
                res->t = t;
                res->id = triIndex[a];
            }
            a += 3;
        }
    }
}
