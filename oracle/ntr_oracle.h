/*
 * ntr_oracle.h -- CPU ORACLE for the NTrace BVH build-and-trace hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product
 * (ntrace_amd/) never links, imports or falls back to anything in oracle/.
 *
 * It restates, in plain C99, the reference's *CPU* tracer and the semantics of
 * its LBVH builder.  Every function cites the reference file:line it follows
 * (paths relative to the reference checkout, src/...).
 *
 * PARITY UNPINNED: the reference ships no golden vectors / known-answer tests
 * for this path, and its sources cannot be compiled in this
 * image without writing stand-ins for <windows.h>, <cuda.h>, <mmsystem.h>,
 * <shlwapi.h> (src/framework/base/DLLImports.hpp:38-58 includes them
 * unconditionally from base/Math.hpp:29), so there is no oracle/_ref build.
 * The oracle is cross-checked instead against an independent numpy binary32
 * restatement (tests/np_tracer.py) and against brute-force intersection.
 *
 * Canonical arithmetic: IEEE-754 binary32, round-to-nearest-even, no FMA
 * contraction, no fast-math (build: gcc -O2 -ffp-contract=off -fno-fast-math
 * -msse2 -mfpmath=sse).  The reference host code is built by MSVC /fp:fast
 * (rt.vcxproj:106), which defines no canonical bits; the strict-IEEE evaluation
 * of the source expressions, in source order, is the definition used here.
 */
#ifndef NTR_ORACLE_H
#define NTR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/rt/Util.hpp:62-71  (32 bytes) */
typedef struct OrcRay {
    float ox, oy, oz, tmin;
    float dx, dy, dz, tmax;
} OrcRay;

/* src/rt/Util.hpp:77-87  (16 bytes) */
typedef struct OrcResult {
    int32_t id;   /* -1 = miss */
    float   t;
    int32_t padA;
    int32_t padB;
} OrcResult;

/* Counters of src/rt/cuda/CudaBVH.cpp:746-749, 1107-1111 plus what SURVEY.md
 * section 8(d) needs for the algorithmic-bytes formula. */
typedef struct OrcTraceStats {
    int64_t numRays;
    int64_t numInnerVisits;   /* = numNodeTests / 2 */
    int64_t numTriTests;
    int64_t numLeafVisits;    /* terminator reads */
    int64_t numHits;          /* rays with id != -1 */
    int64_t maxStackDepth;
} OrcTraceStats;

/* CudaBVH::trace, BVHLayout_Compact, single tree (CudaBVH.cpp:213-302).
 * nodes/woop/triIndex: byte-identical Compact buffers (CudaBVH.hpp:42-56).
 * results[i].id/.t are written; padA/padB are left untouched (the CPU tracer
 * without RayStats does not write them).  Returns 0, or -1 on traversal stack
 * overflow (reference stack is S32[100], CudaBVH.cpp:701). */
int orc_trace_compact(const void* nodes, const void* woop, const int32_t* triIndex,
                      const OrcRay* rays, OrcResult* results, int32_t numRays,
                      int32_t anyHit, OrcTraceStats* stats /* may be NULL */);

/* Same, ray range statically sharded over nthreads pthreads (bench cpu_baseline). */
int orc_trace_compact_mt(const void* nodes, const void* woop, const int32_t* triIndex,
                         const OrcRay* rays, OrcResult* results, int32_t numRays,
                         int32_t anyHit, int32_t nthreads, OrcTraceStats* stats);

/* Single-threaded trace that also returns per-ray visit counts (workload analysis). */
int orc_trace_compact_counts(const void* nodes, const void* woop, const int32_t* triIndex,
                             const OrcRay* rays, OrcResult* results, int32_t numRays, int32_t anyHit,
                             int32_t* perRayInner, int32_t* perRayTris);

/* Intersect::RayBox (Util.cpp:34-46): out[0]=tmin, out[1]=tmax. */
void orc_ray_box(const float lo[3], const float hi[3], const OrcRay* ray, float out[2]);

/* Intersect::RayTriangleWoop (Util.cpp:99-127): returns t or FLT_MAX (miss);
 * uv (may be NULL) receives u,v on hit. */
float orc_ray_triangle_woop(const float z[4], const float u[4], const float v[4],
                            const OrcRay* ray, float* uv);

/* Brute force: closest accepted t over ALL Woop triangles of a Compact triWoop
 * buffer (skipping terminators), ignoring the BVH.  Used to cross-check the
 * traversal semantics. numFloat4 = woop bytes / 16. */
void orc_bruteforce_closest(const void* woop, const int32_t* triIndex, int32_t numFloat4,
                            const OrcRay* rays, OrcResult* results, int32_t numRays);

/* ---- LBVH builder restatement (src/rt/bvh/HLBVH) ---------------------------- */

typedef struct OrcLbvh {
    void*    nodes;      int64_t nodesBytes;     /* Compact nodes, 64 B each  */
    void*    woop;       int64_t woopBytes;      /* Compact triWoop            */
    int32_t* triIndex;   int64_t triIndexBytes;  /* Compact triIndex           */
    uint32_t* mortonSorted;                      /* n sorted keys              */
    int32_t*  triSorted;                         /* n sorted triangle ids      */
    int32_t  numInner, numLeaves, numLevels;
} OrcLbvh;

/* calcMorton (emitTreeKernel.cu:655-691) */
void orc_lbvh_morton(int32_t numTris, const int32_t* triVtxIdx /*3/tri*/, const float* vtxPos /*3/vtx*/,
                     const float sceneMin[3], const float sceneMax[3], uint32_t* keys, int32_t* idx);

/* calcWoop (emitTreeKernel.cu:574-635): 12 floats per triangle, original order. */
void orc_lbvh_woop(int32_t numTris, const int32_t* triVtxIdx, const float* vtxPos, float* woop12);

/* buildLBVH (HLBVHBuilder.cpp:451-593) in canonical (deterministic) node order:
 * breadth-first by level, queue order within a level.  See oracle/README. */
int orc_lbvh_build(int32_t numTris, const int32_t* triVtxIdx, int32_t numVerts, const float* vtxPos,
                   const float sceneMin[3], const float sceneMax[3],
                   int32_t leafSize, float epsilon, OrcLbvh* out);
void orc_lbvh_free(OrcLbvh* b);

/* Canonical DFS signature of a Compact BVH (topology + child boxes + per-leaf
 * sorted triangle-id lists); two BVHs that differ only in node numbering / leaf
 * placement yield the same 64-bit value.  SURVEY.md section 8(a) parity note for L3. */
uint64_t orc_bvh_canonical_hash(const void* nodes, int64_t nodesBytes, const void* woop,
                                const int32_t* triIndex, int32_t hashWoop);

#ifdef __cplusplus
}
#endif
#endif
