/*
 * ntrace_amd.h -- C-ABI of the MI355X-native NTrace tracer backend (libntrace_amd.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Each entry point cites the reference interface it replaces (paths relative to
 * the NTrace checkout).  The reference-side binding is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns NTR_OK (0) or a negative NtrStatus; the message of
 *     the last failure on the calling thread is ntr_last_error().  The C++ host
 *     mirror (ntrace_amd/host) maps non-zero to FW::fail(), as the reference's
 *     CudaModule::checkError does (src/framework/gpu/CudaModule.cpp).
 *   - "d_" pointers are device (HIP) pointers owned by the caller (Buffer in the
 *     reference, src/framework/gpu/Buffer.hpp:107-113); the library never frees
 *     or retains them past the call (or past stream completion for async calls).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - calls are thread-compatible: one caller per device at a time.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with NTR_ERR_NO_DEVICE.
 */
#ifndef NTRACE_AMD_H
#define NTRACE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTR_API __attribute__((visibility("default")))

typedef enum NtrStatus {
    NTR_OK = 0,
    NTR_ERR_INVALID = -1,     /* bad argument                                  */
    NTR_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at init   */
    NTR_ERR_HIP = -3,         /* a HIP call failed (message has the HIP error) */
    NTR_ERR_LAYOUT = -4,      /* "CudaBVHTracer: Incorrect BVH layout!"        */
    NTR_ERR_UNKNOWN_KERNEL = -5,
    NTR_ERR_OVERFLOW = -6,    /* traversal stack / workspace overflow          */
    NTR_ERR_NOMEM = -7
} NtrStatus;

/* src/rt/kernels/CudaTracerKernels.hpp:52-63 */
typedef enum NtrBVHLayout {
    NTR_BVHLayout_AOS_AOS = 0,
    NTR_BVHLayout_AOS_SOA,
    NTR_BVHLayout_SOA_AOS,
    NTR_BVHLayout_SOA_SOA,
    NTR_BVHLayout_Compact,
    NTR_BVHLayout_Compact2,
    NTR_BVHLayout_CPU,
    NTR_BVHLayout_Max
} NtrBVHLayout;

/* src/rt/kernels/CudaTracerKernels.hpp:69-75 (KernelConfig) */
typedef struct NtrKernelConfig {
    int32_t bvhLayout;
    int32_t blockWidth;            /* wave64: 64                           */
    int32_t blockHeight;           /* waves per workgroup                  */
    int32_t usePersistentThreads;
} NtrKernelConfig;

/* src/rt/Util.hpp:62-71 */
typedef struct NtrRay {
    float ox, oy, oz, tmin;
    float dx, dy, dz, tmax;
} NtrRay;

/* src/rt/Util.hpp:77-87.  The tracer writes id and t; padA/padB receive the
 * barycentrics (u,v) bit patterns on a hit like STORE_RESULT
 * (CudaTracerKernels.hpp:222) and are 0 on a miss.  Parity is on (id,t). */
typedef struct NtrRayResult {
    int32_t id;
    float   t;
    int32_t padA;
    int32_t padB;
} NtrRayResult;

/* ---- library / device ------------------------------------------------------ */

NTR_API const char* ntr_last_error(void);
NTR_API int         ntr_version(void);

/* Thin device-memory plumbing for the C++ Buffer mirror
 * (src/framework/gpu/Buffer.cpp:238- cuMemAlloc/cuMemcpy* call sites). */
NTR_API int ntr_device_count(int* count);
NTR_API int ntr_set_device(int device);
NTR_API int ntr_malloc(void** d_ptr, size_t bytes);
NTR_API int ntr_free(void* d_ptr);
NTR_API int ntr_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
NTR_API int ntr_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
NTR_API int ntr_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);
NTR_API int ntr_memset(void* d_dst, int value, size_t bytes, void* stream);
NTR_API int ntr_stream_synchronize(void* stream);

/* ---- tracer ---------------------------------------------------------------- */

/* Replaces the per-kernel-file `queryConfig` launch + g_config read-back
 * (src/rt/cuda/CudaBVHTracer.cpp:52-84).  `kernelName` keeps the reference's
 * file-name selectors: "fermi_speculative_while_while",
 * "tesla_persistent_while_while", "tesla_persistent_speculative_while_while",
 * "kepler_dynamic_fetch" (all mapped to precompiled CDNA4 variants).
 *
 * The answer describes the NAMED body (per-ray kernel: 64-thread workgroups, not persistent; tesla_* / kepler_*: persistent waves in
 * 256-thread workgroups) and every name consumes BVHLayout_Compact.  A name selects semantics, and the semantics of all names are
 * identical here (same hit records, bit for bit); which body actually traces a batch follows the batch (csrc/trace_plan.h ROUTING):
 * any-hit launches run the per-ray body under every name, closest-hit launches of >= 2^20 rays are launched as both bodies and the
 * device's batch word (ntr_predict_batch_coherence) decides which one works -- the per-ray body on coherent batches, the persistent
 * dynamic-fetch body on incoherent ones --, smaller launches run the named body.  NTR_TRACE_ROUTE=0 (environment, read once) forces
 * the named body always; ntr_trace_plan() reports the decision for a given batch. */
NTR_API int ntr_query_config(const char* kernelName, NtrKernelConfig* config);

/* Replaces the `trace_bvh` launch of CudaBVHTracer::traceBatch
 * (src/rt/cuda/CudaBVHTracer.cpp:88-168; signature TRACE_FUNC_BVH,
 * src/rt/kernels/CudaTracerKernels.hpp:99-112; nodesB-D / trisB-C exist only for
 * the SOA layouts and are dropped).
 *
 *   numRays == 0          -> NTR_OK, *seconds = 0 (CudaBVHTracer.cpp:92-94)
 *   layout != the kernel's desired layout -> NTR_ERR_LAYOUT (:99-100)
 *   seconds != NULL       -> launch is bracketed by HIP events on `stream`, the
 *                            call blocks and returns the kernel's GPU time in
 *                            seconds (CudaKernel::launchTimed, CudaKernel.cpp:188-221)
 *   seconds == NULL       -> asynchronous launch on `stream`.
 *
 * nodesBytes / triWoopBytes are the buffer extents the reference binds as texture sizes
 * (setTexRef(..., size), CudaBVHTracer.cpp:142-150); here they bound the kernels' buffer
 * descriptors: triWoopBytes < 4 GiB; nodesBytes <= 0x76543200, because Compact child pointers are signed
 * 32-bit byte offsets and 0x76543210 is the traversal sentinel (EntrypointSentinel, CudaTracerKernels.hpp:38).
 *
 * Results follow the reference CPU tracer bit for bit in (id, t): miss =
 * (-1, ray.tmax) (CudaBVH.cpp:273-274).  `bvhFlags`: 0, or hints from
 * ntr_bvh_validate() (hints only select between two exact code paths).
 *
 * Closest-hit launches of the per-ray kernel with at least 2^20 rays dispatch their 256-ray blocks in
 * predicted-cost order (two small launches in front of the trace kernel, inside the timed bracket; the
 * order never changes a result; NTR_TRACE_PREDICT=0 disables it).  Launches of
 * "kepler_dynamic_fetch" end with ray splitting: once the ray pool is dry, lanes without a ray traverse
 * stack entries of the wave's live rays, and a helper's result counts only where it provably is the
 * result the ray alone would have computed (csrc/trace_split.h; NTR_TRACE_SPLIT_SLICE=0 disables it) --
 * records are the reference CPU tracer's either way.  An asynchronous call can be captured
 * into a HIP graph and replayed once a first call on that stream has allocated its scratch buffers. */
NTR_API int ntr_trace_bvh(const char* kernelName, int32_t numRays, int32_t anyHit,
                          const NtrRay* d_rays, NtrRayResult* d_results,
                          const void* d_nodes, int64_t nodesBytes,
                          const void* d_triWoop, int64_t triWoopBytes, const int32_t* d_triIndex,
                          int32_t layout, uint32_t bvhFlags, void* stream, float* seconds);

/* Device status of the launches made since the last check, for callers of the ASYNCHRONOUS form
 * (seconds == NULL), which cannot report it themselves.  The reference's launches are synchronous and its
 * kernels silently assume a 64-entry stack (STACK_SIZE, CudaTracerKernels.hpp:37); here a ray whose
 * traversal stack would exceed 104 entries (LDS + scratch, more than the CPU tracer's 100, CudaBVH.cpp:701)
 * sets a sticky bit in a per-device status word instead of corrupting memory -- its hit record is then not
 * to be trusted.  ntr_trace_status waits for `stream`, returns the word in *statusBits (may be NULL), clears
 * it, and returns NTR_ERR_OVERFLOW if the overflow bit was set (bit 0), NTR_OK otherwise.  Timed calls
 * (seconds != NULL) perform the same check themselves.  The word is shared by all streams of the device. */
NTR_API int ntr_trace_status(void* stream, uint32_t* statusBits);

/* Measurement aid (bench.py extras.gather_roof; no counterpart in the reference): every one of `waves` x `lanesPerWave` lanes walks
 * a dependent chain of `steps` random 64-byte records of a `tableBytes` table (four 16-byte loads per record, the next index a hash of
 * the bytes just loaded) -- the memory side of a divergent traversal without its arithmetic.  *seconds = best of three launches;
 * waves x lanesPerWave x steps / seconds is the record rate.  On a table larger than the Infinity Cache with a full grid this is the
 * practical roof of an HBM-resident BVH's fetches (56 G records/s = 3.6 TB/s on MI355X, whatever the record size: DESIGN.md 4.1). */
NTR_API int ntr_selftest_gather_rate(int64_t tableBytes, int32_t waves, int32_t lanesPerWave, int32_t steps, void* stream, float* seconds);

/* HIP graphs.  An asynchronous ntr_trace_bvh (seconds == NULL) may be captured into a HIP graph.  Nothing can be allocated
 * during a capture, and the scratch of a captured launch must outlive the graph, so the library hands such a launch scratch
 * of its OWN from per-device stores that are filled outside captures and never recycled while pinned:
 *   - pool counters of the persistent kernels: 192 captured launches per device;
 *   - dispatch-order prediction scratch (closest-hit per-ray launches of >= 2^20 rays): private to each captured launch
 *     (a replay on any stream never shares it with a live launch); every live launch of a size keeps 4 spares of that size
 *     ready, ntr_trace_graph_reserve(launches, numRays) provisions more (48 entries per device in all);
 *   - the top-of-tree table of a BVH (16 BVHs per device): trace the BVH once, or call ntr_bvh_validate, before capturing.
 * The predicted order is an optimisation: a captured launch that finds no spare scratch is captured in buffer order
 * (no error).
 * When one of the other stores is exhausted the capture-time call fails with NTR_ERR_NOMEM / NTR_ERR_INVALID and says so.  A host that
 * re-captures graphs (e.g. every frame) calls ntr_trace_graph_release_all() once the graphs holding earlier captures are
 * destroyed: it waits for the device and returns every pinned resource of the current device to its store. */
NTR_API int ntr_trace_graph_reserve(int32_t launches, int32_t numRays);
NTR_API int ntr_trace_graph_release_all(void);

/* Scheduling state is kept PER DEVICE (one mutex, 96 automatic hints, 48 prediction scratches, 16 top-of-tree tables each), so the
 * host threads of a thread-per-GPU driver (ntr_dist_init_all) neither share entries nor contend for one lock.  Within a device an
 * entry belongs to the stream that created it and is recycled only by that stream (the launch path never synchronises or records an
 * event).  ntr_stream_release(stream): call before destroying a stream -- waits for it and returns the automatic hints and the
 * prediction scratch it owns on the current device.  Without it the entries of dead streams stay allocated for the life of the
 * process; a device whose 96 hint entries all belong to dead streams gives later streams buffer / predicted order only (about
 * 9 % on re-traced batches), never an error. */
NTR_API int ntr_stream_release(void* stream);

/* CPU-side self test of the automatic-hint table (no device needed): `devices` simulated devices each trace `keysPerDevice` distinct
 * batches on a stream of their own, `rounds` times; hintedLastRound[d] = batches of device d that found their hint entry in the
 * last round (keysPerDevice for every device while keysPerDevice <= 96: no device is starved by the others). */
NTR_API int ntr_selftest_auto_hint_table(int32_t devices, int32_t keysPerDevice, int32_t rounds, int32_t* hintedLastRound);

/* The PLAN of a trace launch (diagnostic; no device needed): everything ntr_trace_bvh decides from the tunables, the kernel name and the
 * batch's sizes and flags alone -- which kernel variant, how many workgroups, which loop, pool geometry, and which scheduling aids
 * (automatic hint, dispatch-order prediction, pool depth decided on the device) the launch will ask for.  ntr_trace_bvh computes exactly
 * this (csrc/trace_plan.h) before it touches the device; what depends on run-time state (which hint entry, which prediction scratch)
 * is not part of it.  No counterpart in the reference (CudaBVHTracer::traceBatch sizes its grid inline, CudaBVHTracer.cpp:152-160). */
typedef struct NtrTracePlan {
    int32_t variant;            /* NTR_VARIANT_* after the stats override (csrc/trace_kernels.h) */
    int32_t launchVariant;      /* the kernel that is launched */
    int32_t launchBlocks;       /* its workgroups */
    int32_t numBlocks;          /* per-ray kernels: 256-ray blocks (unit of dispatch order and cost feedback); persistent: the grid */
    int32_t orderBlocks;        /* 256-ray blocks of the batch */
    int32_t chunk, fetchThreshold, leafSwitchBelow, octant, flatFetch, uniformPrologue, splitSlice;
    int32_t numHeads, shardRays, numBlocksIncoherent;   /* persistent kernels: pool heads, rays per head, grid of an incoherent batch */
    int32_t numBlocksDivergent; /* ... of a batch whose rays start together and wander apart (NTR_BATCH_DIVERGENT only) */
    int32_t wholeWave;          /* persistent, dynamic fetch: waves start in whole-wave mode and switch per wave (csrc/trace_kernels.hip) */
    int32_t prefetchAfter;      /* persistent: iterations into a chunk after which a wave posts the dequeue of its next one (< 0: never) */
    int32_t unified;            /* persistent: unified-step loop (kepler_dynamic_fetch) */
    int32_t minipool;           /* closest-hit per-ray launch that may run as wave-private ray pools */
    int32_t poolKConst;         /* pool depth when it is not decided on the device */
    int32_t poolKFromDevice;    /* it is: the prediction of this launch, or the batch's hint, holds K */
    int32_t minipoolWide;       /* K of a batch the device finds incoherent */
    int32_t hintable;           /* the variant honours a scheduling hint */
    int32_t useAutoHint;        /* the batch is looked up in the library's own hint table */
    int32_t predictable;        /* dispatch-order prediction applies when no measured order is at hand */
    int32_t persistentOrder;    /* ... for a persistent launch: the pool is handed out in predicted order */
    int32_t probeOnRefresh;     /* a hinted batch's pool depth is estimated again on the hint's refresh launches */
    int32_t coherentRoute;      /* routing by coherence (ntr_query_config): 0 = the named body; 1 = BOTH bodies are launched and the device's
                                 * batch word decides which one works (closest-hit launches large enough for the estimate); 2 = the per-ray
                                 * body under a persistent name (any-hit launches) */
    int32_t persistentVariant, persistentBlocks, persistentFetchThreshold;   /* the persistent side of a launch (named, or routed) */
    int32_t perrayBlocks, perrayFetchThreshold;                              /* the per-ray side of a routed launch */
} NtrTracePlan;
#define NTR_PLAN_FLAG_STATS 1        /* ntr_trace_bvh_stats */
#define NTR_PLAN_FLAG_CAPTURING 2    /* the stream is being captured into a HIP graph */
#define NTR_PLAN_FLAG_CALLER_HINT 4  /* the caller passes an NtrSchedHint */
NTR_API int ntr_trace_plan(const char* kernelName, int32_t numRays, int32_t anyHit, uint64_t nodesAddr, int64_t nodesBytes,
                           uint64_t triWoopAddr, int64_t triWoopBytes, uint32_t bvhFlags, int32_t numCUs, int32_t flags,
                           NtrTracePlan* plan);
/* One launch in the life of a scheduling hint: out[0] = the hint's pool-depth words are cleared, out[1] = the launch records per-block
 * costs (refresh), out[2] = the launch is dispatched in the hint's order. */
NTR_API int ntr_trace_plan_hint_step(int32_t valid, int32_t predicted, int32_t uses, int32_t out[3]);

/* Re-reads the NTR_* environment tunables.  They are read once, at first use; sweep scripts
 * that change a variable inside one process call this afterwards.  Not needed by applications. */
NTR_API int ntr_tunables_reload(void);

/* Scheduling hint (no counterpart in the reference; its kernels take rays in buffer order).
 *
 * The per-ray kernel's launch time is set by where the long-lived waves start: blocks of 256 rays
 * that traverse deep geometry live several times longer than the average block, and when they start
 * late they are the tail of the launch.  A hint object remembers, per block, how long its slowest
 * wave lived in the previous trace of the SAME logical batch (the benchmark protocol traces every
 * batch 1 + warmup + measure times, App.cpp:955-958; an interactive renderer traces nearly the same
 * rays frame after frame) and makes the next launch start the heaviest cost classes first.  Only the
 * ORDER in which blocks are dispatched changes: every ray is traced exactly as without a hint and
 * the hit records are identical.  A hint made for other rays is harmless (any order is valid); it
 * adapts after a launch or two.  The caller owns the hint and passes one per logical batch; it is
 * bound to the first (numRays) it is used with and re-initialises itself when that changes.
 * The persistent kernels honour a hint too (round 6): their pool is handed out in the hint's order and whole-wave chunks record
 * their lives as block costs.  ONE STREAM AT A TIME: a hint's arrays are read by the launches it was passed to and rewritten when it is
 * rebound (another ray count) or refreshed; use a hint on one stream only, or synchronise that stream before passing the hint to a
 * launch on another (the launch path itself never synchronises or records events: ADVICE r05). */
typedef struct NtrSchedHint NtrSchedHint;
NTR_API int ntr_sched_hint_create(NtrSchedHint** out);
NTR_API int ntr_sched_hint_destroy(NtrSchedHint* hint);
/* Forget what was measured (e.g. after a camera cut). */
NTR_API int ntr_sched_hint_reset(NtrSchedHint* hint);
/* ntr_trace_bvh with a scheduling hint (hint == NULL: identical to ntr_trace_bvh). */
NTR_API int ntr_trace_bvh_hinted(const char* kernelName, int32_t numRays, int32_t anyHit,
                                 const NtrRay* d_rays, NtrRayResult* d_results,
                                 const void* d_nodes, int64_t nodesBytes,
                                 const void* d_triWoop, int64_t triWoopBytes, const int32_t* d_triIndex,
                                 int32_t layout, uint32_t bvhFlags, void* stream, float* seconds,
                                 NtrSchedHint* hint);

/* A hint can also start from a PREDICTION instead of a measurement: ntr_sched_hint_predict binds the hint to a batch of numBlocks 256-ray
 * blocks and orders them by the caller's per-block cost estimates (any monotone score; heaviest class first).  The next launch with the
 * hint uses that order, measures, and refines it as usual.  For secondary batches the estimate comes from the tree itself:
 *   ntr_bvh_leaf_depths        once per BVH: d_depthByTri[t] = depth of the leaf that holds triangle t (number of inner nodes from the
 *                              root down; triangles in no leaf: 0).  One small launch per tree level; blocking.  *levels = levels walked.
 *   ntr_secondary_block_costs  per batch, beside ray generation: d_blockCost[b] = the deepest leaf among the input rays (primary hits
 *                              d_inResults[first .. first + count)) whose numSamples output rays fall into block b; misses count 0.
 *                              d_blockCost holds (count * numSamples + 255) / 256 words; the call clears them first.
 * Short secondary rays mostly pay for descending to where they start: on the bench frame's AO batches this order recovers what the
 * order learned from a previous launch gives (-9 % launch time), without a previous launch (EXPERIMENTS.md).  No counterpart in the
 * reference, which generates secondary rays (RayGen::ao, src/rt/ray/RayGen.cpp) and traces them in buffer order. */
NTR_API int ntr_sched_hint_predict(NtrSchedHint* hint, const uint32_t* d_blockCost, int32_t numBlocks, void* stream);
NTR_API int ntr_bvh_leaf_depths(const void* d_nodes, int64_t nodesBytes, const void* d_triWoop, int64_t triWoopBytes,
                                const int32_t* d_triIndex, int32_t numTris, int32_t* d_depthByTri, int32_t* levels, void* stream);
NTR_API int ntr_secondary_block_costs(const NtrRayResult* d_inResults, int32_t first, int32_t count, int32_t numSamples,
                                      const int32_t* d_depthByTri, int32_t numTris, uint32_t* d_blockCost, void* stream);

/* Cost estimate of every 256-ray block of a batch, without tracing it: d_blockCost[b] = number of boxes of the BVH's top-of-tree
 * table (the child boxes of the nodes of depth <= 9) that the block's sample ray (its 100th) intersects -- the predictor behind
 * the dispatch order of large closest-hit launches (Spearman 0.86-0.89 against the true block cost on primary batches).  No
 * counterpart in the reference; a multi-GPU host uses it to cut a frame into ranges of equal predicted cost instead of equal
 * ray count (ntrace_amd/dist.py FramePlan).  d_blockCost: (numRays + 255) / 256 words.  Asynchronous on `stream`. */
NTR_API int ntr_predict_block_costs(int32_t numRays, const NtrRay* d_rays, const void* d_nodes, int64_t nodesBytes,
                                    uint32_t* d_blockCost, void* stream);

/* Coherence estimate of a batch, without tracing it, from two sample rays (the 100th and the 227th) of every 256-ray block:
 * d_out[0] = blocks whose samples start further apart than 1/8 of the scene's extent, d_out[1] = the divergence score: 4 for every block
 * whose samples start together but point more than 60 degrees apart AND reach further than 1/8 of the scene's extent (bounce rays of a
 * diffuse batch; the short rays of an AO batch do not count) + 1 for every block whose sample ray is degenerate (a missed pixel's
 * secondary ray: no say), d_out[2] = the batch word large closest-hit launches derive on the device: bits 0-15 the pool K
 * of the per-ray kernel (1 = one ray per lane; K > 1 = a wave owns K x 64 rays and refills its finished lanes from them: chosen when
 * origins are scattered in at least half of the blocks -- 4 on trees of 32 MB of nodes and more for batches of 1.5 M rays and more,
 * else 2), bit 16 (NTR_BATCH_DIVERGENT) set when 4 x d_out[0] + d_out[1] reaches the number of blocks (a quarter of the blocks with live
 * rays are incoherent one way or the other).  K > 1 or bit 16 =
 * "incoherent": such a batch is traced by the persistent dynamic-fetch body under every kernel name, a coherent one by the per-ray
 * body (ntr_query_config).  The launch itself does not call this -- its dispatch-order prediction computes the same words -- it is the query
 * for tests and for hosts that plan batches.  No counterpart in the reference.  d_out: 3 words.  Asynchronous on `stream`. */
NTR_API int ntr_predict_batch_coherence(int32_t numRays, const NtrRay* d_rays, const void* d_nodes, int64_t nodesBytes,
                                        uint32_t* d_out, void* stream);

/* Traversal counters: the reference's RayStats (src/rt/bvh/BVH.hpp:44-60), filled by its
 * CPU tracer at src/rt/cuda/CudaBVH.cpp:746-757 and 1107-1111.  numInnerVisits =
 * numNodeTests / 2.  These define the algorithmic bytes of a batch:
 *   48*numRays + 64*numInnerVisits + 48*numTriTests + 16*numLeafVisits + 4*numHits. */
typedef struct NtrTraceStats {
    int64_t numRays;
    int64_t numInnerVisits;
    int64_t numTriTests;
    int64_t numLeafVisits;   /* leaf terminators read */
    int64_t numHits;
} NtrTraceStats;

/* ntr_trace_bvh through an instrumented kernel: same results, plus the counters.
 * Blocking; not a timed path. */
NTR_API int ntr_trace_bvh_stats(const char* kernelName, int32_t numRays, int32_t anyHit,
                                const NtrRay* d_rays, NtrRayResult* d_results,
                                const void* d_nodes, int64_t nodesBytes,
                                const void* d_triWoop, int64_t triWoopBytes, const int32_t* d_triIndex,
                                int32_t layout, uint32_t bvhFlags, void* stream, NtrTraceStats* stats);

/* One pass over a Compact node buffer computing hint flags for ntr_trace_bvh:
 *   NTR_BVH_FINITE   every box coordinate is finite and |x| < 2^100.
 *   NTR_BVH_FASTDIV  every box coordinate has |x| < 2^55: together with a per-ray check this
 *                    is the range in which the kernels' refactored divide is the hardware
 *                    divide (see trace_kernels.hip).
 *   NTR_BVH_NOTINY   every box coordinate is 0 or |x| >= 2^-93 (lets rays with an exactly-zero
 *                    origin component use the same path).
 *   NTR_BVH_ORDERED  every child box has lo <= hi on each axis: for a wave whose rays share the signs of their direction
 *                    components the kernels then know which of a slab's two quotients is the smaller one without comparing.
 *   NTR_BVH_WIDE_LEAVES  leaves hold two or more triangles on average (the device LBVH with leafSize 8; a host SAH tree built with
 *                    leaf preferences (1,1) has one): the per-ray kernel then advances every lane by one node OR one triangle
 *                    per iteration instead of alternating between a node phase and a leaf phase (trace_kernels.hip). */
#define NTR_BVH_FINITE 1u
#define NTR_BVH_FASTDIV 2u
#define NTR_BVH_NOTINY 4u
#define NTR_BVH_ORDERED 8u
#define NTR_BVH_WIDE_LEAVES 16u
/* bit 16 of the batch word (ntr_predict_batch_coherence d_out[2]): long rays that start together and point apart in a quarter of the blocks */
#define NTR_BATCH_DIVERGENT 0x10000u
NTR_API int ntr_bvh_validate(const void* d_nodes, int64_t nodesBytes, uint32_t* flags, void* stream);

/* Device self test: counts quotients x[i]/d[j] for which the FAST divide differs from the
 * hardware `/` (must be 0 inside the FASTDIV range).  Diagnostic, blocking. */
NTR_API int ntr_selftest_division(const float* d_x, int32_t nx, const float* d_d, int32_t nd,
                                  uint32_t* mismatches, void* stream);
/* The same on the quotients that are hardest to round: for every significand D in [2^23, 2^24) the significands X whose quotient X / D
 * lies within 8 / (2^24 D) of a midpoint of two neighbouring floats (enumerated on the device, both quotient binades; what
 * scripts/studies/div_one_correction_check.py checks in exact integer arithmetic), as x = X 2^(xExp + i - 23), d = +-D 2^(dExp + j - 23),
 * i, j = 0..3.  pairs = quotients tested, mismatches = FAST differs from `/` (must be 0).  Diagnostic, blocking. */
NTR_API int ntr_selftest_division_hard(int32_t xExp, int32_t dExp, uint64_t* pairs, uint64_t* mismatches, void* stream);

/* ---- ray production (callers of the hot path) -------- */

/* PixelTable::recalculate (src/rt/ray/PixelTable.cpp:57-143): 8x8 pixel blocks in Morton
 * order then edge stripes.  Either output may be NULL. */
NTR_API int ntr_pixel_table(int32_t w, int32_t h, int32_t* d_indexToPixel, int32_t* d_pixelToIndex, void* stream);

/* rayGenPrimaryKernel (src/rt/ray/RayGenKernels.cu:77-125; RayGenPrimaryInput,
 * RayGenKernels.hpp:37-49).  nscreenToWorld is row-major 4x4.  kernelSeed is the value the
 * reference kernel receives: 0, or Random(seed).getU32() (RayGen.cpp:66). */
NTR_API int ntr_raygen_primary(NtrRay* d_rays, int32_t* d_idToSlot, int32_t* d_slotToID,
                               const int32_t* d_indexToPixel, const float origin[3],
                               const float nscreenToWorld[16], int32_t w, int32_t h, float maxDist,
                               uint32_t kernelSeed, void* stream);

/* rayGenAOKernel (src/rt/ray/RayGenKernels.cu:129-236; RayGenAOInput, RayGenKernels.hpp:53-66):
 * numSamples cosine-hemisphere rays per input ray [firstInputSlot, +numInputRays); rays of
 * missed inputs are degenerate (tmax = -1).  d_triNormals: 3 floats per scene triangle. */
NTR_API int ntr_raygen_ao(NtrRay* d_outRays, int32_t* d_outIDToSlot, int32_t* d_outSlotToID,
                          const NtrRay* d_inRays, const NtrRayResult* d_inResults,
                          const float* d_triNormals, int32_t firstInputSlot, int32_t numInputRays,
                          int32_t numSamples, float maxDist, uint32_t kernelSeed, void* stream);

/* rayGenShadowKernel (src/rt/ray/RayGenKernels.cu:240-301; RayGen::shadow, RayGen.cpp:114-150): numSamples rays per input ray
 * [firstInputSlot, +numInputRays) from its hit point (backed off 1e-2 along the ray) towards quasi-random points of the cube of
 * half-edge lightRadius around lightPos; tmax = the distance to that point, rays of missed inputs are degenerate (tmax = -1).
 * kernelSeed as for ntr_raygen_ao (Random(seed).getU32(), RayGen.cpp:139).  The output batch is an any-hit batch. */
NTR_API int ntr_raygen_shadow(NtrRay* d_outRays, int32_t* d_outIDToSlot, int32_t* d_outSlotToID, const NtrRay* d_inRays,
                              const NtrRayResult* d_inResults, int32_t firstInputSlot, int32_t numInputRays, int32_t numSamples,
                              const float lightPos[3], float lightRadius, uint32_t kernelSeed, void* stream);

/* countHitsKernel (src/rt/cuda/RendererKernels.cu:174-226; Renderer::getTotalNumRays,
 * Renderer.cpp:676-709): number of results with id != -1.  Blocking. */
NTR_API int ntr_count_hits(const NtrRayResult* d_results, int32_t numRays, int32_t* count, void* stream);

/* ---- on-device LBVH build ------------------------------------------------------------------ */

/* Result of ntr_lbvh_build: exact buffer sizes (the reference resizes its buffers to these,
 * HLBVHBuilder.cpp:382-386) and per-phase GPU times (HLBVHBuilder::getGPUTime, :571-573). */
typedef struct NtrLbvhResult {
    int32_t numNodes, numLeaves, numLevels, pad;
    /* extents of what was written.  nodesBytes >= 64 * numNodes and triWoopBytes >= 16 * (3 * numTris + numLeaves): equality except
     * where the depth rule (level bit 0, emitTreeKernel.cu:289-292) made a leaf of more than leafSize equal Morton codes -- the node
     * indices and terminator slots set aside inside such a leaf stay unused and are zero-filled. */
    int64_t nodesBytes, triWoopBytes, triIndexBytes;
    float   seconds;                                   /* whole build, GPU time          */
    /* phase times.  Default (bottom-up) path: Morton codes + digit histograms; the four sort passes; woopMs ~ 0; emitMs =
     * leaf marks + their prefix counts; refitMs = bottom-up emit writing nodes, boxes, Woop rows (agglomerate kernels + equal-key runs).
     * Legacy paths: per-triangle box terms (per-level path: Woop rows) in woopMs; emitMs = top pass (per-level: all levels);
     * refitMs = subtree emit + refit + top refit + Woop placement (per-level: refit). */
    float   mortonMs, sortMs, woopMs, emitMs, refitMs;
} NtrLbvhResult;

/* Worst-case output sizes for numTris triangles (what HLBVHBuilder allocates before the build,
 * HLBVHBuilder.cpp:532-538, 772-784). */
NTR_API int ntr_lbvh_capacity(int32_t numTris, int64_t* nodesBytes, int64_t* triWoopBytes, int64_t* triIndexBytes);

/* HLBVHBuilder::buildLBVH (src/rt/bvh/HLBVH/HLBVHBuilder.cpp:451-593; the path taken for
 * !hlbvh || hlbvhBits == 10, :44-47): Morton codes -> stable radix sort -> Woop rows -> level-by-level
 * emit -> bottom-up refit, all on the device, into caller-owned BVHLayout_Compact buffers of at least
 * ntr_lbvh_capacity() bytes.  d_triVtxIndex: 3 ints per triangle, d_vtxPos: 3 floats per vertex
 * (Scene::getTriVtxIndexBuffer / getVtxPosBuffer), sceneMin/Max = Scene::getBBox.  Renderer passes
 * leafSize 8, epsilon 0.001 (Renderer.cpp:203-207).  Blocking. */
NTR_API int ntr_lbvh_build(int32_t numTris, const int32_t* d_triVtxIndex, int32_t numVerts, const float* d_vtxPos,
                           const float sceneMin[3], const float sceneMax[3], int32_t leafSize, float epsilon,
                           void* d_nodes, int64_t nodesCapacity, void* d_triWoop, int64_t triWoopCapacity,
                           int32_t* d_triIndex, int64_t triIndexCapacity, NtrLbvhResult* result, void* stream);

/* The builder keeps one grow-only scratch workspace per device between builds (a rebuild per frame must not pay allocations):
 * about 175 B per triangle on the default path (1.75 GB after a 10 M-triangle build).  A host that builds once and then only
 * traces returns it with this call (it waits for the device first); the next build allocates again.  ntr_ray_morton_sort keeps its
 * temporaries the same way (about 40 B per ray of the largest batch sorted so far); this call returns them too. */
NTR_API int ntr_lbvh_release_workspace(void);

/* reconstructKernel (src/rt/cuda/RendererKernels.cu:59-172; ReconstructInput, RendererKernels.hpp:46-70;
 * Renderer::updateResult, Renderer.cpp:583-659): hit records of one batch -> ABGR8 pixels.
 * rayType 0 = primary, 1 = AO, 2 = diffuse (textured / path-traced / VPL shading: out of scope). */
NTR_API int ntr_reconstruct(int32_t rayType, int32_t numRaysPerPrimary, int32_t firstPrimary, int32_t numPrimary,
                            const int32_t* d_primarySlotToID, const NtrRayResult* d_primaryResults,
                            const int32_t* d_batchIDToSlot, const NtrRayResult* d_batchResults,
                            const uint32_t* d_triMaterialColor, const uint32_t* d_triShadedColor,
                            uint32_t* d_pixels, void* stream);

/* RayBuffer::mortonSort (src/rt/ray/RayBuffer.cpp:103-165; kernels RayBufferKernels.cu:70-197): reorder a
 * batch by the 192-bit origin/direction Morton key.  The reference sorts the keys on the CPU; here the
 * whole pipeline runs on the device.  Ties keep their original slot order.  Out-of-place. Blocking. */
NTR_API int ntr_ray_morton_sort(int32_t numRays, const NtrRay* d_inRays, const int32_t* d_inSlotToID,
                                NtrRay* d_outRays, int32_t* d_outIDToSlot, int32_t* d_outSlotToID,
                                void* stream, float* seconds);

/* ---- multi-GPU (new design: the reference is single-device, its launches synchronous on one
 * context, src/framework/gpu/CudaKernel.cpp:188-221) ----------------------------------------------------------------
 * One process per GPU, or one host thread per GPU in one process.  Rays shard by screen tile: the primary-ray index
 * space [0, W*H) -- already 8 x 8 pixel blocks in Morton order (PixelTable, src/rt/ray/PixelTable.cpp:77-122) -- is
 * cut into `world` contiguous ranges of whole 64-ray blocks; a rank's AO / diffuse rays derive from its own primary
 * hits, so no ray ever crosses a rank boundary.  The BVH is built once and replicated by broadcast; the ONLY
 * collective of a frame is the final gather of hit records (16 B per primary ray) or of the RGBA8 framebuffer (4 B per
 * pixel: 8.3 MB at 1080p) to the root, over RCCL (grouped point-to-point sends over xGMI).  RCCL is bound at run time
 * (dlopen), only by the ntr_dist_* calls.  The group calls are collective: every rank makes the same call. */
typedef struct NtrDist NtrDist;
#define NTR_DIST_ID_BYTES 128
/* The rank-th of `world` contiguous, align-ray-aligned ranges of [0, numPrimary) (the first ranges one block longer
 * when the blocks do not divide), and the AO batches of a range as RayGen::batching cuts them (src/rt/ray/RayGen.cpp:
 * 582-602; at most maxBatchRays output rays each): first[b], count[b] are input slots.  No device, no RCCL. */
NTR_API int ntr_frame_shard(int32_t numPrimary, int32_t rank, int32_t world, int32_t align, int32_t* lo, int32_t* hi);
NTR_API int ntr_frame_ao_batches(int32_t lo, int32_t hi, int32_t samples, int32_t maxBatchRays, int32_t* first, int32_t* count,
                                 int32_t capacity, int32_t* numBatches);
/* Group set-up.  Processes: the root calls ntr_dist_unique_id and hands the 128 bytes to every rank by any means (file,
 * socket, MPI, environment); every rank then calls ntr_dist_init on ITS device (ntr_set_device first).  Threads of one
 * process: ntr_dist_init_all creates one group object per device (devices == NULL: 0 .. numDevices - 1); thread i uses
 * out[i] after ntr_set_device(devices[i]). */
NTR_API int ntr_dist_unique_id(char id[NTR_DIST_ID_BYTES]);
NTR_API int ntr_dist_init(const char id[NTR_DIST_ID_BYTES], int32_t rank, int32_t world, NtrDist** out);
NTR_API int ntr_dist_init_all(int32_t numDevices, const int32_t* devices, NtrDist** out /* numDevices entries */);
NTR_API int ntr_dist_info(const NtrDist* dist, int32_t* rank, int32_t* world);
NTR_API int ntr_dist_destroy(NtrDist* dist);
/* BVH replication: `bytes` of device memory from `root` to every rank (asynchronous on `stream`); _bvh does the three
 * BVHLayout_Compact buffers (every rank passes buffers of the root's sizes). */
NTR_API int ntr_dist_broadcast(NtrDist* dist, void* d_buf, int64_t bytes, int32_t root, void* stream);
NTR_API int ntr_dist_broadcast_bvh(NtrDist* dist, void* d_nodes, int64_t nodesBytes, void* d_triWoop, int64_t triWoopBytes,
                                   int32_t* d_triIndex, int64_t triIndexBytes, int32_t root, void* stream);
/* The frame's one collective.  _records: rank r's hit records of its range (hi_r - lo_r records, d_ownRecords) land at
 * d_fullRecords + lo_r on the root (numPrimary records; ignored elsewhere).  _pixels: rank r's pixels -- the tiles of
 * its range inside its own W*H framebuffer d_ownPixels, as ntr_reconstruct wrote them -- land in the root's framebuffer
 * d_fullPixels; d_slotToPixel is the PixelTable's index-to-pixel map (ntr_pixel_table), d_scratch numPrimary words on
 * every rank.  Asynchronous on `stream`.  As with any collective, every rank of the group makes the matching call: a rank that
 * returns an argument error has posted nothing and its peers wait for it (argument errors must be uniform across ranks; the ranges
 * are, by construction).  _records_cuts: the same gather for ranges the host cut itself (cuts[0] = 0 <= cuts[1] <= ... <=
 * cuts[world] = numPrimary, the same table on every rank: ranges of equal predicted cost instead of equal ray counts).
 * bench.py --gpus N gathers through these calls (torch.distributed only carries the 128-byte id and the barriers);
 * tests/test_dist_native_gpu.py runs them over two devices from two host threads wherever the box has two GPUs (on one-GPU boxes:
 * world size 1, and the N-rank flow at N = 2-3 with host-staged collectives, bench.py --dist-backend gloo --one-device). */
NTR_API int ntr_dist_gather_records(NtrDist* dist, const NtrRayResult* d_ownRecords, int32_t numPrimary, int32_t align,
                                    NtrRayResult* d_fullRecords, int32_t root, void* stream);
NTR_API int ntr_dist_gather_records_cuts(NtrDist* dist, const NtrRayResult* d_ownRecords, const int32_t* cuts /* world + 1 */,
                                         NtrRayResult* d_fullRecords, int32_t root, void* stream);
NTR_API int ntr_dist_gather_pixels(NtrDist* dist, const uint32_t* d_ownPixels, const int32_t* d_slotToPixel, int32_t numPrimary,
                                   int32_t align, uint32_t* d_fullPixels, uint32_t* d_scratch, int32_t root, void* stream);

/* ---- host-side BVH production (no device work) ---------------------------- */

/* Host SAH build + Compact flatten: `BVH bvh(scene, platform, params);
 * CudaBVH(bvh, BVHLayout_Compact)` of Renderer::getCudaBVH
 * (src/rt/cuda/Renderer.cpp:282-285; SAHBVHBuilder.cpp:51-254; CudaBVH.cpp:579-687)
 * with Platform("GPU") and setLeafPreferences(minLeaf,maxLeaf) (Renderer.cpp:88-89
 * uses 1,1).  Returns an opaque handle holding host copies of the three buffers. */
typedef struct NtrHostBvh NtrHostBvh;
typedef struct NtrHostBvhInfo {
    const void*    nodes;     int64_t nodesBytes;
    const void*    triWoop;   int64_t triWoopBytes;
    const int32_t* triIndex;  int64_t triIndexBytes;
    int32_t layout;
    int32_t numInnerNodes, numLeafNodes, maxDepth;
    float   buildSeconds;
} NtrHostBvhInfo;
NTR_API int  ntr_sah_build(int32_t numTris, const int32_t* triVtxIndex /* 3 per tri */,
                           int32_t numVerts, const float* vtxPos /* 3 per vertex */,
                           int32_t minLeafSize, int32_t maxLeafSize, NtrHostBvh** out);
NTR_API int  ntr_host_bvh_info(const NtrHostBvh* bvh, NtrHostBvhInfo* info);
NTR_API void ntr_host_bvh_free(NtrHostBvh* bvh);

/* A host BVH object over copies of existing BVHLayout_Compact buffers (e.g. the three buffers of a bvhcache file,
 * CudaBVH::CudaBVH(InputStream&), src/rt/cuda/CudaBVH.cpp:105-116, or a device-built LBVH downloaded to the host). */
NTR_API int ntr_host_bvh_wrap(const void* nodes, int64_t nodesBytes, const void* triWoop, int64_t triWoopBytes,
                              const int32_t* triIndex, int64_t triIndexBytes, NtrHostBvh** out);

/* CudaAS::trace(RayBuffer&, Buffer& visibility) -- the reference's HOST tracer (src/rt/cuda/CudaAS.hpp:62,
 * CudaBVH::trace, src/rt/cuda/CudaBVH.cpp:213-302; used by its CPURenderer and by BASELINE configuration 1): rays
 * and results are HOST arrays, visibility (may be NULL / 0) receives 1 at the id of every triangle hit, stats (may
 * be NULL) the reference's RayStats counters (numLeafVisits / numHits stay 0: the reference does not count them).
 * Single-threaded, like the reference.  This is an API of its own for host-side callers, NOT a fallback of
 * ntr_trace_bvh, which never runs on the CPU and fails without a HIP device. */
NTR_API int ntr_host_bvh_trace(const NtrHostBvh* bvh, int32_t numRays, int32_t anyHit, const NtrRay* rays,
                               NtrRayResult* results, int32_t* visibility, int32_t numVisibility, NtrTraceStats* stats);

/* ---- scene ingest (host only) ----------------------------------------------- */

/* CameraControls::decodeSignature / encodeSignature (src/framework/3d/CameraControls.cpp:342-399, 471-545):
 * out = position[3], forward[3], up[3], speed, fov, near, far, keepAligned. */
NTR_API int ntr_camera_decode(const char* signature, float out[16]);
NTR_API int ntr_camera_reencode(const char* signature, char* out, int32_t outSize);
/* invert(fitToView(-1, 2) * perspective * worldToCamera) of Renderer::beginFrame (Renderer.cpp:473-477),
 * row-major, ready for ntr_raygen_primary. */
NTR_API int ntr_camera_nscreen_to_world(const char* signature, int32_t viewW, int32_t viewH, float matrix[16],
                                        float position[3], float* cameraFar);

/* Wavefront OBJ import with the reference's vertex / triangle numbering
 * (src/framework/io/MeshWavefrontIO.cpp:412-485, src/rt/Scene.cpp:101-136). */
typedef struct NtrObjMesh NtrObjMesh;
NTR_API int  ntr_obj_load(const char* path, NtrObjMesh** out, int32_t* numTris, int32_t* numVerts, int32_t* numSubmeshes);
NTR_API int  ntr_obj_get(const NtrObjMesh* mesh, int32_t* triVtxIndex /* 3/tri */, float* vtxPos /* 3/vertex */);
NTR_API void ntr_obj_free(NtrObjMesh* mesh);

#ifdef __cplusplus
}
#endif
#endif
