// ntr_api.cpp -- C-ABI of libntrace_amd.so (declared in include/ntrace_amd.h).
//
// Thin layer: argument checks mirroring the reference's host-side checks, HIP
// event timing mirroring CudaKernel::launchTimed (src/framework/gpu/CudaKernel.cpp:188-221),
// and kernel launches.  There is deliberately no CPU path here: with no HIP device
// every compute entry point fails (NTR_ERR_NO_DEVICE / NTR_ERR_HIP).

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "ntrace_amd.h"
#include "ntr_internal.h"
#include "trace_kernels.h"
#include "trace_plan.h"

namespace {

thread_local char g_err[512] = "";

}  // namespace

namespace ntr {

int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what)
{
    int code = (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
                   ? NTR_ERR_NO_DEVICE : NTR_ERR_HIP;
    return set_error(code, "%s: %s", what, hipGetErrorString(e));
}

// Per-device workspace: ring of pool counters + sticky status word.
struct DeviceState {
    bool init = false;
    int32_t* counters = nullptr;  // kNumCounters ints
    unsigned int* status = nullptr;
    unsigned long long* stats = nullptr;  // 4 counters of the stats variant
    int next = 0;
    int nextPinned = 0;
    int numCUs = 0;
};
static constexpr int kNumCounters = 64;  // ring of counter sets (kPoolHeadsMax heads x 64 B each)
static constexpr int kPinnedCounters = 192;  // counter sets handed to launches captured into HIP graphs: never reused
static constexpr int kMaxDevices = 64;
static constexpr int64_t kMaxNodesBytes = 0x76543200ll;  // largest multiple of 64 below the sentinel 0x76543210
static DeviceState g_dev[kMaxDevices];
static std::mutex g_mu;
static std::mutex g_statusMu;

bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

int get_device_state(DeviceState** out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (dev < 0 || dev >= kMaxDevices) return set_error(NTR_ERR_INVALID, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceState& s = g_dev[dev];
    if (!s.init) {
        hipDeviceProp_t prop;
        NTR_HIP(hipGetDeviceProperties(&prop, dev));
        s.numCUs = prop.multiProcessorCount;
        // Counters sit 64 B apart so concurrent launches never share a line.
        NTR_HIP(hipMalloc((void**)&s.counters, (kNumCounters + kPinnedCounters) * kPoolHeadsMax * 64));
        NTR_HIP(hipMalloc((void**)&s.status, 64));
        NTR_HIP(hipMalloc((void**)&s.stats, 256));
        NTR_HIP(hipMemset(s.status, 0, 64));
        s.init = true;
    }
    *out = &s;
    return NTR_OK;
}

struct KernelInfo {
    const char* name;
    int variant;
    NtrKernelConfig cfg;
};

// Reference kernel selectors (file names under src/rt/kernels/) -> CDNA4 variants.
// All CDNA4 variants consume BVHLayout_Compact (the reference fork asserts Compact
// for every BVH it builds, src/rt/cuda/CudaBVH.cpp:65-82).
static const KernelInfo kKernels[] = {
    {"fermi_speculative_while_while", NTR_VARIANT_PERRAY,
     {NTR_BVHLayout_Compact, 64, 1, 0}},   // the per-ray kernel is launched in 64-thread workgroups (one wave)
    {"tesla_persistent_while_while", NTR_VARIANT_PERSISTENT,
     {NTR_BVHLayout_Compact, 64, NTR_TRACE_WAVES_PER_BLOCK, 1}},
    {"tesla_persistent_speculative_while_while", NTR_VARIANT_PERSISTENT,
     {NTR_BVHLayout_Compact, 64, NTR_TRACE_WAVES_PER_BLOCK, 1}},
    {"kepler_dynamic_fetch", NTR_VARIANT_PERSISTENT,
     {NTR_BVHLayout_Compact, 64, NTR_TRACE_WAVES_PER_BLOCK, 1}},
};

static const KernelInfo* find_kernel(const char* name)
{
    if (!name) return nullptr;
    for (const KernelInfo& k : kKernels)
        if (strcmp(k.name, name) == 0) return &k;
    return nullptr;
}

}  // namespace ntr

// Tunables.  Environment overrides exist for benchmarking sweeps; they are read ONCE, when the library is first
// used (and again on ntr_tunables_reload(), which the sweep scripts call after changing a variable), never per
// launch.  No pointer is ever taken from the environment.
static int env_int(const char* name, int def)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : def;
}

namespace ntr {
static Tunables g_tun;
static bool g_tunLoaded = false;
static std::mutex g_tunMu;

static void tunables_load_locked()
{
    Tunables t;
    t.chunk = env_int("NTR_TRACE_CHUNK", 64);
    t.fetchThreshold = env_int("NTR_TRACE_FETCH_THRESHOLD", -1);  // -1: 24 for kepler_dynamic_fetch, 0 otherwise
    t.leafSwitchBelow = env_int("NTR_TRACE_LEAF_SWITCH", -1);     // -1: 32 for closest-hit, 24 for any-hit launches (bench-protocol sweep, scripts/jobs/gpu_job_r02ls.sh)
    t.blocksPerCU = env_int("NTR_TRACE_BLOCKS_PER_CU", 7);          // persistent kernels: 7 x 4 waves per CU -- what 69 VGPRs let be resident (a workgroup that is not resident at launch holds its statically assigned first chunks back until another one leaves)
    t.blocksPerCUIncoherent = env_int("NTR_TRACE_BLOCKS_PER_CU_INCOHERENT", 3);   // persistent kernels, batches the device finds incoherent (scattered origins): fewer rays in flight = less queueing per step (scripts/studies/inflight_sweep.py)
    t.blocksPerCUDivergent = env_int("NTR_TRACE_BLOCKS_PER_CU_DIVERGENT", 4);     // ... batches whose rays start together and wander apart (a diffuse batch)
    t.octant = env_int("NTR_TRACE_OCTANT", 1);
    t.closestWaves = env_int("NTR_TRACE_CLOSEST_WAVES", 1);      // likewise for closest-hit launches: primary +2.1 % with 1
    t.anyHitWaves = env_int("NTR_TRACE_ANYHIT_WAVES", 1);        // waves per workgroup of plain any-hit launches of the per-ray kernel (1, 2, 4): AO +1.7 % with 1
    t.flatFetch = env_int("NTR_TRACE_FLAT_FETCH", 1);             // unified-step loop: one group of global loads per iteration (0 = two masked groups of range-checked buffer loads)
    t.uniformPrologue = env_int("NTR_TRACE_UNIFORM_PROLOGUE", 1);  // per-ray kernels: scalar node fetches while the lanes of a fresh wave all hold the same inner node
    t.splitSlice = env_int("NTR_TRACE_SPLIT_SLICE", 8);   // persistent kernels, unified-step loop: once the pool is dry, lanes without a ray take over stack entries of the wave's live rays; looked at every N steps (0 = off)
    t.wholeWave = env_int("NTR_TRACE_WHOLE_WAVE", 1);      // kepler_dynamic_fetch: waves start in whole-wave mode and switch to single-lane refills per wave (0 = dynamic fetch from the start, as until round 5)
    t.prefetchAfter = env_int("NTR_TRACE_PREFETCH_AFTER", 8);   // persistent kernels: iterations into a chunk after which a wave posts the dequeue of its next one (-1 = never)
    t.minipool = env_int("NTR_TRACE_MINIPOOL", -1);              // closest-hit per-ray launches: rays owned by a wave / 64.  -1: decided per batch on the device (1, or minipoolWide when the prediction finds the batch incoherent); 0: the plain per-ray kernel; 1 ... 16: forced
    t.minipoolWide = env_int("NTR_TRACE_MINIPOOL_WIDE", -1);     // K of an incoherent batch: 2 / 4, or -1 = by tree size (4 from 32 MB of nodes up)
    t.minipoolThreshold = env_int("NTR_TRACE_MINIPOOL_THRESHOLD", 48);   // refill a wave's finished lanes when fewer than this many are live
    t.unified = env_int("NTR_TRACE_UNIFIED", 1);                  // kepler_dynamic_fetch: unified-step loop (0 = while-while loop + dynamic fetch)
    t.perrayUnified = env_int("NTR_TRACE_PERRAY_UNIFIED", 1);     // per-ray kernel with the unified-step loop: 1 = always (the default since the one-correction divide: AO batches on one-triangle-leaf trees -4 %), 0 = never, -1 = closest-hit launches always, any-hit launches only on trees flagged NTR_BVH_WIDE_LEAVES (the rule of round 3)
    t.poolHeads = env_int("NTR_TRACE_POOL_HEADS", 128);           // persistent kernels: 8..1024, a multiple of 8 (sweep: scripts/studies/persist_diag.py)
    t.autoHint = env_int("NTR_TRACE_AUTO_HINT", 1);               // dispatch order learned from the previous launch of the same batch (stream, rays, count, BVH)
    t.autoHintMinRays = env_int("NTR_TRACE_AUTO_HINT_MIN_RAYS", 1 << 17);
    t.route = env_int("NTR_TRACE_ROUTE", 1);   // batches are traced by the body that is fast on them, whatever the kernel name (trace_plan.h ROUTING); 0 = the named body always
    t.persistentHints = env_int("NTR_TRACE_PERSISTENT_HINTS", 1);   // the persistent kernels honour scheduling hints too (pool handed out in the hint's order, chunk lives recorded as block costs)
    t.predict = env_int("NTR_TRACE_PREDICT", 1);
    t.predictPersistent = env_int("NTR_TRACE_PREDICT_PERSISTENT", 1);   // persistent kernels: pool in predicted-cost order (closest-hit launches of >= predictMinRays)
    t.predictDepth = env_int("NTR_TRACE_PREDICT_DEPTH", 9);
    t.predictMinRays = env_int("NTR_TRACE_PREDICT_MIN_RAYS", 1 << 20);
    t.predictMinNodes = env_int("NTR_TRACE_PREDICT_MIN_NODES", 4096);
    t.schedRefreshEvery = env_int("NTR_SCHED_REFRESH_EVERY", 16);
    t.schedClasses = env_int("NTR_SCHED_CLASSES", 32);
    t.lbvhSplit = env_int("NTR_LBVH_SPLIT", 3072);
    t.lbvhSubThreads = env_int("NTR_LBVH_SUB_THREADS", 128);
    t.lbvhAggLds = env_int("NTR_LBVH_AGG_LDS", 1);     // bottom-up emit: meetings inside a tile through LDS
    t.lbvhMortonKeys = env_int("NTR_LBVH_MORTON_KEYS", 0);  // triangles per thread of the Morton / histogram kernel (0 = 4)
    t.lbvhMortonThreads = env_int("NTR_LBVH_MORTON_THREADS", 0);
    t.lbvhMarkThreads = env_int("NTR_LBVH_MARK_THREADS", 0);      // workgroup size of the leaf-mark kernel (256 / 1024; 0 = by size)
    t.lbvhSortItems = env_int("NTR_LBVH_SORT_ITEMS", 0);   // keys per thread of a one-sweep tile (8 / 16 / 24 / 32; 0 = by size)
    t.lbvhAggStaged = env_int("NTR_LBVH_AGG_STAGED", -1);  // bottom-up emit in two launches: -1 = from 2^20 triangles, 0 / 1 = never / always
    if (t.chunk < 1) t.chunk = 1;
    g_tun = t;
    g_tunLoaded = true;
}

Tunables tunables()
{
    std::lock_guard<std::mutex> lk(g_tunMu);
    if (!g_tunLoaded) tunables_load_locked();
    return g_tun;
}
}  // namespace ntr

extern "C" int ntr_tunables_reload(void)
{
    std::lock_guard<std::mutex> lk(ntr::g_tunMu);
    ntr::tunables_load_locked();
    return NTR_OK;
}


using namespace ntr;

extern "C" {

const char* ntr_last_error(void) { return g_err; }
int ntr_version(void) { return 100; }

int ntr_device_count(int* count)
{
    if (!count) return set_error(NTR_ERR_INVALID, "ntr_device_count: null argument");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount"); }
    return NTR_OK;
}

int ntr_set_device(int device) { NTR_HIP(hipSetDevice(device)); return NTR_OK; }

int ntr_malloc(void** d_ptr, size_t bytes)
{
    if (!d_ptr) return set_error(NTR_ERR_INVALID, "ntr_malloc: null argument");
    *d_ptr = nullptr;
    if (bytes == 0) return NTR_OK;
    NTR_HIP(hipMalloc(d_ptr, bytes));
    return NTR_OK;
}

int ntr_free(void* d_ptr)
{
    if (d_ptr) NTR_HIP(hipFree(d_ptr));
    return NTR_OK;
}

int ntr_memcpy_h2d(void* d, const void* h, size_t n, void* stream)
{
    if (n == 0) return NTR_OK;
    NTR_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)stream));
    NTR_HIP(hipStreamSynchronize((hipStream_t)stream));
    return NTR_OK;
}

int ntr_memcpy_d2h(void* h, const void* d, size_t n, void* stream)
{
    if (n == 0) return NTR_OK;
    NTR_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)stream));
    NTR_HIP(hipStreamSynchronize((hipStream_t)stream));
    return NTR_OK;
}

int ntr_memcpy_d2d(void* dst, const void* src, size_t n, void* stream)
{
    if (n == 0) return NTR_OK;
    NTR_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return NTR_OK;
}

int ntr_memset(void* d, int value, size_t n, void* stream)
{
    if (n == 0) return NTR_OK;
    NTR_HIP(hipMemsetAsync(d, value, n, (hipStream_t)stream));
    return NTR_OK;
}

int ntr_stream_synchronize(void* stream)
{
    NTR_HIP(hipStreamSynchronize((hipStream_t)stream));
    return NTR_OK;
}

int ntr_query_config(const char* kernelName, NtrKernelConfig* config)
{
    if (!config) return set_error(NTR_ERR_INVALID, "ntr_query_config: null config");
    const KernelInfo* k = find_kernel(kernelName);
    if (!k) {
        // CudaBVHTracer::setKernel leaves bvhLayout = BVHLayout_Max when queryConfig
        // does not run (CudaBVHTracer.cpp:66-71).
        config->bvhLayout = NTR_BVHLayout_Max;
        config->blockWidth = config->blockHeight = config->usePersistentThreads = 0;
        return set_error(NTR_ERR_UNKNOWN_KERNEL, "unknown kernel '%s'", kernelName ? kernelName : "(null)");
    }
    *config = k->cfg;
    return NTR_OK;
}

// ---- dispatch-order prediction (sched_kernels.hip) ---------------------------------------------------
// Top-of-tree box tables, one per node buffer seen (keyed by pointer and size; rebuilt by
// ntr_bvh_validate, which hosts call after every (re)build).  A stale table only costs scheduling quality.
struct TopTable {
    const void* nodes = nullptr;
    int64_t bytes = 0;
    int device = -1;
    void* table = nullptr;           // 2 float4 per box
    unsigned int* count = nullptr;   // boxes in the table
    unsigned long long lastUse = 0;
    bool pinned = false;             // referenced by a captured HIP graph: never evicted
};
static constexpr int kTopTables = 16;
static constexpr size_t kTopTableBytes = (((size_t)2 << NTR_TOP_DEPTH_MAX) + 16) * 32;  // + padding read by predict_kernel's batches

// Class counters / lists / block order of one prediction.  A live (not captured) launch uses the entry owned by its stream
// (launches on one stream are ordered; two streams must not share an entry).  A launch that is being captured into a HIP graph
// gets an entry of its OWN, taken from spares that live launches provision (nothing may be allocated during a capture): a graph
// replayed on whatever stream then never shares order[] with a live launch or with another captured launch.  Pinned entries are
// returned to the spares by ntr_trace_graph_release_all().
struct PredictScratch {
    enum State { FREE = 0, LIVE, SPARE, PINNED };
    State state = FREE;
    void* stream = nullptr;              // LIVE: the owning stream (the null stream is a stream like any other)
    int device = -1;
    unsigned int* classCount = nullptr;  // NTR_SCHED_PRED_WORDS words: class counters + incoherent-block counter (zero whenever no prediction is
                                         // in flight) + the mini-pool K of the last prediction
    unsigned int* classList = nullptr;
    unsigned int* order = nullptr;
    int capBlocks = 0;
    unsigned long long lastUse = 0;
};
static constexpr int kScratch = 48;
static constexpr int kScratchSpares = 4;   // spares a live launch keeps ready (per device, sized for the largest launch seen) for captured launches
static constexpr int kScratchLive = 16;    // streams with an entry of their own before the least recently used one is recycled


// Scheduling hint (include/ntrace_amd.h): per-block cost of the previous launch -> block order of the next.
struct NtrSchedHint {
    unsigned int* order = nullptr;  // device, numBlocks entries
    unsigned int* cost = nullptr;   // device, numBlocks entries
    int numBlocks = 0;              // 0 = unbound
    int capBlocks = 0;              // blocks the arrays have room for (caller-owned hints grow, never shrink: a frame's short last batch
                                    // must not cost a hipFree -- a device-wide synchronisation -- and two hipMallocs every frame)
    int device = -1;
    int uses = 0;                   // launches since the hint was (re)bound
    bool valid = false;             // order[] holds a permutation
    bool predicted = false;         // order[] comes from ntr_sched_hint_predict and has not been used yet
};

// Automatic scheduling feedback.  The launch time of the per-ray kernel is set by where its long-lived blocks start;
// what a launch MEASURED about its blocks is the best order for the next launch of the same batch.  The reference's benchmark traces
// every batch 1 + warm-up + measure times (App.cpp:955-958), a renderer with a parked or slowly moving camera regenerates nearly the
// same rays into the same buffers frame after frame -- so the library keeps, per (stream, ray buffer, ray count, ray kind, BVH), the
// scheduling hint a caller could have kept by hand (NtrSchedHint) and uses it without being asked.  Only the dispatch ORDER changes;
// a stale entry (new rays at the old address) is merely a worse order and adapts within a launch or two.  Entries are per stream
// (launches on one stream are ordered, so order[] is never rewritten under a launch that reads it); captured launches do not use them.
struct AutoHint {
    const void* rays = nullptr;
    const void* nodes = nullptr;
    void* stream = nullptr;
    int numRays = 0, anyHit = 0, device = -1;
    bool used = false;
    int sightings = 0;              // launches of this key: storage is allocated at the second one (a batch seen once pays nothing)
    unsigned long long lastUse = 0;
    NtrSchedHint hint;              // (its arrays come from the stream-ordered allocator: hipMallocAsync / hipFreeAsync on `stream`)
};
static constexpr int kAutoHints = 96;

// Scheduling state of ONE device: top-of-tree tables, prediction scratch, automatic hints (and, in the A/B build, hand-off queues), under
// the device's own mutex.  Until round 4 these were process-global tables behind one mutex: with one host thread per GPU
// (ntr_dist_init_all) eight devices x 17 batches are 136 automatic-hint keys for 96 entries, so the devices that came
// last silently ran in buffer order -- and MAX over ranks is the metric -- while every launch of every device took the same lock.
struct SchedState {
    std::mutex mu;
    unsigned long long clock = 0;   // LRU stamps
    TopTable top[kTopTables];
    PredictScratch scratch[kScratch];
    AutoHint autoHints[kAutoHints];
};
static std::atomic<SchedState*> g_sched[kMaxDevices];   // created on a device's first use, never destroyed (entries hold device memory of a live context)

static int sched_state(int dev, SchedState** out)
{
    if (dev < 0 || dev >= kMaxDevices) return set_error(NTR_ERR_INVALID, "device index %d out of range", dev);
    // the launch path only reads the pointer (acquire): the host threads of a thread-per-GPU driver share no lock here; the process-wide
    // mutex is taken once per device, to create its state
    SchedState* ss = g_sched[dev].load(std::memory_order_acquire);
    if (!ss) {
        std::lock_guard<std::mutex> lk(g_mu);
        ss = g_sched[dev].load(std::memory_order_relaxed);
        if (!ss) {
            ss = new (std::nothrow) SchedState();
            if (!ss) return set_error(NTR_ERR_NOMEM, "out of host memory (scheduling state of device %d)", dev);
            g_sched[dev].store(ss, std::memory_order_release);
        }
    }
    *out = ss;
    return NTR_OK;
}
static int sched_state_current(SchedState** out, int* devOut = nullptr)
{
    int dev = 0;
    NTR_HIP(hipGetDevice(&dev));
    if (devOut) *devOut = dev;
    return sched_state(dev, out);
}


static int top_table_get(const void* d_nodes, int64_t nodesBytes, hipStream_t s, bool rebuild, TopTable** out)
{
    SchedState* ss = nullptr;
    int dev = 0;
    const int src = sched_state_current(&ss, &dev);
    if (src != NTR_OK) return src;
    std::lock_guard<std::mutex> lk(ss->mu);
    const bool capturing = stream_is_capturing(s);
    TopTable* t = nullptr;
    TopTable* lru = nullptr;
    for (auto& e : ss->top) {
        if (e.nodes == d_nodes && e.bytes == nodesBytes && e.device == dev) { t = &e; break; }
        if (!e.pinned && (!lru || e.lastUse < lru->lastUse)) lru = &e;
    }
    bool build = rebuild;
    if (!t) {
        if (!lru) return set_error(NTR_ERR_NOMEM, "ntr_trace_bvh: every top-of-tree table is held by a captured HIP graph");
        if (capturing) return set_error(NTR_ERR_INVALID, "ntr_trace_bvh: first launch on a BVH cannot be captured (trace it once, or call ntr_bvh_validate, before capturing)");
        t = lru;
        if (t->table && t->device != dev) { (void)hipFree(t->table); (void)hipFree(t->count); t->table = nullptr; t->count = nullptr; }
        if (!t->table) {
            NTR_HIP(hipMalloc(&t->table, kTopTableBytes));
            NTR_HIP(hipMalloc((void**)&t->count, sizeof(unsigned int)));
        }
        t->nodes = d_nodes; t->bytes = nodesBytes; t->device = dev;
        build = true;
    }
    t->lastUse = ++ss->clock;
    if (capturing) t->pinned = true;
    if (build) {
        const hipError_t e = ntr_launch_top_table(d_nodes, (unsigned int)nodesBytes, tunables().predictDepth, t->table, t->count, s);
        if (e != hipSuccess) return hip_fail(e, "top_table launch");
    }
    *out = t;
    return NTR_OK;
}

extern "C" int ntr_top_table_refresh(const void* d_nodes, int64_t nodesBytes, void* stream)
{
    TopTable* t = nullptr;
    return top_table_get(d_nodes, nodesBytes, (hipStream_t)stream, true, &t);
}

static int scratch_alloc(PredictScratch* p, int dev, int numBlocks)
{
    if (!p->classCount) {
        NTR_HIP(hipMalloc((void**)&p->classCount, NTR_SCHED_PRED_WORDS * sizeof(unsigned int)));
        NTR_HIP(hipMemset(p->classCount, 0, NTR_SCHED_PRED_WORDS * sizeof(unsigned int)));
    }
    if (p->capBlocks < numBlocks) {
        if (p->classList) { (void)hipFree(p->classList); (void)hipFree(p->order); p->classList = p->order = nullptr; p->capBlocks = 0; }
        NTR_HIP(hipMalloc((void**)&p->classList, (size_t)NTR_SCHED_PRED_CLASSES * numBlocks * sizeof(unsigned int)));
        NTR_HIP(hipMalloc((void**)&p->order, (size_t)numBlocks * sizeof(unsigned int)));
        p->capBlocks = numBlocks;
    }
    p->device = dev;
    return NTR_OK;
}

static void scratch_free(PredictScratch* p)
{
    if (p->classCount) (void)hipFree(p->classCount);
    if (p->classList) (void)hipFree(p->classList);
    if (p->order) (void)hipFree(p->order);
    *p = PredictScratch();
}

// spares of at least numBlocks on the state's device (its mutex held)
static int scratch_count_spares(SchedState* ss, int dev, int numBlocks)
{
    int n = 0;
    for (auto& e : ss->scratch)
        if (e.state == PredictScratch::SPARE && e.device == dev && e.capBlocks >= numBlocks) n++;
    return n;
}

// makes sure `want` spares of at least numBlocks exist on the state's device (its mutex held, not capturing): smaller spares are regrown first (nothing
// references a spare), then free slots are taken; running out of slots is not an error here
static int scratch_provision_spares(SchedState* ss, int dev, int numBlocks, int want)
{
    int have = scratch_count_spares(ss, dev, numBlocks);
    for (auto& e : ss->scratch) {
        if (have >= want) break;
        if (e.state == PredictScratch::SPARE && e.device == dev && e.capBlocks < numBlocks) {
            const int rc = scratch_alloc(&e, dev, numBlocks);
            if (rc != NTR_OK) return rc;
            have++;
        }
    }
    for (auto& e : ss->scratch) {
        if (have >= want) break;
        if (e.state != PredictScratch::FREE) continue;
        const int rc = scratch_alloc(&e, dev, numBlocks);
        if (rc != NTR_OK) return rc;
        e.state = PredictScratch::SPARE;
        have++;
    }
    return NTR_OK;
}

static int predict_scratch_get(hipStream_t s, int numBlocks, PredictScratch** out)
{
    SchedState* ss = nullptr;
    int dev = 0;
    const int src = sched_state_current(&ss, &dev);
    if (src != NTR_OK) return src;
    std::lock_guard<std::mutex> lk(ss->mu);
    const bool capturing = stream_is_capturing(s);
    if (capturing) {   // a private entry from the spares: the smallest that fits
        PredictScratch* best = nullptr;
        for (auto& e : ss->scratch)
            if (e.state == PredictScratch::SPARE && e.device == dev && e.capBlocks >= numBlocks && (!best || e.capBlocks < best->capBlocks)) best = &e;
        // no spare of this size: the launch is captured without a predicted order (an optimisation, not a contract);
        // ntr_trace_graph_reserve provisions spares for graphs that want it
        if (!best) { *out = nullptr; return NTR_OK; }
        best->state = PredictScratch::PINNED;
        best->lastUse = ++ss->clock;
        *out = best;
        return NTR_OK;
    }
    PredictScratch* p = nullptr;
    PredictScratch* lru = nullptr;
    PredictScratch* empty = nullptr;
    int live = 0;
    for (auto& e : ss->scratch) {
        if (e.state == PredictScratch::LIVE) {
            live++;
            if (e.stream == (void*)s && e.device == dev) { p = &e; break; }
            if (!lru || e.lastUse < lru->lastUse) lru = &e;
        } else if (e.state == PredictScratch::FREE && !empty) {
            empty = &e;
        }
    }
    if (!p) {
        p = (empty && live < kScratchLive) ? empty : (lru ? lru : empty);
        if (!p) return set_error(NTR_ERR_NOMEM, "ntr_trace_bvh: every prediction scratch is held by a captured HIP graph (ntr_trace_graph_release_all)");
        if (p->state == PredictScratch::LIVE) {
            NTR_HIP(hipDeviceSynchronize());  // the evicted stream's launches may still read it
            scratch_free(p);
        }
    } else if (p->capBlocks < numBlocks && p->classList) {
        NTR_HIP(hipStreamSynchronize(s));
    }
    int rc = scratch_alloc(p, dev, numBlocks);
    if (rc != NTR_OK) return rc;
    p->state = PredictScratch::LIVE;
    p->stream = (void*)s;
    p->lastUse = ++ss->clock;
    rc = scratch_provision_spares(ss, dev, numBlocks, kScratchSpares);
    if (rc != NTR_OK) return rc;
    *out = p;
    return NTR_OK;
}

static void sched_hint_release(NtrSchedHint* h)
{
    if (h->order) (void)hipFree(h->order);
    if (h->cost) (void)hipFree(h->cost);
    h->order = h->cost = nullptr;
    h->numBlocks = 0; h->capBlocks = 0; h->uses = 0; h->valid = false; h->predicted = false;
}

// Binds a caller-owned hint to a batch of numBlocks blocks on `dev`: the arrays are reallocated only to grow (or on another device); a
// hint bound to a different block count starts over.
static int sched_hint_bind(NtrSchedHint* h, int numBlocks, int dev)
{
    if (h->device != dev || h->capBlocks < numBlocks || !h->order) {
        sched_hint_release(h);
        // order[numBlocks .. numBlocks + 2]: the batch's coherence words, the last one its mini-pool K (0 = not estimated yet, read as 1):
        // written by the dispatch-order prediction of the batch's first launch, or by the coherence probe of its refresh launches
        NTR_HIP(hipMalloc((void**)&h->order, ((size_t)numBlocks + 3) * sizeof(unsigned int)));
        NTR_HIP(hipMalloc((void**)&h->cost, (size_t)numBlocks * sizeof(unsigned int)));
        h->capBlocks = numBlocks;
        h->device = dev;
    }
    if (h->numBlocks != numBlocks) {
        h->numBlocks = numBlocks;
        h->uses = 0; h->valid = false; h->predicted = false;
    }
    return NTR_OK;
}

// returns an automatic hint's arrays in stream order: behind every launch of `s` that reads them (no synchronisation, no event)
static void auto_hint_release_async(NtrSchedHint* h, hipStream_t s)
{
    if (h->order) (void)hipFreeAsync(h->order, s);
    if (h->cost) (void)hipFreeAsync(h->cost, s);
    h->order = h->cost = nullptr;
    h->numBlocks = 0; h->uses = 0; h->valid = false; h->predicted = false;
}

// The hint of this batch, or null: the first launch of a key only registers it (no allocation, no hint); from the second on the key owns a
// hint.  Nothing here synchronises, records an event or frees under a launch in flight: an entry is recycled only if it holds no storage
// or belongs to THIS stream (its arrays are then freed in stream order, behind the launches that read them), and if no entry can be had
// the launch simply goes without (buffer / predicted order).
// The table logic alone (no HIP call: ntr_selftest_auto_hint_table drives it on the CPU tier).  Returns the entry of the key on a repeated
// sighting; on a first sighting registers the key and returns null, *evicted then being an entry whose arrays the caller must return in
// the order of `stream` (or null).  *noEntry: the table had no entry this launch may take.
static AutoHint* auto_hint_lookup(SchedState* ss, const void* d_rays, const void* d_nodes, int numRays, int anyHit, void* stream, int dev,
                                  NtrSchedHint** evicted, bool* noEntry)
{
    *evicted = nullptr;
    *noEntry = false;
    AutoHint* hit = nullptr;
    AutoHint* freeE = nullptr;
    for (auto& e : ss->autoHints) {
        if (e.used && e.rays == d_rays && e.nodes == d_nodes && e.numRays == numRays && e.anyHit == anyHit && e.stream == stream) { hit = &e; break; }
        if (!e.used && !freeE) freeE = &e;
    }
    if (hit) {
        hit->lastUse = ++ss->clock;
        hit->sightings++;
        return hit;
    }
    AutoHint* v = freeE;
    if (!v) {   // least recently used among the entries this launch may recycle: those without storage, and this stream's own
        for (auto& e : ss->autoHints) {
            if (e.hint.order && e.stream != stream) continue;
            if (!v || e.lastUse < v->lastUse) v = &e;
        }
        if (!v) { *noEntry = true; return nullptr; }
        if (v->hint.order) *evicted = &v->hint;
    }
    v->rays = d_rays; v->nodes = d_nodes; v->numRays = numRays; v->anyHit = anyHit; v->stream = stream; v->device = dev;
    v->used = true;
    v->sightings = 1;
    v->lastUse = ++ss->clock;
    v->hint.uses = 0; v->hint.valid = false;
    return nullptr;   // first sighting: registered, not hinted
}

static int auto_hint_get(const void* d_rays, const void* d_nodes, int numRays, int anyHit, hipStream_t s, int numBlocks, NtrSchedHint** out)
{
    *out = nullptr;
    SchedState* ss = nullptr;
    int dev = 0;
    const int src = sched_state_current(&ss, &dev);
    if (src != NTR_OK) return src;
    std::lock_guard<std::mutex> lk(ss->mu);
    NtrSchedHint* evicted = nullptr;
    bool noEntry = false;
    AutoHint* hit = auto_hint_lookup(ss, d_rays, d_nodes, numRays, anyHit, (void*)s, dev, &evicted, &noEntry);
    if (evicted) auto_hint_release_async(evicted, s);
    if (!hit) return NTR_OK;
    NtrSchedHint* h = &hit->hint;
    if (h->numBlocks != numBlocks || h->device != dev) {   // second sighting: the storage
        if (h->order) auto_hint_release_async(h, s);
        unsigned int* order = nullptr;
        unsigned int* cost = nullptr;
        // (failing here -- another thread capturing in global mode, memory -- only means: no hint for this launch)
        if (hipMallocAsync((void**)&order, ((size_t)numBlocks + 3) * sizeof(unsigned int), s) != hipSuccess) { (void)hipGetLastError(); return NTR_OK; }
        if (hipMallocAsync((void**)&cost, (size_t)numBlocks * sizeof(unsigned int), s) != hipSuccess) { (void)hipGetLastError(); (void)hipFreeAsync(order, s); return NTR_OK; }
        h->order = order; h->cost = cost;
        h->numBlocks = numBlocks;
        h->device = dev;
        h->uses = 0; h->valid = false;
    }
    *out = h;
    return NTR_OK;
}

static int trace_impl(const char* kernelName, int32_t numRays, int32_t anyHit, const NtrRay* d_rays,
                      NtrRayResult* d_results, const void* d_nodes, int64_t nodesBytes, const void* d_triWoop,
                      int64_t triWoopBytes, const int32_t* d_triIndex, int32_t layout, uint32_t bvhFlags,
                      void* stream, float* seconds, NtrTraceStats* stats, NtrSchedHint* hint = nullptr)
{
    if (seconds) *seconds = 0.0f;
    if (stats) memset(stats, 0, sizeof(*stats));
    const KernelInfo* k = find_kernel(kernelName);
    if (!k) return set_error(NTR_ERR_UNKNOWN_KERNEL, "unknown kernel '%s'", kernelName ? kernelName : "(null)");
    if (numRays < 0) return set_error(NTR_ERR_INVALID, "ntr_trace_bvh: numRays < 0");
    if (numRays == 0) return NTR_OK;  // CudaBVHTracer.cpp:92-94
    if (!d_nodes || !d_triWoop || !d_triIndex)
        return set_error(NTR_ERR_INVALID, "CudaBVHTracer: No BVH!");  // :97-98
    if (layout != k->cfg.bvhLayout)
        return set_error(NTR_ERR_LAYOUT, "CudaBVHTracer: Incorrect BVH layout!");  // :99-100
    if (!d_rays || !d_results) return set_error(NTR_ERR_INVALID, "ntr_trace_bvh: null ray/result buffer");
    // The sizes play the role of the reference's texref extents (setTexRef(..., size),
    // CudaBVHTracer.cpp:142-150); buffer descriptors address at most 4 GiB.
    // Compact child pointers are S32 byte offsets and 0x76543210 is the traversal's stack sentinel
    // (EntrypointSentinel, CudaTracerKernels.hpp:38): a node at or beyond that offset cannot be addressed.
    if (nodesBytes < 64 || (nodesBytes % 64) != 0 || nodesBytes > kMaxNodesBytes)
        return set_error(NTR_ERR_INVALID, "ntr_trace_bvh: node buffer size must be a multiple of 64 in [64, 0x76543200]");
    if (triWoopBytes < 16 || (triWoopBytes % 16) != 0 || triWoopBytes > 0xFFFFFFFFll)
        return set_error(NTR_ERR_INVALID, "ntr_trace_bvh: triWoop buffer size must be a multiple of 16 in [16, 4 GiB)");

    DeviceState* ds = nullptr;
    int rc = get_device_state(&ds);
    if (rc != NTR_OK) return rc;
    hipStream_t s = (hipStream_t)stream;

    // ---- plan: a pure function of the tunables and the batch (trace_plan.h; ntr_trace_plan exposes it to the CPU test tier) ----------
    const Tunables tun = tunables();
    TraceBatchDesc bd;
    bd.variant = k->variant;
    bd.dynamicFetch = strcmp(k->name, "kepler_dynamic_fetch") == 0;
    bd.numRays = numRays;
    bd.anyHit = anyHit != 0;
    bd.nodesBytes = nodesBytes; bd.triWoopBytes = triWoopBytes;
    bd.nodesAddr = (uint64_t)d_nodes; bd.woopAddr = (uint64_t)d_triWoop;
    bd.bvhFlags = bvhFlags;
    bd.wantStats = stats != nullptr;
    bd.capturing = stream_is_capturing(s);
    bd.callerHint = hint != nullptr;
    bd.numCUs = ds->numCUs;
    const TracePlan pl = plan_trace(tun, bd);
    const int variant = pl.variant, orderBlocks = pl.orderBlocks;

    // ---- launch: bind the run-time state the plan asks for (counters, hints, prediction scratch), then the kernels ------------------
    TraceParams p;
    p.numRays = numRays;
    p.anyHit = anyHit ? 1 : 0;
    p.rays = d_rays;
    p.results = d_results;
    p.nodes = d_nodes;
    p.woop = d_triWoop;
    p.nodesBytes = (uint32_t)nodesBytes;
    p.woopBytes = (uint32_t)triWoopBytes;
    p.triIndex = d_triIndex;
    p.status = ds->status;
    p.counter = nullptr;
    p.shardRays = 0;
    p.numHeads = pl.numHeads;
    p.numBlocks = 0;
    p.numBlocksIncoherent = 0;
    p.numBlocksDivergent = 0;
    p.orderBlocks = 0;
    p.chunk = pl.chunk;
    p.fetchThreshold = pl.fetchThreshold;
    p.wholeWave = pl.wholeWave;
    p.prefetchAfter = pl.prefetchAfter;
    p.bvhFlags = bvhFlags;
    p.flatFetch = pl.flatFetch;
    p.uniformPrologue = pl.uniformPrologue;
    p.splitSlice = pl.splitSlice;
    p.leafSwitchBelow = pl.leafSwitchBelow;
    p.octant = pl.octant;
    p.stats = ds->stats;
    p.order = nullptr;
    p.cost = nullptr;
    p.poolK = nullptr;
    p.poolKConst = pl.poolKConst;
    p.routeSkip = 0;
    if (stats) NTR_HIP(hipMemsetAsync(ds->stats, 0, 4 * sizeof(unsigned long long), s));

    // no hint from the caller: the library's own, keyed by (stream, batch, BVH)
    if (pl.useAutoHint) {
        rc = auto_hint_get(d_rays, d_nodes, numRays, anyHit ? 1 : 0, s, orderBlocks, &hint);
        if (rc != NTR_OK) return rc;
    }
    // Scheduling hint: the per-ray kernel dispatches blocks in the hint's order, the persistent kernels hand their pool out in it; on
    // refresh launches per-block costs are recorded, from which the next order is derived right after the launch.
    bool refresh = false;
    if (hint && pl.hintable) {
        int dev = 0;
        NTR_HIP(hipGetDevice(&dev));
        if (hint->numBlocks != orderBlocks || hint->device != dev) {
            rc = sched_hint_bind(hint, orderBlocks, dev);   // (automatic hints arrive bound: auto_hint_get)
            if (rc != NTR_OK) return rc;
        }
        const HintStep hs = plan_hint_step(tun, hint->valid, hint->predicted, hint->uses);
        if (hs.zeroK) {
            const hipError_t zk = ntr_launch_zero_words(hint->order + orderBlocks, 3, s);
            if (zk != hipSuccess) return hip_fail(zk, "zero_words launch");
        }
        refresh = hs.refresh;
        hint->predicted = false;
        hint->uses++;
        if (hs.useOrder) p.order = hint->order;
        if (refresh) {
            const hipError_t ze = ntr_launch_zero_words(hint->cost, orderBlocks, s);
            if (ze != hipSuccess) return hip_fail(ze, "zero_words launch");
            p.cost = hint->cost;
        }
    }

    // Dispatch-order prediction (plan_trace: which launches qualify).  A launch whose hint holds no measured order yet -- the first one
    // of a batch -- is predicted like an unhinted one.
    TopTable* predTable = nullptr;
    PredictScratch* predScratch = nullptr;
    if (pl.predictable && !(hint && hint->valid) && !p.order) {
        rc = top_table_get(d_nodes, nodesBytes, s, false, &predTable);
        if (rc != NTR_OK) return rc;
        rc = predict_scratch_get(s, orderBlocks, &predScratch);
        if (rc != NTR_OK) return rc;
        if (predScratch) p.order = predScratch->order;
        else predTable = nullptr;   // (a captured launch that found no spare scratch: buffer order)
    }

    // A hinted batch is predicted once (its hint then holds a measured order); its coherence words -- the batch word: mini-pool K, routing --
    // are estimated again on the hint's refresh launches by a probe of their own (three small launches, on the refresh launches only -- trace_plan.h plan_hint_step): rays drift.
    const bool probeCoherence = !predScratch && hint && refresh && pl.probeOnRefresh;
    if (probeCoherence) {
        rc = top_table_get(d_nodes, nodesBytes, s, false, &predTable);
        if (rc != NTR_OK) return rc;
    }
    // The batch word (sched_kernels.hip pool_k): the prediction of this launch writes it, or the batch's hint kept it from its first,
    // predicted launch (zero -- "coherent", K = 1 -- when there never was one).  It sets the mini-pool depth of the per-ray launch, the grid
    // and refill policy of a persistent launch, and -- routed launches -- which of the two bodies works.
    const unsigned int* word = nullptr;
    if (predScratch) word = predScratch->classCount + NTR_SCHED_PRED_CLASSES + 2;
    else if (hint && pl.hintable && hint->numBlocks == orderBlocks && hint->order) word = hint->order + orderBlocks + 2;
    const bool routed = pl.coherentRoute == 1 && word != nullptr;
    const bool persistentSide = variant == NTR_VARIANT_PERSISTENT || routed;
    const bool perraySide = variant != NTR_VARIANT_PERSISTENT || routed;
    if ((pl.minipool && pl.poolKFromDevice) || persistentSide) p.poolK = word;

    TraceParams pp = p;   // the persistent side's parameters
    if (persistentSide) {
        {
            // a launch that is being captured into a HIP graph keeps its pool heads for the graph's lifetime
            std::lock_guard<std::mutex> lk(g_mu);
            if (bd.capturing) {
                if (ds->nextPinned >= kPinnedCounters)
                    return set_error(NTR_ERR_NOMEM, "ntr_trace_bvh: more than %d persistent launches captured into HIP graphs", kPinnedCounters);
                pp.counter = ds->counters + kPoolHeadsMax * 16 * (kNumCounters + ds->nextPinned++);
            } else {
                pp.counter = ds->counters + kPoolHeadsMax * 16 * ds->next;
                ds->next = (ds->next + 1) % kNumCounters;
            }
        }
        pp.numBlocks = pl.persistentBlocks;
        pp.numBlocksIncoherent = pl.numBlocksIncoherent;
        pp.numBlocksDivergent = pl.numBlocksDivergent;
        pp.fetchThreshold = pl.persistentFetchThreshold;
        pp.shardRays = pl.shardRays;
        if (pp.order) {   // every head walks its share of the order: ranges of whole 256-ray blocks
            pp.orderBlocks = orderBlocks;
            pp.shardRays = ((orderBlocks + pp.numHeads - 1) / pp.numHeads) * 256;
        }
        pp.routeSkip = routed ? NTR_ROUTE_SKIP_COHERENT : 0;
    }
    if (routed) {
        p.routeSkip = NTR_ROUTE_SKIP_INCOHERENT;
        if (variant == NTR_VARIANT_PERSISTENT) { p.fetchThreshold = pl.perrayFetchThreshold; p.poolKConst = 1; }
    }

    struct EventPair {   // destroyed on every return path of the timed bracket
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    hipEvent_t& ev0 = ev.a;
    hipEvent_t& ev1 = ev.b;
    if (seconds) {
        NTR_HIP(hipEventCreate(&ev0));
        NTR_HIP(hipEventCreate(&ev1));
        NTR_HIP(hipStreamSynchronize(s));  // launchTimed syncs first (CudaKernel.cpp:193)
        NTR_HIP(hipEventRecord(ev0, s));
    }
    if (predScratch) {  // inside the timed bracket: the prediction is part of what the launch costs
        const hipError_t pe = ntr_launch_predict(d_rays, numRays, orderBlocks, predTable->table, predTable->count, predScratch->classCount,
                                                 predScratch->classList, predScratch->order,
                                                 (hint && pl.hintable) ? hint->order + orderBlocks + 2 : nullptr, pl.minipoolWide, s);
        if (pe != hipSuccess) return hip_fail(pe, "predict launch");
    } else if (probeCoherence) {   // (also inside the bracket)
        const hipError_t ce = ntr_launch_coherence(d_rays, numRays, orderBlocks, predTable->table, predTable->count, hint->order + orderBlocks,
                                                  pl.minipoolWide, s);
        if (ce != hipSuccess) return hip_fail(ce, "coherence launch");
    }
    hipError_t le = hipSuccess;
    if (persistentSide) {
        // the pool heads are cleared by a kernel: memset nodes do not survive HIP graph replays (see sched_kernels.hip)
        le = ntr_launch_zero_words(pp.counter, kPoolHeadsMax * 16, s);
        if (le != hipSuccess) return hip_fail(le, "zero_words launch");
        le = ntr_launch_trace(pl.persistentVariant, &pp, pl.persistentBlocks, s);
        if (le != hipSuccess) return hip_fail(le, "trace_bvh launch");
    }
    if (perraySide) {
        le = variant == NTR_VARIANT_PERSISTENT ? ntr_launch_trace(NTR_VARIANT_PERRAY_UNIFIED_MINI, &p, pl.perrayBlocks, s)
                                               : ntr_launch_trace(pl.launchVariant, &p, pl.launchBlocks, s);
        if (le != hipSuccess) return hip_fail(le, "trace_bvh launch");
    }
    if (seconds) NTR_HIP(hipEventRecord(ev1, s));
    if (refresh) {
        le = ntr_launch_sched_order(hint->cost, orderBlocks, tun.schedClasses, hint->order, s);
        if (le != hipSuccess) return hip_fail(le, "sched_order launch");
        hint->valid = true;
    }
    if (seconds) {
        NTR_HIP(hipEventSynchronize(ev1));
        float ms = 0.0f;
        NTR_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        *seconds = ms * 1e-3f;
        unsigned int st = 0;   // fetch-and-clear in one device-side step (the word is shared by all streams of the device)
        std::lock_guard<std::mutex> slk(g_statusMu);
        const hipError_t xe = ntr_launch_status_exchange(ds->status, ds->status + 8, s);
        if (xe != hipSuccess) return hip_fail(xe, "status_exchange launch");
        NTR_HIP(hipMemcpyAsync(&st, ds->status + 8, sizeof(st), hipMemcpyDeviceToHost, s));
        NTR_HIP(hipStreamSynchronize(s));
        if (st & NTR_STATUS_STACK_OVERFLOW) return set_error(NTR_ERR_OVERFLOW, "trace_bvh: traversal stack overflow");
    }
    if (stats) {
        unsigned long long h[4];
        NTR_HIP(hipMemcpyAsync(h, ds->stats, sizeof(h), hipMemcpyDeviceToHost, s));
        NTR_HIP(hipStreamSynchronize(s));
        stats->numRays = numRays;
        stats->numInnerVisits = (int64_t)h[0];
        stats->numTriTests = (int64_t)h[1];
        stats->numLeafVisits = (int64_t)h[2];
        stats->numHits = (int64_t)h[3];
    }
    return NTR_OK;
}

int ntr_trace_bvh(const char* kernelName, int32_t numRays, int32_t anyHit, const NtrRay* d_rays,
                  NtrRayResult* d_results, const void* d_nodes, int64_t nodesBytes, const void* d_triWoop,
                  int64_t triWoopBytes, const int32_t* d_triIndex, int32_t layout, uint32_t bvhFlags, void* stream,
                  float* seconds)
{
    return trace_impl(kernelName, numRays, anyHit, d_rays, d_results, d_nodes, nodesBytes, d_triWoop, triWoopBytes,
                      d_triIndex, layout, bvhFlags, stream, seconds, nullptr);
}

int ntr_trace_bvh_hinted(const char* kernelName, int32_t numRays, int32_t anyHit, const NtrRay* d_rays,
                         NtrRayResult* d_results, const void* d_nodes, int64_t nodesBytes, const void* d_triWoop,
                         int64_t triWoopBytes, const int32_t* d_triIndex, int32_t layout, uint32_t bvhFlags, void* stream,
                         float* seconds, NtrSchedHint* hint)
{
    return trace_impl(kernelName, numRays, anyHit, d_rays, d_results, d_nodes, nodesBytes, d_triWoop, triWoopBytes,
                      d_triIndex, layout, bvhFlags, stream, seconds, nullptr, hint);
}

int ntr_trace_plan(const char* kernelName, int32_t numRays, int32_t anyHit, uint64_t nodesAddr, int64_t nodesBytes, uint64_t triWoopAddr,
                   int64_t triWoopBytes, uint32_t bvhFlags, int32_t numCUs, int32_t flags, NtrTracePlan* plan)
{
    if (!plan) return set_error(NTR_ERR_INVALID, "ntr_trace_plan: null plan");
    memset(plan, 0, sizeof(*plan));
    const KernelInfo* k = find_kernel(kernelName);
    if (!k) return set_error(NTR_ERR_UNKNOWN_KERNEL, "unknown kernel '%s'", kernelName ? kernelName : "(null)");
    if (numRays < 0 || numCUs < 1 || nodesBytes < 0 || triWoopBytes < 0) return set_error(NTR_ERR_INVALID, "ntr_trace_plan: bad argument");
    TraceBatchDesc bd;
    bd.variant = k->variant;
    bd.dynamicFetch = strcmp(k->name, "kepler_dynamic_fetch") == 0;
    bd.numRays = numRays;
    bd.anyHit = anyHit != 0;
    bd.nodesBytes = nodesBytes; bd.triWoopBytes = triWoopBytes;
    bd.nodesAddr = nodesAddr; bd.woopAddr = triWoopAddr;
    bd.bvhFlags = bvhFlags;
    bd.wantStats = (flags & NTR_PLAN_FLAG_STATS) != 0;
    bd.capturing = (flags & NTR_PLAN_FLAG_CAPTURING) != 0;
    bd.callerHint = (flags & NTR_PLAN_FLAG_CALLER_HINT) != 0;
    bd.numCUs = numCUs;
    *plan = plan_trace(tunables(), bd);
    return NTR_OK;
}

int ntr_trace_plan_hint_step(int32_t valid, int32_t predicted, int32_t uses, int32_t out[3])
{
    if (!out || uses < 0) return set_error(NTR_ERR_INVALID, "ntr_trace_plan_hint_step: bad argument");
    const HintStep h = plan_hint_step(tunables(), valid != 0, predicted != 0, uses);
    out[0] = h.zeroK; out[1] = h.refresh; out[2] = h.useOrder;
    return NTR_OK;
}

int ntr_trace_status(void* stream, uint32_t* statusBits)
{
    if (statusBits) *statusBits = 0;
    DeviceState* ds = nullptr;
    const int rc = get_device_state(&ds);
    if (rc != NTR_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    unsigned int st = 0;
    // fetched and cleared in ONE device-side step: a bit set by a launch on another stream is either in this answer or in the next
    std::lock_guard<std::mutex> lk(g_statusMu);   // the result word ds->status[8] is shared by the callers of one device
    const hipError_t xe = ntr_launch_status_exchange(ds->status, ds->status + 8, s);
    if (xe != hipSuccess) return hip_fail(xe, "status_exchange launch");
    NTR_HIP(hipMemcpyAsync(&st, ds->status + 8, sizeof(st), hipMemcpyDeviceToHost, s));
    NTR_HIP(hipStreamSynchronize(s));
    if (statusBits) *statusBits = st;
    if (st & NTR_STATUS_STACK_OVERFLOW)
        return set_error(NTR_ERR_OVERFLOW, "trace_bvh: traversal stack overflow in a launch since the last status check");
    return NTR_OK;
}


int ntr_predict_block_costs(int32_t numRays, const NtrRay* d_rays, const void* d_nodes, int64_t nodesBytes, uint32_t* d_blockCost, void* stream)
{
    if (numRays < 0) return set_error(NTR_ERR_INVALID, "ntr_predict_block_costs: numRays < 0");
    if (numRays == 0) return NTR_OK;
    if (!d_rays || !d_nodes || !d_blockCost) return set_error(NTR_ERR_INVALID, "ntr_predict_block_costs: null argument");
    if (nodesBytes < 64 || (nodesBytes % 64) != 0 || nodesBytes > kMaxNodesBytes)
        return set_error(NTR_ERR_INVALID, "ntr_predict_block_costs: node buffer size must be a multiple of 64 in [64, 0x76543200]");
    TopTable* t = nullptr;
    const int rc = top_table_get(d_nodes, nodesBytes, (hipStream_t)stream, false, &t);
    if (rc != NTR_OK) return rc;
    const hipError_t e = ntr_launch_predict_costs(d_rays, numRays, (numRays + 255) / 256, t->table, t->count, d_blockCost, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "predict_costs launch");
    return NTR_OK;
}

int ntr_predict_batch_coherence(int32_t numRays, const NtrRay* d_rays, const void* d_nodes, int64_t nodesBytes, uint32_t* d_out, void* stream)
{
    if (numRays < 0) return set_error(NTR_ERR_INVALID, "ntr_predict_batch_coherence: numRays < 0");
    if (!d_out) return set_error(NTR_ERR_INVALID, "ntr_predict_batch_coherence: null argument");
    if (numRays > 0 && (!d_rays || !d_nodes)) return set_error(NTR_ERR_INVALID, "ntr_predict_batch_coherence: null argument");
    if (numRays == 0) {   // nothing to look at: {0, 0, K = 1} without touching the node buffer
        const unsigned int none[3] = {0u, 0u, 1u};
        NTR_HIP(hipMemcpyAsync(d_out, none, sizeof(none), hipMemcpyHostToDevice, (hipStream_t)stream));
        NTR_HIP(hipStreamSynchronize((hipStream_t)stream));
        return NTR_OK;
    }
    if (nodesBytes < 64 || (nodesBytes % 64) != 0 || nodesBytes > kMaxNodesBytes)
        return set_error(NTR_ERR_INVALID, "ntr_predict_batch_coherence: node buffer size must be a multiple of 64 in [64, 0x76543200]");
    TopTable* t = nullptr;
    const int rc = top_table_get(d_nodes, nodesBytes, (hipStream_t)stream, false, &t);
    if (rc != NTR_OK) return rc;
    const hipError_t e = ntr_launch_coherence(d_rays, numRays, (numRays + 255) / 256, t->table, t->count, d_out, minipool_wide(tunables(), nodesBytes, numRays),
                                              (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "coherence launch");
    return NTR_OK;
}

int ntr_trace_graph_reserve(int32_t launches, int32_t numRays)
{
    if (launches < 0 || numRays < 0) return set_error(NTR_ERR_INVALID, "ntr_trace_graph_reserve: negative argument");
    DeviceState* ds = nullptr;
    int rc = get_device_state(&ds);
    if (rc != NTR_OK) return rc;
    int dev = 0;
    NTR_HIP(hipGetDevice(&dev));
    const int numBlocks = (numRays + 255) / 256;
    SchedState* ss = nullptr;
    rc = sched_state(dev, &ss);
    if (rc != NTR_OK) return rc;
    std::lock_guard<std::mutex> lk(ss->mu);
    if (numBlocks > 0 && launches > 0) {
        rc = scratch_provision_spares(ss, dev, numBlocks, launches);
        if (rc != NTR_OK) return rc;
        if (scratch_count_spares(ss, dev, numBlocks) < launches)
            return set_error(NTR_ERR_NOMEM, "ntr_trace_graph_reserve: at most %d prediction scratches exist (ntr_trace_graph_release_all returns the pinned ones)", kScratch);
    }
    return NTR_OK;
}

int ntr_trace_graph_release_all(void)
{
    int dev = 0;
    NTR_HIP(hipGetDevice(&dev));
    NTR_HIP(hipDeviceSynchronize());   // replays in flight still read the scratch
    SchedState* ss = nullptr;
    const int rc = sched_state(dev, &ss);
    if (rc != NTR_OK) return rc;
    {
        std::lock_guard<std::mutex> lk(ss->mu);
        for (auto& e : ss->scratch)
            if (e.state == PredictScratch::PINNED) e.state = PredictScratch::SPARE;   // back to the spares
        for (auto& t : ss->top) t.pinned = false;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[dev].nextPinned = 0;
    return NTR_OK;
}

int ntr_stream_release(void* stream)
{
    // The library keeps per-stream scheduling state on the current device -- automatic hints (their arrays come from the stream-ordered
    // allocator) and the stream's prediction scratch -- and recycles an entry only for the stream that owns it (nothing on the launch path
    // synchronises).  A host that destroys a stream returns that state here first; otherwise entries of dead streams stay allocated for
    // the life of the process, and a table full of them (96 hints per device) leaves later streams without the learned dispatch order.
    SchedState* ss = nullptr;
    int dev = 0;
    int rc = sched_state_current(&ss, &dev);
    if (rc != NTR_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (stream_is_capturing(s)) return set_error(NTR_ERR_INVALID, "ntr_stream_release: the stream is being captured");
    NTR_HIP(hipStreamSynchronize(s));   // launches in flight still read the arrays
    std::lock_guard<std::mutex> lk(ss->mu);
    for (auto& e : ss->autoHints)
        if (e.used && e.stream == stream) {
            if (e.hint.order) auto_hint_release_async(&e.hint, s);
            e = AutoHint();
        }
    for (auto& e : ss->scratch)
        if (e.state == PredictScratch::LIVE && e.stream == stream) scratch_free(&e);
    return NTR_OK;
}

int ntr_selftest_auto_hint_table(int32_t devices, int32_t keysPerDevice, int32_t rounds, int32_t* hintedLastRound)
{
    // CPU-side check of the automatic-hint table logic (no device needed): `devices` fake devices, each tracing `keysPerDevice` distinct
    // batches on one stream of its own, `rounds` times over; hintedLastRound[d] = batches of device d that found their entry in the last
    // round.  With per-device tables every device is served alike (keysPerDevice each while keysPerDevice <= 96).
    if (devices < 1 || devices > kMaxDevices || keysPerDevice < 1 || rounds < 1 || !hintedLastRound)
        return set_error(NTR_ERR_INVALID, "ntr_selftest_auto_hint_table: bad argument");
    for (int d = 0; d < devices; d++) hintedLastRound[d] = 0;
    SchedState* fake = new (std::nothrow) SchedState[devices];   // (never touches g_sched: the live tables stay as they are)
    if (!fake) return set_error(NTR_ERR_NOMEM, "ntr_selftest_auto_hint_table: out of host memory");
    for (int r = 0; r < rounds; r++)
        for (int k = 0; k < keysPerDevice; k++)        // batch by batch across the devices, as eight host threads would interleave
            for (int d = 0; d < devices; d++) {
                NtrSchedHint* evicted = nullptr;
                bool noEntry = false;
                std::lock_guard<std::mutex> lk(fake[d].mu);
                AutoHint* hit = auto_hint_lookup(&fake[d], (const void*)(uintptr_t)(0x1000 + 64 * k), (const void*)(uintptr_t)0x10, 1 << 20, k & 1,
                                                 (void*)(uintptr_t)(d + 1), d, &evicted, &noEntry);
                if (hit) hit->hint.order = (unsigned int*)(uintptr_t)1;   // stands for "storage allocated" (a later eviction must not take it from another stream)
                if (hit && r == rounds - 1) hintedLastRound[d]++;
            }
    delete[] fake;
    return NTR_OK;
}

int ntr_sched_hint_create(NtrSchedHint** out)
{
    if (!out) return set_error(NTR_ERR_INVALID, "ntr_sched_hint_create: null out");
    *out = new (std::nothrow) NtrSchedHint();
    return *out ? NTR_OK : set_error(NTR_ERR_NOMEM, "ntr_sched_hint_create: out of memory");
}

int ntr_sched_hint_destroy(NtrSchedHint* hint)
{
    if (!hint) return NTR_OK;
    sched_hint_release(hint);
    delete hint;
    return NTR_OK;
}

int ntr_sched_hint_reset(NtrSchedHint* hint)
{
    if (!hint) return set_error(NTR_ERR_INVALID, "ntr_sched_hint_reset: null hint");
    hint->uses = 0;
    hint->valid = false;
    hint->predicted = false;
    return NTR_OK;
}

int ntr_sched_hint_predict(NtrSchedHint* hint, const uint32_t* d_blockCost, int32_t numBlocks, void* stream)
{
    if (!hint || !d_blockCost || numBlocks < 1) return set_error(NTR_ERR_INVALID, "ntr_sched_hint_predict: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    NTR_HIP(hipGetDevice(&dev));
    if (hint->numBlocks != numBlocks || hint->device != dev) {   // (the binding ntr_trace_bvh_hinted would make on first use)
        const int rc = sched_hint_bind(hint, numBlocks, dev);
        if (rc != NTR_OK) return rc;
    }
    NTR_HIP(hipMemcpyAsync(hint->cost, d_blockCost, (size_t)numBlocks * sizeof(unsigned int), hipMemcpyDeviceToDevice, s));
    hipError_t le = ntr_launch_sched_order(hint->cost, numBlocks, tunables().schedClasses, hint->order, s);
    if (le == hipSuccess) le = ntr_launch_zero_words(hint->order + numBlocks, 3, s);   // the batch's coherence words / pool K: not estimated yet
    if (le != hipSuccess) return hip_fail(le, "sched_order launch");
    hint->uses = 0;      // the next launch starts the hint's life: it runs this order; the launches after it measure and refine
    hint->valid = true;
    hint->predicted = true;
    return NTR_OK;
}

int ntr_secondary_block_costs(const NtrRayResult* d_inResults, int32_t first, int32_t count, int32_t numSamples, const int32_t* d_depthByTri,
                              int32_t numTris, uint32_t* d_blockCost, void* stream)
{
    if (first < 0 || count < 0 || numSamples < 1 || numTris < 0 || (count > 0 && (!d_inResults || !d_depthByTri || !d_blockCost)))
        return set_error(NTR_ERR_INVALID, "ntr_secondary_block_costs: bad argument");
    if (count == 0) return NTR_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t blocks = ((int64_t)count * numSamples + 255) / 256;
    if (blocks > 0x7FFFFFFF) return set_error(NTR_ERR_INVALID, "ntr_secondary_block_costs: batch too large");
    hipError_t e = ntr_launch_zero_words(d_blockCost, (int)blocks, s);
    if (e == hipSuccess) e = ntr_launch_secondary_block_costs(d_inResults, first, count, numSamples, d_depthByTri, numTris, d_blockCost, s);
    if (e != hipSuccess) return hip_fail(e, "secondary_block_costs launch");
    return NTR_OK;
}

int ntr_bvh_leaf_depths(const void* d_nodes, int64_t nodesBytes, const void* d_triWoop, int64_t triWoopBytes, const int32_t* d_triIndex,
                        int32_t numTris, int32_t* d_depthByTri, int32_t* maxDepth, void* stream)
{
    if (maxDepth) *maxDepth = 0;
    if (!d_nodes || nodesBytes < 64 || (nodesBytes % 64) != 0 || nodesBytes > 0x76543200ll || !d_triWoop || triWoopBytes < 16 || !d_triIndex ||
        numTris < 1 || !d_depthByTri)
        return set_error(NTR_ERR_INVALID, "ntr_bvh_leaf_depths: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const unsigned int capacity = (unsigned int)(nodesBytes / 64);
    // two frontier queues + their counters (counter k of level L at counts[L & 1]; cleared before it is filled)
    unsigned int* d_q = nullptr;
    NTR_HIP(hipMalloc((void**)&d_q, (2 * (size_t)capacity + 2) * sizeof(unsigned int)));
    unsigned int* q[2] = {d_q, d_q + capacity};
    unsigned int* cnt = d_q + 2 * (size_t)capacity;
    hipError_t e = hipMemsetAsync(d_depthByTri, 0, (size_t)numTris * sizeof(int32_t), s);
    const unsigned int init[3] = {0u, 1u, 0u};   // q[0][0] = root offset 0; counts = {1, 0}
    if (e == hipSuccess) e = hipMemcpyAsync(q[0], &init[0], sizeof(unsigned int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(cnt, &init[1], 2 * sizeof(unsigned int), hipMemcpyHostToDevice, s);
    int depth = 0;
    unsigned long long bound = 1;   // the frontier of level L holds at most min(2^L, capacity) nodes
    unsigned int frontier = 1;
    while (e == hipSuccess && frontier > 0 && depth < 4096) {
        const int in = depth & 1, out = in ^ 1;
        e = ntr_launch_zero_words(cnt + out, 1, s);
        const unsigned int threads = (unsigned int)(bound < capacity ? bound : capacity);
        if (e == hipSuccess)
            e = ntr_launch_leaf_depth_level(d_nodes, (unsigned int)nodesBytes, d_triWoop, (unsigned int)(triWoopBytes / 16), d_triIndex, numTris, q[in], cnt + in,
                                            q[out], cnt + out, capacity, threads, depth, d_depthByTri, s);
        depth++;
        bound = bound < capacity ? bound * 2 : bound;
        if ((depth & 3) == 0 && e == hipSuccess) {   // every fourth level: has the frontier run empty?
            e = hipMemcpyAsync(&frontier, cnt + (depth & 1), sizeof(frontier), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (frontier > capacity) frontier = capacity;
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_q);
    if (e != hipSuccess) return hip_fail(e, "ntr_bvh_leaf_depths");
    if (depth >= 4096) return set_error(NTR_ERR_INVALID, "ntr_bvh_leaf_depths: the tree is deeper than 4096 levels (a cycle in the child references?)");
    if (maxDepth) *maxDepth = depth;
    return NTR_OK;
}

int ntr_trace_bvh_stats(const char* kernelName, int32_t numRays, int32_t anyHit, const NtrRay* d_rays,
                        NtrRayResult* d_results, const void* d_nodes, int64_t nodesBytes, const void* d_triWoop,
                        int64_t triWoopBytes, const int32_t* d_triIndex, int32_t layout, uint32_t bvhFlags,
                        void* stream, NtrTraceStats* stats)
{
    if (!stats) return set_error(NTR_ERR_INVALID, "ntr_trace_bvh_stats: null stats");
    return trace_impl(kernelName, numRays, anyHit, d_rays, d_results, d_nodes, nodesBytes, d_triWoop, triWoopBytes,
                      d_triIndex, layout, bvhFlags, stream, nullptr, stats);
}

int ntr_selftest_division(const float* d_x, int32_t nx, const float* d_d, int32_t nd, uint32_t* mismatches, void* stream)
{
    if (!mismatches || !d_x || !d_d || nx <= 0 || nd <= 0) return set_error(NTR_ERR_INVALID, "ntr_selftest_division: bad argument");
    hipStream_t s = (hipStream_t)stream;
    unsigned int* d_m = nullptr;
    NTR_HIP(hipMalloc((void**)&d_m, sizeof(unsigned int)));
    NTR_HIP(hipMemsetAsync(d_m, 0, sizeof(unsigned int), s));
    hipError_t le = ntr_launch_selftest_division(d_x, d_d, nx, nd, d_m, s);
    if (le != hipSuccess) return hip_fail(le, "selftest launch");
    unsigned int m = 0;
    NTR_HIP(hipMemcpyAsync(&m, d_m, sizeof(m), hipMemcpyDeviceToHost, s));
    NTR_HIP(hipStreamSynchronize(s));
    NTR_HIP(hipFree(d_m));
    *mismatches = m;
    return NTR_OK;
}

int ntr_selftest_division_hard(int32_t xExp, int32_t dExp, uint64_t* pairs, uint64_t* mismatches, void* stream)
{
    // FASTDIV range of the operands the kernel forms (x = X 2^(xExp .. xExp + 3 - 23), d = D 2^(dExp .. dExp + 3 - 23), X, D in [2^23, 2^24))
    if (!pairs || !mismatches || xExp < -84 || xExp > 51 || dExp < -40 || dExp > 16)
        return set_error(NTR_ERR_INVALID, "ntr_selftest_division_hard: bad argument (x exponents in [-84, 51], d exponents in [-40, 16])");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* d_c = nullptr;
    NTR_HIP(hipMalloc((void**)&d_c, 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_c, 0, 2 * sizeof(unsigned long long), s);
    if (e == hipSuccess) e = ntr_launch_selftest_division_hard(xExp, dExp, d_c, s);
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_c, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_c);
    if (e != hipSuccess) return hip_fail(e, "ntr_selftest_division_hard");
    *pairs = h[0];
    *mismatches = h[1];
    return NTR_OK;
}

}  // extern "C"
