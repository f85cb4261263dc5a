// trace_plan.h -- the PLAN of a trace launch: everything ntr_trace_bvh decides from the tunables, the kernel name and the batch's
// sizes and flags alone, as a pure function (no HIP call, no global state).  ntr_api.cpp's trace_impl validates, plans, then launches;
// ntr_trace_plan() exposes the plan through the C-ABI so that the CPU test tier can check it without a device
// (tests/test_trace_plan_cpu.py).  What depends on run-time state -- which hint / prediction scratch / counters a launch gets -- stays
// in the launch half.
#pragma once
#include <stdint.h>
#include <string.h>

#include "ntrace_amd.h"
#include "ntr_internal.h"
#include "trace_kernels.h"

namespace ntr {

static constexpr int kPoolHeadsMax = 1024;

struct TraceBatchDesc {
    int variant;            // the kernel name's variant (NTR_VARIANT_PERRAY / NTR_VARIANT_PERSISTENT)
    bool dynamicFetch;      // the name is kepler_dynamic_fetch
    int32_t numRays;
    bool anyHit;
    int64_t nodesBytes, triWoopBytes;
    uint64_t nodesAddr, woopAddr;
    uint32_t bvhFlags;
    bool wantStats;         // ntr_trace_bvh_stats: the instrumented per-ray kernel, whatever the name
    bool capturing;         // the stream is being captured into a HIP graph
    bool callerHint;        // the caller passed an NtrSchedHint
    int numCUs;
};

typedef NtrTracePlan TracePlan;   // include/ntrace_amd.h: plain int32 fields, so that ntr_trace_plan can hand it out as it is

// Pool K of an incoherent batch.  Long rays (big trees) amortise a deeper private pool, and the batch must oversubscribe the machine
// (rays / 64 / K waves against 7 168 wave slots): below that a launch is bound by its longest rays, and fewer, longer-lived waves only
// lengthen that path.  Box rays, K = 1 / 2 / 4 (scripts/studies/small_batch_minipool.py, profiles/r03_minipool_batch_sizes.jsonl), ms:
//   courtyard-10M  2^19: 2.81 / 3.12 / 3.52   2^20: 3.74 / 3.33 / 3.55   1.5 M: 5.23 / 4.36 / 3.84   2^21: 6.79 / 5.38 / 4.73   2^22: 12.8 / 9.9 / 7.8
//   hairball-2.8M  2^19: 1.76 / 1.50 / 1.69   2^20: 2.79 / 2.38 / 2.39   1.5 M: 3.92 / 3.10 / 3.28   2^21: 5.04 / 3.77 / 3.80   2^22: 9.5 / 6.6 / 6.4
//   atrium-262k    2^19: .230 / .217 / .276   2^20: .374 / .375 / .349   1.5 M: .520 / .483 / .520   2^21: .670 / .598 / .627   2^22: 1.24 / 1.04 / 1.03
inline int minipool_wide(const Tunables& tun, int64_t nodesBytes, int numRays)
{
    if (tun.minipoolWide >= 2 && tun.minipoolWide <= NTR_MINIPOOL_MAX_K) return tun.minipoolWide;
    return (nodesBytes >= (int64_t)32 << 20 && numRays >= (3 << 19)) ? 4 : 2;
}

inline TracePlan plan_trace(const Tunables& tun, const TraceBatchDesc& b)
{
    TracePlan pl;
    memset(&pl, 0, sizeof(pl));
    pl.chunk = tun.chunk;
    {   // the flat fetch addresses both buffers from one scalar base with 32-bit lane offsets: they must lie inside one 4 GiB window
        // (two allocations of one heap practically always do; otherwise the two-descriptor fetch, which has no such condition)
        const uint64_t an = b.nodesAddr, aw = b.woopAddr;
        const uint64_t lo = an < aw ? an : aw;
        const uint64_t hiN = an + (uint64_t)b.nodesBytes, hiW = aw + (uint64_t)b.triWoopBytes;
        const bool oneWindow = ((hiN > hiW ? hiN : hiW) - lo) <= 0xFFFFFFFFull;
        pl.flatFetch = (tun.flatFetch != 0 && b.nodesBytes >= 64 && b.triWoopBytes >= 64 && oneWindow) ? 1 : 0;
    }
    pl.uniformPrologue = tun.uniformPrologue != 0 ? 1 : 0;
    pl.wholeWave = tun.wholeWave != 0 ? 1 : 0;
    pl.prefetchAfter = tun.prefetchAfter;
    pl.splitSlice = tun.splitSlice > 0 ? tun.splitSlice : 0;
    pl.leafSwitchBelow = tun.leafSwitchBelow >= 0 ? tun.leafSwitchBelow : (b.anyHit ? 24 : 32);
    pl.octant = tun.octant;
    pl.numHeads = 8;
    pl.poolKConst = 1;
    pl.minipoolWide = minipool_wide(tun, b.nodesBytes, b.numRays);
    pl.orderBlocks = (b.numRays + 255) / 256;
    const bool bigEnough = b.numRays >= tun.predictMinRays && b.nodesBytes >= (int64_t)tun.predictMinNodes * 64 && tun.predict != 0;

    // RayStats counters (src/rt/bvh/BVH.hpp:44-, filled at CudaBVH.cpp:746-757,1107-1111) are produced by the instrumented per-ray
    // kernel; every variant visits nodes in the same per-ray order, so the counts do not depend on the variant.
    pl.variant = b.wantStats ? NTR_VARIANT_PERRAY_STATS : b.variant;

    // ROUTING (round 6).  The kernel names select semantics, and the semantics of all of them are the same here (identical records by
    // construction); what differs is which batches a body is fast on: the per-ray body on coherent ones (15 against 7.5-8 Grays/s on the
    // headline frame -- a persistent wave idles half of a launch that holds only two to four chunks per wave slot), the persistent
    // dynamic-fetch body with ray splitting on incoherent ones (2.8 against 3.6 ms on 2^21 box rays through the hairball tree).  So, unless
    // NTR_TRACE_ROUTE=0 forces the named body:
    //   * any-hit launches run the per-ray body under every name (coherentRoute 2, decided here);
    //   * closest-hit launches large enough for the device's coherence estimate (the dispatch-order prediction computes it) are launched as
    //     BOTH bodies -- the per-ray one and kepler_dynamic_fetch's persistent one --, each of which leaves at once when the batch word says
    //     the batch is the other's (coherentRoute 1; an empty launch costs 3-10 us);
    //   * everything else runs the named body.
    const bool route = tun.route != 0 && !b.wantStats;
    if (route && b.variant == NTR_VARIANT_PERSISTENT && b.anyHit) {
        pl.variant = NTR_VARIANT_PERRAY;
        pl.coherentRoute = 2;
    }
    // the persistent body of this launch: the name's own when the named body runs, kepler_dynamic_fetch's -- dynamic fetch, unified-step loop,
    // ray splitting: the body that is fast on incoherent batches -- in every routed launch, whatever the name
    const bool routedPersistentName = route && b.variant == NTR_VARIANT_PERSISTENT && !b.anyHit && bigEnough && (256 % pl.chunk) == 0 &&
                                      tun.perrayUnified != 0 && tun.minipool < 0 && tun.predictPersistent != 0;
    const bool persistentDynamic = (b.variant == NTR_VARIANT_PERSISTENT && !routedPersistentName) ? b.dynamicFetch : true;
    pl.unified = persistentDynamic && tun.unified != 0;
    pl.persistentFetchThreshold = tun.fetchThreshold >= 0 ? tun.fetchThreshold : (persistentDynamic ? (pl.unified ? 48 : 24) : 0);
    pl.fetchThreshold = pl.persistentFetchThreshold;

    constexpr int blockThreads = NTR_TRACE_WAVES_PER_BLOCK * 64;
    auto persistent_grid = [&]() {
        // Persistent grid: CUs x resident blocks per CU (the reference hard-codes 720 warps for GT200/Fermi, CudaBVHTracer.cpp:155-159).
        const int blocksPerCU = tun.blocksPerCU;
        pl.persistentBlocks = b.numCUs * blocksPerCU;
        const int needed = (b.numRays + blockThreads - 1) / blockThreads;
        if (pl.persistentBlocks > needed) pl.persistentBlocks = needed;
        const int chunksTotal = (b.numRays + pl.chunk - 1) / pl.chunk;
        pl.numHeads = tun.poolHeads < 8 ? 8 : (tun.poolHeads > kPoolHeadsMax ? kPoolHeadsMax : tun.poolHeads & ~7);
        // (only the dynamic-fetch kernel: its waves stay full from the pool; the while-while persistent kernel refills a wave only when it is
        // empty and loses with fewer waves -- hairball box rays 9.3 -> 14.7 ms)
        if (persistentDynamic && tun.blocksPerCUIncoherent > 0 && tun.blocksPerCUIncoherent < blocksPerCU) {
            pl.numBlocksIncoherent = b.numCUs * tun.blocksPerCUIncoherent;
            if (pl.numBlocksIncoherent > pl.persistentBlocks) pl.numBlocksIncoherent = pl.persistentBlocks;
        }
        if (persistentDynamic && tun.blocksPerCUDivergent > 0 && tun.blocksPerCUDivergent < blocksPerCU) {
            pl.numBlocksDivergent = b.numCUs * tun.blocksPerCUDivergent;
            if (pl.numBlocksDivergent > pl.persistentBlocks) pl.numBlocksDivergent = pl.persistentBlocks;
        }
        pl.shardRays = ((chunksTotal + pl.numHeads - 1) / pl.numHeads) * pl.chunk;
        pl.persistentVariant = pl.unified ? NTR_VARIANT_PERSISTENT_UNIFIED : NTR_VARIANT_PERSISTENT;
    };
    if (pl.variant == NTR_VARIANT_PERSISTENT) {
        persistent_grid();
        pl.numBlocks = pl.persistentBlocks;
    } else {
        pl.numBlocks = (b.numRays + blockThreads - 1) / blockThreads;
    }
    // the per-ray kernel dispatches its blocks in a hint's order; the persistent kernels hand their pool out in it (heavy blocks first:
    // with two to four 64-ray chunks per wave slot the launch ends with whatever long chunk was taken last) and record a block's cost as
    // the life of the whole-wave chunks taken from it
    pl.hintable = pl.variant == NTR_VARIANT_PERRAY || (pl.variant == NTR_VARIANT_PERSISTENT && (256 % pl.chunk) == 0 && tun.persistentHints != 0);
    pl.useAutoHint = !b.callerHint && !b.wantStats && tun.autoHint != 0 && pl.hintable && b.numRays >= tun.autoHintMinRays && !b.capturing;

    // Dispatch-order prediction (sched_kernels.hip): closest-hit launches of the per-ray kernel that are large enough for the tail to
    // outweigh the two small launches (about 30 us; break-even near 1 M rays).  Any-hit batches measured no net gain.  A tree of a few
    // hundred nodes is traced faster than it is predicted: Cornell-box class scenes are left alone.  The persistent kernels hand their
    // pool out in the same predicted order (the heavy blocks' long rays start first instead of being the chunks fetched last): there the
    // prediction covers batches of all 256-ray blocks and needs pool chunks that divide 256.
    pl.persistentOrder = pl.variant == NTR_VARIANT_PERSISTENT && tun.predictPersistent != 0 && (256 % pl.chunk) == 0;
    pl.predictable = (pl.variant == NTR_VARIANT_PERRAY || pl.persistentOrder) && !b.anyHit && bigEnough;
    pl.probeOnRefresh = pl.variant == NTR_VARIANT_PERRAY && !b.anyHit && tun.minipool < 0 && bigEnough;

    // Workgroup size of the per-ray kernel: smaller workgroups retire (and are replaced) sooner.  The dispatch order and the cost
    // feedback stay in units of 256 rays: numBlocks counts those, the launch has 4 / waves workgroups per unit.
    pl.launchVariant = pl.variant;
    pl.launchBlocks = pl.numBlocks;
    if (pl.variant == NTR_VARIANT_PERSISTENT) pl.launchVariant = pl.persistentVariant;
    const int wantWaves = b.anyHit ? tun.anyHitWaves : tun.closestWaves;
    if (pl.variant == NTR_VARIANT_PERRAY && wantWaves < NTR_TRACE_WAVES_PER_BLOCK) {
        const int waves = wantWaves <= 1 ? 1 : 2;
        pl.launchVariant = waves == 1 ? NTR_VARIANT_PERRAY_W1 : NTR_VARIANT_PERRAY_W2;
        pl.launchBlocks = pl.numBlocks * (4 / waves);
        // unified-step loop (one node OR one triangle per lane and iteration, one group of loads): closest-hit launches on any tree
        // (atrium primary +5 %, conference +21 %, LBVH trees +50 %) and any-hit launches (multi-triangle leaves: always ahead; short AO
        // rays in one-triangle-leaf trees: the while-while loop was 2-3 % ahead while a step cost ~100 vector instructions, the unified
        // loop is 4 % ahead since the one-correction divide -- profiles/r04_perray_unified_anyhit_knob.txt)
        if (tun.perrayUnified > 0 || (tun.perrayUnified < 0 && (!b.anyHit || (b.bvhFlags & NTR_BVH_WIDE_LEAVES)))) {
            pl.launchVariant = NTR_VARIANT_PERRAY_UNIFIED_W1;
            pl.launchBlocks = pl.numBlocks * 4;
            // wave-private mini-pool: a wave owns K x 64 rays and refills its finished lanes from them.  K is decided on the device: the
            // prediction of this launch wrote it (incoherent batch: minipoolWide, else 1), or the batch's hint kept it from its first launch.
            if (tun.minipool != 0 && !b.anyHit) {
                pl.launchVariant = NTR_VARIANT_PERRAY_UNIFIED_MINI;
                pl.minipool = true;
                pl.fetchThreshold = tun.minipoolThreshold;
                pl.poolKConst = tun.minipool > 0 ? tun.minipool : 1;
                pl.poolKFromDevice = tun.minipool < 0;
            }
        }
    }

    // closest-hit launches the device can classify: both bodies (see ROUTING above).  The per-ray side is the mini-pool launch -- the
    // instantiation that reads the batch word --, the persistent side the name's own body, or kepler_dynamic_fetch's
    if (route && !b.anyHit && bigEnough && (256 % pl.chunk) == 0) {
        if (pl.variant == NTR_VARIANT_PERRAY && pl.launchVariant == NTR_VARIANT_PERRAY_UNIFIED_MINI && pl.poolKFromDevice) {
            pl.coherentRoute = 1;
            persistent_grid();
        } else if (pl.variant == NTR_VARIANT_PERSISTENT && tun.perrayUnified != 0 && tun.minipool < 0 && pl.predictable) {
            pl.coherentRoute = 1;
        }
    }
    if (pl.coherentRoute == 1) {   // the per-ray side's launch shape (what a fermi launch of this batch has)
        pl.perrayBlocks = pl.orderBlocks * 4;
        pl.perrayFetchThreshold = tun.minipoolThreshold;
    }
    return pl;
}

// One launch in the life of a scheduling hint (pure: the caller applies `uses++`, `predicted = false` afterwards).
struct HintStep {
    bool zeroK;      // a hint that starts over (new, or an automatic one recycled for another batch) forgets its pool K
    bool refresh;    // this launch records per-block costs and the next order is derived from them
    bool useOrder;   // the launch is dispatched in the hint's order
};
inline HintStep plan_hint_step(const Tunables& tun, bool valid, bool predicted, int uses)
{
    HintStep h;
    // costs measured under the natural order differ from those under the derived order, so the first launches all refresh; afterwards
    // the `schedRefreshEvery`-th does, the one twice and the one four times as late, and from then on every fourth period's (slowly
    // drifting rays keep their schedule, and a schedule that has held for 64 launches is measured again less often than a new one: a
    // refresh launch records costs and derives an order -- 25-30 us on a 2^20-ray batch, half of an AO launch's own time; round 6:
    // at 16, 32, 48, ... they were 2.5 % of the headline step)
    const bool firstOfPrediction = predicted && valid;   // (ntr_sched_hint_predict cleared the K words itself)
    h.zeroK = uses == 0 && !firstOfPrediction;
    const int every = tun.schedRefreshEvery;
    const long long late = 4ll * every;
    h.refresh = uses < 3 || every <= 1 || (uses < late ? (uses == every || uses == 2 * every) : (uses % late) == 0);
    if (firstOfPrediction) h.refresh = false;   // the first launch of a predicted order just runs it (a batch traced once pays nothing for feedback)
    h.useOrder = valid;
    return h;
}

}  // namespace ntr
