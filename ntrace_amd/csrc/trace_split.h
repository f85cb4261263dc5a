// trace_split.h -- lanes of a wave that have run out of rays take over parts of the wave's long rays (round 5).
// Included by trace_kernels.hip (needs RayRegs, LaneStack, kSentinel, LDS_DEPTH, NTR_STACK_RESET).
//
// Why.  A divergent closest-hit launch is bound by its longest ray: the 2^21 box rays over the 10 M-triangle tree take 86 steps on
// average, the longest 3 500-3 700, and a step is a dependent fetch -- 1.2 us under load -- so the launch cannot end before
// 3 600 x 1.2 us = 4.3 ms whatever the schedule (EXPERIMENTS.md, "A two-term model").  When the pool has run dry the lanes of a wave go
// idle one by one while its last rays step on alone; an idle lane's fetches cost the wave nothing (the texture path and the vector
// ALU charge per wave instruction, whatever the exec mask).  So the chain itself is cut: a ray's pending work IS its traversal
// stack, the entry at the BOTTOM is the subtree the ray would visit last (usually the largest: the far child of a node near the
// root), and an idle lane can traverse it at once.
//
// Exactness.  The reference's closest-hit record is NOT the smallest t over all triangles: a
// node is skipped when its box lies beyond the closest hit so far, boxes and triangle tests round differently, and a triangle a few ulp
// closer than the record can sit in a skipped node (ten of 2^21 box rays on the atrium tree differ when parts are merged by t).  The
// record is a function of the traversal's HISTORY, so a helper's hit counts only where the history is provably the lone ray's:
//   (1) a helper that traverses subtree e with bound B' and finds NOTHING proves that the lone ray, arriving at e with any bound
//       b <= B', finds nothing either (with no hit the helper's bound stays B': the lone ray visits a subset of its nodes, in the same
//       order -- near / far is decided by the entry distances, which tmax does not enter -- and accepts nothing the helper would not);
//   (2) a helper that found a hit has computed exactly what the lone ray computes IF the lone ray arrives at e with b == B' bit for
//       bit, i.e. if nothing that comes before e in the visiting order has produced a closer hit.
// A lane gives away entries from the bottom of its stack, one per look, slot by slot upwards (slot 1 is visited last); the slots it has
// given away form the "dead zone" under its live entries, and when it has finished what it kept it settles them from the top down,
// which is the lone ray's order: a slot whose helper found nothing is dropped (1); a helper with a hit is accepted only when every
// slot above it is settled, the lane itself has finished, and its bound still is the B' the helper started with (2) -- the hit is then
// the lone ray's next hit (t < B': no tie to break); a helper whose donor's bound has moved gives up at once (its own helpers notice
// and follow), the slot is marked, and the donor traverses that entry itself when its turn comes, with the true bound (a closer hit
// exists by then, so that is usually over at the entry's root).  By induction over the helpers of helpers every lane ends with the
// lone ray's result for its part, and the ray's owner writes the record.
// Any-hit launches (the record is the FIRST hit in visiting order, and the ray ends there) follow the same rules: no lane holds a hit
// while it is on its way, so every bound is the ray's own tmax; a helper's hit is the ray's record exactly when everything before its
// entry has finished without one -- the acceptance rule above --, and a lane that has a hit, its own or one it took over, drops the
// slots it still has out instead of settling them: the lone ray would never have got there.
#pragma once

namespace ntr {

struct SplitState {
    int parent;       // helper: the lane this lane reports to (-1: not a helper)
    int slot;         // helper: the donor's stack slot its entry came from
    float bound0;     // helper: the donor's tmax at that moment, which it started with
    int epoch;        // bumped whenever this lane stops being a helper: a helper whose donor's epoch has moved on is an orphan
    int parentEpoch;  // helper: the donor's epoch when it took the entry
    int base;         // dead zone: stack slots 1 .. base have been given away (mem[base] holds the sentinel, the slots below their entries)
    int topEntry;     // the entry that was in slot `base` (whose place the sentinel takes)
    unsigned int masks;   // bit k: slot k is out with a helper; bit 16 + k: slot k came back unsettled, this lane traverses it itself
};
static_assert(LDS_DEPTH <= 16, "one mask bit per stack slot in LDS");

__device__ __forceinline__ void split_reset(SplitState& s)
{
    s.parent = -1; s.slot = 0; s.bound0 = 0.0f; s.parentEpoch = 0; s.base = 0; s.topEntry = kSentinel; s.masks = 0u;   // (epoch lives on)
}
__device__ __forceinline__ float split_shfl(float v, int l) { return __int_as_float(__builtin_amdgcn_ds_bpermute(l << 2, __float_as_int(v))); }
__device__ __forceinline__ int split_shfl(int v, int l) { return __builtin_amdgcn_ds_bpermute(l << 2, v); }
__device__ __forceinline__ float split_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int split_rank(unsigned long long m)   // set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}
// the dead zone loses its top slot: the slot below becomes the top (its entry moves to the register, the sentinel takes its place)
__device__ __forceinline__ void split_drop_top(SplitState& s, LaneStack& st)
{
    s.base--;
    if (s.base >= 1) { s.topEntry = st.lds[s.base * 64]; st.lds[s.base * 64] = kSentinel; }
}

enum { SPLIT_NONE = 0, SPLIT_NOHIT = 1, SPLIT_GIVEUP = 2, SPLIT_ACCEPT = 3 };

// What the helpers have to say, and what the lanes that have finished their own part do about their dead zone.
__device__ __forceinline__ void split_settle(SplitState& s, RayRegs& r, int& node, LaneStack& st, int& hitAddr, float& hitU, float& hitV,
                                             bool anyHit)
{
    const int lane = threadIdx.x & 63;
    if (__ballot(s.parent >= 0 || s.base > 0) == 0ull) return;
    // ---- helpers look at their donors -------------------------------------------------------------------------------------------
    int code = SPLIT_NONE;
    {
        const int P = s.parent >= 0 ? s.parent : lane;
        const float pBound = split_shfl(r.tmax, P);
        const int pNode = split_shfl(node, P), pEpoch = split_shfl(s.epoch, P);
        const unsigned int pMasks = (unsigned int)split_shfl((int)s.masks, P);
        if (s.parent >= 0) {
            const bool done = node == kSentinel && s.base == 0;
            const bool moved = __float_as_int(pBound) != __float_as_int(s.bound0);
            const unsigned int above = ((pMasks | (pMasks >> 16)) & 0xFFFFu) >> (s.slot + 1);   // unsettled slots of the donor above this one
            if (pEpoch != s.parentEpoch) { s.parent = -1; s.epoch++; split_reset(s); node = kSentinel; hitAddr = -1; }   // orphan: nobody waits for it
            else if (done && hitAddr < 0) code = SPLIT_NOHIT;
            else if (moved) code = SPLIT_GIVEUP;
            else if (done && pNode == kSentinel && above == 0u) code = SPLIT_ACCEPT;
        }
    }
    // ---- reports reach the donors one at a time (a donor may have several helpers) -----------------------------------------------
    for (;;) {
        const unsigned long long rep = __ballot(code != SPLIT_NONE);
        if (rep == 0ull) break;
        const int h = (int)__builtin_ctzll(rep);
        const int P = __builtin_amdgcn_readlane(s.parent, h), c = __builtin_amdgcn_readlane(code, h), k = __builtin_amdgcn_readlane(s.slot, h);
        const int aH = __builtin_amdgcn_readlane(hitAddr, h);
        const float tH = split_readlane(r.tmax, h), uH = split_readlane(hitU, h), vH = split_readlane(hitV, h);
        if (lane == P) {
            s.masks &= ~(1u << k);
            if (c == SPLIT_GIVEUP) s.masks |= 1u << (16 + k);
            else if (c == SPLIT_ACCEPT) { r.tmax = tH; hitAddr = aH; hitU = uH; hitV = vH; }
        }
        if (lane == h) {   // the helper's lane is free again
            code = SPLIT_NONE;
            s.epoch++;
            split_reset(s);
            node = kSentinel; hitAddr = -1;
        }
    }
    // ---- lanes that have finished what they kept settle their dead zone from the top: the lone ray's order -----------------------
    while (node == kSentinel && s.base > 0) {
        if (anyHit && hitAddr >= 0) {             // any hit: the ray has ended; what is still out is of no interest (its helpers become orphans)
            s.epoch++;
            s.base = 0; s.masks = 0u; s.topEntry = kSentinel;
            break;
        }
        const unsigned int out = 1u << s.base, redo = 1u << (16 + s.base);
        if (s.masks & out) break;                 // its helper is still out
        if (s.masks & redo) {                     // came back unsettled: traversed here, now, with the true bound
            s.masks &= ~redo;
            node = s.topEntry;
            split_drop_top(s, st);
            st.sp = s.base; st.tos = kSentinel;   // (an empty stack on top of the dead zone: the next push keeps the sentinel in slot `base`)
        } else {
            split_drop_top(s, st);                // settled: nothing there, or its hit was taken over
        }
    }
}

// The k-th idle lane takes the bottom live stack entry of the k-th lane that has one to give: the ray with the donor's tmax of this
// moment, an empty stack, the entry as its node.  Only entries in LDS slots are given (one mask bit per slot).
__device__ __forceinline__ void split_donate(SplitState& s, RayRegs& r, int& node, LaneStack& st, int rayIdx, int& hitAddr, float& hitU,
                                             float& hitV, bool& nice)
{
    const int lane = threadIdx.x & 63;
    const bool idle = rayIdx < 0 && s.parent < 0;
    const unsigned long long idleMask = __ballot(idle);
    if (idleMask == 0ull) return;
    const int bottom = 1 + s.base;                               // mem[0] is the sentinel under every stack
    const bool can = node != kSentinel && bottom < st.sp && bottom < LDS_DEPTH;
    const unsigned long long donors = __ballot(can);
    if (donors == 0ull) return;
    const int pairs = min((int)__popcll(idleMask), (int)__popcll(donors));
    const bool giving = can && split_rank(donors) < pairs;
    const bool taking = idle && split_rank(idleMask) < pairs;
    int give = kSentinel;
    if (giving) {
        give = st.lds[bottom * 64];
        if (s.base >= 1) st.lds[s.base * 64] = s.topEntry;       // the old top of the dead zone gets its entry back,
        st.lds[bottom * 64] = kSentinel;                         // the new one holds the sentinel the donor's pops end at
        s.topEntry = give;
        s.base = bottom;
        s.masks |= 1u << bottom;
    }
    int D = lane;   // the k-th taker's donor: the k-th set bit of `donors` (scalar loop, a few instructions per pair)
    {
        unsigned long long dm = donors, im = idleMask;
        for (int k = 0; k < pairs; k++) {
            const int d = (int)__builtin_ctzll(dm), i = (int)__builtin_ctzll(im);
            dm &= dm - 1ull; im &= im - 1ull;
            if (lane == i) D = d;
        }
    }
    const int gNode = split_shfl(give, D), gNice = split_shfl(nice ? 1 : 0, D), gSlot = split_shfl(s.base, D), gEpoch = split_shfl(s.epoch, D);
    const float ox = split_shfl(r.ox, D), oy = split_shfl(r.oy, D), oz = split_shfl(r.oz, D), tmin = split_shfl(r.tmin, D);
    const float dx = split_shfl(r.dx, D), dy = split_shfl(r.dy, D), dz = split_shfl(r.dz, D), tmax = split_shfl(r.tmax, D);
    const float rx = split_shfl(r.rx, D), ry = split_shfl(r.ry, D), rz = split_shfl(r.rz, D);
    if (taking) {
        r.ox = ox; r.oy = oy; r.oz = oz; r.tmin = tmin; r.dx = dx; r.dy = dy; r.dz = dz; r.tmax = tmax; r.rx = rx; r.ry = ry; r.rz = rz;
        nice = gNice != 0;
        node = gNode;
        NTR_STACK_RESET(st);
        hitAddr = -1; hitU = hitV = 0.0f;
        split_reset(s);
        s.parent = D; s.slot = gSlot; s.bound0 = tmax; s.parentEpoch = gEpoch;
    }
}

}  // namespace ntr
