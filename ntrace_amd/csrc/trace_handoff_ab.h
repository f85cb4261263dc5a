// trace_handoff_ab.h -- tail hand-off of the mini-pool waves (round 4).  A/B build only (-DNTR_AB, libntrace_amd_ab.so): measured
// 5-10 % SLOWER than leaving the tails alone on every batch it targets (EXPERIMENTS.md, "tail hand-off"), so the product library
// carries none of it.  Included by trace_kernels.hip inside namespace ntr, after the traversal loops; tests/test_handoff_gpu.py
// runs it against the A/B library.
#pragma once

// ---- tail hand-off (round 4): continuation queue of the mini-pool waves ------------------------------------------------------------
// A pool wave whose own rays are all started and of which fewer than T are still live keeps a whole wave slot busy for a handful of
// lanes (lane utilisation 0.15 on the 10 M-triangle tree, profiles/r03zz_courtyard10m_pmc_summary.json).  Such a wave now either FILLS
// its free lanes with continuations other waves left in a queue, or -- while enough waves are still running to pick them up -- APPENDS
// its own live rays to the queue and exits.  A continuation is the ray's complete traversal state (current node, shrunken tmax, hit so
// far, stack), so the ray goes on exactly where it stood: its visiting order, and with it its hit record, cannot change.
// Model first (scripts/studies/tail_handoff_model.py, profiles/r04_tail_handoff_model_*.jsonl): 1.8-3.4x fewer wave-iterations for K = 4.
//
// Queue: NTR_CONT_SHARDS independent shards (a wave uses shard = its ordinal % shards: the counters of one shard see 1/64 of the
// traffic).  Shard control line (128 B): [0] reserved = slots producers took, [1] popped = slots consumers claimed, [2] exited = waves
// of the shard that are gone.  Slot (128 B): [0] rayIdx -- doubling as the ready flag, -1 = empty --, node, tmax, hitAddr, hitU, hitV, sp,
// tos, then up to CONT_STACK stack entries.  Every access is an agent-scope relaxed atomic (sc1: served by the coherent level, the
// per-XCD L2s are not coherent with each other); a producer lane drains its stores (s_waitcnt vmcnt(0)) before it sets its slot's flag, a
// consumer lane polls its slot's flag before it loads the slot (MI355X_MICROARCH, inter-workgroup visibility: sc1 both sides).
//   producer: reserved += n (one atomic per wave); slots beyond the shard's capacity are VOID: the lane keeps its ray.
//   consumer: CAS on popped, never beyond reserved (a claimed slot has a producer that will fill it without waiting for anyone).
//   exit:     exited += 1 AFTER the wave's last reservation; the wave that completes its shard finds every reservation made and drains
//             what nobody claimed.  No wave ever waits for a wave that could be waiting for it.
// The counters are cleared by a kernel before the launch; slots are returned to -1 by their consumer.
static constexpr int CONT_STACK = NTR_CONT_SLOT_WORDS - 8;

struct ContShard {
    unsigned int* ctl;            // this wave's shard
    unsigned long long* slots;    // its slots, as 8-byte words
    int capacity;                 // slots of the shard
    int waves;                    // waves of the launch that use the shard
};

__device__ __forceinline__ unsigned int cont_ld(const unsigned int* a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long cont_ld64(const unsigned long long* a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cont_st64(unsigned long long* a, unsigned int lo, unsigned int hi)
{
    __hip_atomic_store(a, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A lane's traversal state, as the out-of-line hand-off routines see it (the kernel keeps these in registers; it packs them only around
// the rare calls, so the cold code costs the hot loop no register).
struct LaneState {
    RayRegs r;
    int rayIdx, node, hitAddr, sp, tos, nice;
    float hitU, hitV;
};

// Appends the rays of the lanes in `mask` to the shard and empties those lanes (a lane whose slot lies beyond the shard's capacity keeps its ray).
__device__ __noinline__ void cont_produce(unsigned int* ctl, unsigned long long* slots, int capacity, unsigned long long mask, LaneState& ls, lds_int* lds, int* spill)
{
    const int n = __popcll(mask);
    unsigned int base = 0;
    if (threadIdx.x == 0) base = __hip_atomic_fetch_add(ctl + 0, (unsigned int)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = __builtin_amdgcn_readfirstlane(base);
    const bool mine = (mask >> threadIdx.x) & 1ull;
    const unsigned int idx = base + (unsigned int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
    if (mine && idx < (unsigned int)capacity) {
        unsigned long long* slot = slots + (size_t)idx * (NTR_CONT_SLOT_WORDS / 2);
        cont_st64(slot + 1, __float_as_uint(ls.r.tmax), (unsigned int)ls.hitAddr);
        cont_st64(slot + 2, __float_as_uint(ls.hitU), __float_as_uint(ls.hitV));
        cont_st64(slot + 3, (unsigned int)ls.sp, (unsigned int)ls.tos);
        for (int i = 0; i < ls.sp; i += 2) {
            const int e0 = i < LDS_DEPTH ? lds[i * 64] : spill[i - LDS_DEPTH];
            const int e1 = (i + 1 < ls.sp) ? ((i + 1) < LDS_DEPTH ? lds[(i + 1) * 64] : spill[i + 1 - LDS_DEPTH]) : 0;
            cont_st64(slot + 4 + (i >> 1), (unsigned int)e0, (unsigned int)e1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the slot has reached the coherent level before its flag says so
        cont_st64(slot + 0, (unsigned int)ls.rayIdx, (unsigned int)ls.node);
        ls.rayIdx = -1;
        ls.node = kSentinel;
        ls.sp = 0;
        ls.tos = kSentinel;
    }
}

// Fills up to `want` empty lanes (rayIdx < 0) with continuations of the shard.  Returns the number of lanes filled.
__device__ __noinline__ int cont_consume(unsigned int* ctl, unsigned long long* slots, int capacity, const NtrRay* rays, uint32_t bvhFlags, unsigned int* status,
                                         int want, LaneState& ls, lds_int* lds, int* spill)
{
    unsigned int base = 0;
    int take = 0;
    if (threadIdx.x == 0) {
        for (int tries = 0; tries < 8; tries++) {
            const unsigned int pp = cont_ld(ctl + 1);
            const unsigned int rr = cont_ld(ctl + 0);   // read after popped: reserved only grows, so rr - pp never overstates what a claim from pp may take
            const int avail = (int)(rr - pp);
            if (avail <= 0) break;
            const int t = min(want, avail);
            unsigned int expected = pp;
            if (__hip_atomic_compare_exchange_strong(ctl + 1, &expected, pp + (unsigned int)t, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                base = pp;
                take = t;
                break;
            }
        }
    }
    base = __builtin_amdgcn_readfirstlane(base);
    take = __builtin_amdgcn_readfirstlane(take);
    if (take == 0) return 0;
    const unsigned long long empty = __ballot(ls.rayIdx < 0);
    const int prefix = __builtin_amdgcn_mbcnt_hi((unsigned)(empty >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)empty, 0));
    const unsigned int idx = base + (unsigned int)prefix;
    const bool mine = ls.rayIdx < 0 && prefix < take && idx < (unsigned int)capacity;   // (a void slot holds nothing: its producer kept the ray)
    if (mine) {
        unsigned long long* slot = slots + (size_t)idx * (NTR_CONT_SLOT_WORDS / 2);
        unsigned long long w0;
        unsigned int spins = 0;
        while ((int)(unsigned int)(w0 = cont_ld64(slot + 0)) < 0) {     // the producer reserved this slot before the claim: it is on its way
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 22)) { atomicOr(status, NTR_STATUS_HANDOFF_TIMEOUT); break; }
        }
        if ((int)(unsigned int)w0 >= 0) {
            const unsigned long long w1 = cont_ld64(slot + 1), w2 = cont_ld64(slot + 2), w3 = cont_ld64(slot + 3);
            ls.rayIdx = (int)(unsigned int)w0;
            load_ray(rays, ls.rayIdx, ls.r);
            ls.nice = ray_is_nice(ls.r, bvhFlags) ? 1 : 0;
            ls.node = (int)(unsigned int)(w0 >> 32);
            ls.r.tmax = __uint_as_float((unsigned int)w1);
            ls.hitAddr = (int)(unsigned int)(w1 >> 32);
            ls.hitU = __uint_as_float((unsigned int)w2);
            ls.hitV = __uint_as_float((unsigned int)(w2 >> 32));
            ls.sp = (int)(unsigned int)w3;
            ls.tos = (int)(unsigned int)(w3 >> 32);
            for (int i = 0; i < ls.sp; i += 2) {
                const unsigned long long e = cont_ld64(slot + 4 + (i >> 1));
                if (i < LDS_DEPTH) lds[i * 64] = (int)(unsigned int)e; else spill[i - LDS_DEPTH] = (int)(unsigned int)e;
                if (i + 1 < ls.sp) { if (i + 1 < LDS_DEPTH) lds[(i + 1) * 64] = (int)(unsigned int)(e >> 32); else spill[i + 1 - LDS_DEPTH] = (int)(unsigned int)(e >> 32); }
            }
            cont_st64(slot + 0, 0xFFFFFFFFu, 0u);   // the slot is free again (for the next launch: a slot is used once per launch)
        }
    }
    return __popcll(__ballot(mine && ls.rayIdx >= 0));
}

// packs / unpacks the register state around the out-of-line calls
#define NTR_LANE_PACK(ls) do { (ls).r = r; (ls).rayIdx = rayIdx; (ls).node = node; (ls).hitAddr = hitAddr; (ls).sp = st.sp; (ls).tos = st.tos; \
                               (ls).nice = nice ? 1 : 0; (ls).hitU = hitU; (ls).hitV = hitV; } while (0)
#define NTR_LANE_UNPACK(ls) do { r = (ls).r; rayIdx = (ls).rayIdx; node = (ls).node; hitAddr = (ls).hitAddr; st.sp = (ls).sp; st.tos = (ls).tos; \
                                 nice = (ls).nice != 0; hitU = (ls).hitU; hitV = (ls).hitV; } while (0)

