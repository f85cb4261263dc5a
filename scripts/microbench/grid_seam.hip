// Microbenchmark: what does a phase seam cost inside ONE launch (a grid barrier) against a kernel boundary?  The question behind
// "run the small-scene LBVH build as a single persistent launch" (ten dependent, latency-bound phases of 4-58 us on 262 k triangles).
// P dependent phases; in every phase each workgroup reads the 4 KB another workgroup wrote in the previous phase (so the seam has to make
// other CUs' / other XCDs' stores visible) and writes its own 4 KB:
//   launches     P plain kernels on one stream
//   counter      one kernel, seams = one monotonic counter (lane 0: release fence, atomic add, relaxed sc1 poll + s_sleep, acquire fence)
//   xcd          one kernel, seams = per-XCD counters -> top counter -> per-XCD generation words (census of the XCD populations first)
// Every spin is bounded (give-up word); the result of the last phase is checked against the host's arithmetic (a stale read shows).
// Build: hipcc -O3 --offload-arch=gfx950 grid_seam.hip -o grid_seam ; run: ./grid_seam [wgs=256] [phases=10]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int THREADS = 256;
constexpr int WORDS = 1024;   // 4 KB per workgroup and phase
constexpr unsigned SPIN_LIMIT = 4000000u;

__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void phase_body(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int wg, int wgs, int phase)
{
    const int from = (wg * 7 + phase * 13 + 1) % wgs;   // another workgroup's block of the previous phase
    for (int i = threadIdx.x; i < WORDS; i += THREADS) dst[(size_t)wg * WORDS + i] = src[(size_t)from * WORDS + i] * 2654435761u + (unsigned)(wg + phase);
}

__global__ __launch_bounds__(THREADS) void phase_kernel(const unsigned* src, unsigned* dst, int wgs, int phase)
{
    phase_body(src, dst, blockIdx.x, wgs, phase);
}

// ---- seam 1: one counter ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool barrier_counter(unsigned* counter, unsigned target, unsigned* giveup)
{
    __syncthreads();   // every wave's stores are issued ...
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // ... and written back before the arrival
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_rlx(counter) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { atomicExch(giveup, 1u); ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(THREADS) void fused_counter(unsigned* a, unsigned* b, int wgs, int phases, unsigned* sync)
{
    for (int p = 0; p < phases; p++) {
        phase_body((p & 1) ? b : a, (p & 1) ? a : b, blockIdx.x, wgs, p);
        if (p + 1 < phases && !barrier_counter(sync, (unsigned)(wgs * (p + 1)), sync + 32)) return;
    }
}

// ---- seam 2: XCD-hierarchical ------------------------------------------------------------------------------------------------------
// sync layout (words, 32 apart = own 128-byte lines): [0] top counter, [32] give-up, [64 + 32 x] census of XCD x, [64 + 32 (8 + x)] arrivals of
// XCD x, [64 + 32 (16 + x)] generation of XCD x
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID, 4 bits

__device__ __forceinline__ bool barrier_xcd(unsigned* sync, unsigned xcc, unsigned population, unsigned numXcc, unsigned epoch)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        unsigned* arrivals = sync + 64 + 32 * (8 + xcc);
        unsigned* gen = sync + 64 + 32 * (16 + xcc);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        if (old + 1 == population * epoch) {   // the XCD's last arriver of this epoch: goes to the top, then opens its XCD
            __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_rlx(sync) < numXcc * epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { atomicExch(sync + 32, 2u); ok = false; break; }
            }
            __hip_atomic_store(gen, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_rlx(gen) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { atomicExch(sync + 32, 3u); ok = false; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(THREADS) void fused_xcd(unsigned* a, unsigned* b, int wgs, int phases, unsigned* sync)
{
    __shared__ unsigned s_pop, s_num;
    const unsigned xcc = xcc_id();
    // census: how many workgroups of this launch sit on each XCD (no placement assumption), behind one plain counter barrier
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sync + 64 + 32 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!barrier_counter(sync + 16, (unsigned)wgs, sync + 32)) return;
    if (threadIdx.x == 0) {
        unsigned num = 0;
        for (int x = 0; x < 8; x++) num += ld_rlx(sync + 64 + 32 * x) != 0u;
        s_pop = ld_rlx(sync + 64 + 32 * xcc);
        s_num = num;
    }
    __syncthreads();
    const unsigned pop = s_pop, num = s_num;
    for (int p = 0; p < phases; p++) {
        phase_body((p & 1) ? b : a, (p & 1) ? a : b, blockIdx.x, wgs, p);
        if (p + 1 < phases && !barrier_xcd(sync, xcc, pop, num, (unsigned)(p + 1))) return;
    }
}

static void host_expected(std::vector<unsigned>& a, std::vector<unsigned>& b, int wgs, int phases)
{
    for (int p = 0; p < phases; p++) {
        std::vector<unsigned>& src = (p & 1) ? b : a;
        std::vector<unsigned>& dst = (p & 1) ? a : b;
        for (int wg = 0; wg < wgs; wg++) {
            const int from = (wg * 7 + p * 13 + 1) % wgs;
            for (int i = 0; i < WORDS; i++) dst[(size_t)wg * WORDS + i] = src[(size_t)from * WORDS + i] * 2654435761u + (unsigned)(wg + p);
        }
    }
}

int main(int argc, char** argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const int phases = argc > 2 ? atoi(argv[2]) : 10;
    const size_t n = (size_t)wgs * WORDS;
    std::vector<unsigned> h0(n), ea, eb;
    unsigned x = 99;
    for (auto& v : h0) { x = x * 1664525u + 1013904223u; v = x; }
    ea = h0; eb.assign(n, 0u);
    host_expected(ea, eb, wgs, phases);
    const std::vector<unsigned>& expect = (phases & 1) ? eb : ea;   // the last phase p = phases - 1 wrote b if p is even
    unsigned *d_a, *d_b, *d_sync;
    hipMalloc(&d_a, n * 4); hipMalloc(&d_b, n * 4); hipMalloc(&d_sync, 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned> got(n);
    auto check = [&](const char* name, float us) {
        hipMemcpy(got.data(), (phases & 1) ? d_b : d_a, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) bad += got[i] != expect[i];
        unsigned giveup = 0;
        hipMemcpy(&giveup, d_sync + 32, 4, hipMemcpyDeviceToHost);
        printf("{\"mode\": \"%s\", \"wgs\": %d, \"phases\": %d, \"us_total\": %.2f, \"us_per_phase\": %.2f, \"wrong_words\": %zu, \"gave_up\": %u}\n", name, wgs, phases, us,
               us / phases, bad, giveup);
    };
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            hipMemcpy(d_a, h0.data(), n * 4, hipMemcpyHostToDevice);
            hipMemset(d_b, 0, n * 4);
            hipMemset(d_sync, 0, 4096 * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) {
                for (int p = 0; p < phases; p++)
                    hipLaunchKernelGGL(phase_kernel, dim3(wgs), dim3(THREADS), 0, 0, (p & 1) ? d_b : d_a, (p & 1) ? d_a : d_b, wgs, p);
            } else if (mode == 1) {
                hipLaunchKernelGGL(fused_counter, dim3(wgs), dim3(THREADS), 0, 0, d_a, d_b, wgs, phases, d_sync);
            } else {
                hipLaunchKernelGGL(fused_xcd, dim3(wgs), dim3(THREADS), 0, 0, d_a, d_b, wgs, phases, d_sync);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        check(mode == 0 ? "launches" : mode == 1 ? "counter" : "xcd", best * 1e3f);
    }
    // one phase alone: what a phase costs without any seam
    {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(phase_kernel, dim3(wgs), dim3(THREADS), 0, 0, d_a, d_b, wgs, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("{\"mode\": \"one_phase_alone\", \"wgs\": %d, \"us_total\": %.2f}\n", wgs, best * 1e3f);
    }
    return 0;
}
