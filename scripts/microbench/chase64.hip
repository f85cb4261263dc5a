// Microbenchmark: the memory side of a divergent traversal, without the arithmetic.  Every active lane walks a dependent chain of random
// 64-byte records (4 x global_load_dwordx4, the next index is a hash of the bytes just loaded) -- what a ray does from node to node.
// Per table size (Infinity-Cache resident / HBM resident), per number of active lanes in a wave and per number of waves:
//   time per step (the latency a ray sees under that load) and records / s, bytes / s (the throughput roof of this access pattern).
// Build: hipcc -O3 --offload-arch=gfx950 chase64.hip -o chase64 ; run: ./chase64 [tableKB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ table, unsigned int mask, int steps, int activeLanes, unsigned int* out)
{
    const int tid = blockIdx.x * 64 + threadIdx.x;
    if ((int)threadIdx.x >= activeLanes) return;
    unsigned int rec = ((unsigned)tid * 2654435761u) & mask;
    unsigned int acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4* q = table + (size_t)rec * 4;
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        const unsigned int h = a.x ^ b.y ^ c.z ^ d.w ^ a.w ^ d.x;
        acc += h;
        rec = ((h ^ (unsigned)tid * 0x9E3779B9u) * 2654435761u + (unsigned)s * 40503u) & mask;
    }
    out[tid] = acc;
}

int main(int argc, char** argv)
{
    const size_t tableBytes = (size_t)(argc > 1 ? atol(argv[1]) : 786432) << 10;   // KB (a power of two keeps the mask exact)
    size_t recs = tableBytes / 64, pow2 = 1;
    while (pow2 * 2 <= recs) pow2 *= 2;
    const unsigned int mask = (unsigned int)(pow2 - 1);
    std::vector<unsigned int> h(tableBytes / 4);
    unsigned int x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    void* d_t; unsigned int* d_o;
    hipMalloc(&d_t, tableBytes); hipMalloc(&d_o, (size_t)65536 * 64 * 4);
    hipMemcpy(d_t, h.data(), tableBytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 256;
    printf("table %zu KB (%zu records addressed)\n", tableBytes >> 10, pow2);
    for (int waves : {32768, 8192, 7168, 2048, 1024, 256}) {
        for (int lanes : {64, 16, 4, 1}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(chase, dim3(waves), dim3(64), 0, 0, (const uint4*)d_t, mask, steps, lanes, d_o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double n = (double)waves * lanes * steps;
            const double rounds = waves > 8192 ? waves / 8192.0 : 1.0;   // (8 waves per SIMD are resident at once)
            printf("waves %6d lanes %2d: %8.3f ms  %6.3f us/step  %7.2f Grec/s  %6.2f TB/s\n", waves, lanes, best, best * 1e3 / steps / rounds, n / best / 1e6, n * 64 / best / 1e9);
        }
    }
    return 0;
}
