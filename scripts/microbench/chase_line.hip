// Microbenchmark for the round-5 idea "widen the fetch, not the step" (EXPERIMENTS.md): records of 64 bytes, two to a 128-byte line.
//   A  every step goes to a random record and loads its 64 bytes                                   (today's traversal)
//   B  the same, plus ONE 4-byte load of the line's other half with every step (result only summed)  (is the ride-along load free?)
//   C  every second step goes to the OTHER half of the line just visited (no prefetch)               (a child in the same line, found cold)
//   D  like C, with B's ride-along load on the first visit of a line                                 (the child's half already on its way)
//   E, F  like B, D with TWO ride-along loads, one per 32-byte sector of the other half
//   G-J   like C, the second step of a pair 128 B / 1 KB / 4 KB / 64 KB further instead of in the same line (does nearness beyond the line help?)
// Per variant: time per step and records/s at 8 192 / 2 048 / 1 024 waves of 64 lanes, 768 MB table.
// Build: hipcc -O3 --offload-arch=gfx950 chase_line.hip -o chase_line ; run: ./chase_line [tableMB=768]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// NEAR > 0 (with MODE 2): the second step of a pair goes to the first half of the line NEAR lines further instead of this line's other half
template <int MODE, int NEAR = 0>
__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ table, unsigned int numLines, int steps, unsigned int* out)
{
    const int tid = blockIdx.x * 64 + threadIdx.x;
    unsigned int line = ((unsigned)tid * 2654435761u) % numLines, half = 0;
    unsigned int acc = 0, side = 0;
    for (int s = 0; s < steps; s++) {
        const uint4* q = NEAR > 0 ? table + ((size_t)(line + (half ? NEAR : 0)) * 8) : table + ((size_t)line * 8 + half * 4);
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        if ((MODE == 1) || (MODE == 3 && half == 0)) side += reinterpret_cast<const unsigned int*>(table + ((size_t)line * 8 + (half ^ 1u) * 4))[0];
        if ((MODE == 4) || (MODE == 5 && half == 0)) {   // both 32-byte sectors of the other half
            const unsigned int* o = reinterpret_cast<const unsigned int*>(table + ((size_t)line * 8 + (half ^ 1u) * 4));
            side += o[0] + o[8];
        }
        const unsigned int h = a.x ^ b.y ^ c.z ^ d.w ^ a.w ^ d.x;
        acc += h;
        if ((MODE == 2 || MODE == 3 || MODE == 5) && half == 0) {
            half = 1;    // next: the other half of this line (address known only now: it depends on nothing loaded, but the walk is in order)
        } else {
            half = 0;
            line = (((h ^ (unsigned)tid * 0x9E3779B9u) * 2654435761u + (unsigned)s * 40503u) >> 7) % numLines;
        }
    }
    out[tid] = acc + side;
}

template <int MODE, int NEAR = 0>
static void run(const char* name, const uint4* d_t, size_t tableBytes, unsigned int* d_o, hipEvent_t e0, hipEvent_t e1)
{
    const unsigned int numLines = (unsigned int)(tableBytes / 128) - 1u - 4096u;
    const int steps = 192;
    for (int waves : {8192, 2048, 1024}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((chase<MODE, NEAR>), dim3(waves), dim3(64), 0, 0, d_t, numLines, steps, d_o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double n = (double)waves * 64 * steps;
        printf("{\"variant\": \"%s\", \"waves\": %d, \"ms\": %.3f, \"us_per_step\": %.3f, \"gsteps_per_s\": %.2f}\n", name, waves, best, best * 1e3 / steps, n / best / 1e6);
    }
}

int main(int argc, char** argv)
{
    const size_t tableBytes = (size_t)(argc > 1 ? atol(argv[1]) : 768) << 20;
    std::vector<unsigned int> h(tableBytes / 4);
    unsigned int x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    void* d_t; unsigned int* d_o;
    if (hipMalloc(&d_t, tableBytes) != hipSuccess || hipMalloc(&d_o, (size_t)8192 * 64 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemcpy(d_t, h.data(), tableBytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run<0>("A random 64 B per step", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    run<1>("B + 4-byte load of the line's other half", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    run<2>("C every second step: other half of the same line, cold", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    run<3>("D like C with the ride-along load on the first visit", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    run<4>("E like B with TWO 4-byte loads (both 32-byte sectors of the other half)", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    run<5>("F like D with the two ride-along loads", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    run<2, 1>("G every second step: the NEXT line (128 B further), cold", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    run<2, 8>("H every second step: 1 KB further, cold", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    run<2, 32>("I every second step: 4 KB further, cold", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    run<2, 512>("J every second step: 64 KB further, cold", (const uint4*)d_t, tableBytes, d_o, e0, e1);
    return 0;
}
