// Microbenchmark: the request-rate roof of dependent random fetches beyond the L2 as a function of the RECORD SIZE.  chase64.hip measured
// 56 G records/s for 64-byte records (and the same for 32): is the roof per request, per 64-byte sector or per 128-byte line?  The answer
// decides whether a wider node (two tree levels per fetch: 192 bytes) could shorten a divergent ray's chain of dependent fetches without
// paying for it in throughput (what comes next).  Every lane walks a chain of random records of REC bytes (REC / 16 loads of 16
// bytes, the next index a hash of the bytes loaded), records aligned to 64 bytes (a 192-byte record straddles two or three 128-byte lines).
// Build: hipcc -O3 --offload-arch=gfx950 chase_rec.hip -o chase_rec ; run: ./chase_rec [tableMB=768]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int VEC>   // 16-byte pieces per record
__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ table, unsigned int numRecs, unsigned int strideVec, int steps, unsigned int* out)
{
    const int tid = blockIdx.x * 64 + threadIdx.x;
    unsigned int rec = ((unsigned)tid * 2654435761u) % numRecs;
    unsigned int acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4* q = table + (size_t)rec * strideVec;
        uint4 v[VEC];
#pragma unroll
        for (int k = 0; k < VEC; k++) v[k] = q[k];
        unsigned int h = 0;
#pragma unroll
        for (int k = 0; k < VEC; k++) h ^= v[k].x + v[k].w * 31u;
        acc += h;
        rec = (((h ^ (unsigned)tid * 0x9E3779B9u) * 2654435761u + (unsigned)s * 40503u) >> 7) % numRecs;
    }
    out[tid] = acc;
}

template <int VEC>
static void run(const uint4* d_t, size_t tableBytes, unsigned int* d_o, hipEvent_t e0, hipEvent_t e1, int alignBytes)
{
    const int rec = VEC * 16;
    const unsigned int stride = (unsigned int)(((rec + alignBytes - 1) / alignBytes) * alignBytes);
    const unsigned int numRecs = (unsigned int)(tableBytes / stride) - 1u;
    const int steps = 192;
    for (int waves : {8192, 2048, 1024}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(chase<VEC>, dim3(waves), dim3(64), 0, 0, d_t, numRecs, stride / 16u, steps, d_o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double n = (double)waves * 64 * steps;
        printf("{\"record_bytes\": %d, \"stride\": %u, \"waves\": %d, \"ms\": %.3f, \"us_per_step\": %.3f, \"grecords_per_s\": %.2f, \"TBps\": %.2f}\n", rec, stride, waves, best,
               best * 1e3 / steps, n / best / 1e6, n * rec / best / 1e9);
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
    run<2>((const uint4*)d_t, tableBytes, d_o, e0, e1, 64);    // 32 B
    run<4>((const uint4*)d_t, tableBytes, d_o, e0, e1, 64);    // 64 B
    run<8>((const uint4*)d_t, tableBytes, d_o, e0, e1, 64);    // 128 B at 64-byte alignment (half of them straddle two lines)
    run<8>((const uint4*)d_t, tableBytes, d_o, e0, e1, 128);   // 128 B, line aligned
    run<12>((const uint4*)d_t, tableBytes, d_o, e0, e1, 64);   // 192 B
    run<16>((const uint4*)d_t, tableBytes, d_o, e0, e1, 128);  // 256 B, line aligned
    return 0;
}
