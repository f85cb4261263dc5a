// Microbenchmark: random gather of 64-byte records, one record per lane per step.
//   mode 0: each lane issues 4 x dwordx4 for its own record            (64 distinct 16-B pieces / instr)
//   mode 1: quad-cooperative: instr j, lane l loads piece (l&3) of the record of lane (l&~3)+j
//           (4 adjacent lanes read 64 contiguous bytes), then a 4x4 quad transpose by DPP
//   mode 2: like 0 but 2 x dwordx4 (32 B records)  -- scaling check
// Build: hipcc -O3 --offload-arch=gfx950 gather64.hip -o gather64 ; run: ./gather64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __amdgpu_buffer_rsrc_t Rsrc;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld4(Rsrc r, int ofs) { return __builtin_amdgcn_raw_buffer_load_b128(r, ofs, 0, 0); }

template <int CTRL>
__device__ __forceinline__ unsigned int qperm(unsigned int v)
{
    return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
// quad_perm selecting lane k of the quad for every lane: ctrl = k | k<<2 | k<<4 | k<<6
#define QBCAST(k) ((k) | ((k) << 2) | ((k) << 4) | ((k) << 6))

template <int MODE>
__global__ __launch_bounds__(256) void gather(const void* table, unsigned int bytes, const int* idx0, int steps, unsigned int* out)
{
    const Rsrc rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(table), 0, (int)bytes, 0x00020000);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int rec = idx0[tid];
    const unsigned int mask = bytes / 64 - 1;
    unsigned int acc = 0;
    for (int s = 0; s < steps; s++) {
        u32x4 a, b, c, d;
        if (MODE == 0) {
            const int o = rec * 64;
            a = ld4(rs, o); b = ld4(rs, o + 16); c = ld4(rs, o + 32); d = ld4(rs, o + 48);
        } else if (MODE == 2) {
            const int o = rec * 64;
            a = ld4(rs, o); b = ld4(rs, o + 16); c = a; d = b;
        } else {
            // record offsets of the 4 lanes of my quad
            const int o = rec * 64;
            const int o0 = (int)qperm<QBCAST(0)>((unsigned)o), o1 = (int)qperm<QBCAST(1)>((unsigned)o);
            const int o2 = (int)qperm<QBCAST(2)>((unsigned)o), o3 = (int)qperm<QBCAST(3)>((unsigned)o);
            const int piece = (lane & 3) * 16;
            const u32x4 p0 = ld4(rs, o0 + piece), p1 = ld4(rs, o1 + piece), p2 = ld4(rs, o2 + piece), p3 = ld4(rs, o3 + piece);
            // lane q+k holds piece k of records q+0..q+3 in p0..p3.  I need pieces 0..3 of my record j=l&3:
            // piece k of record j sits in lane q+k, register p_j.
            const int j = lane & 3;
            u32x4 mine[4];
            for (int k = 0; k < 4; k++) {
                // value broadcast from lane q+k of each candidate register, then select by j
                u32x4 v;
                for (int e = 0; e < 4; e++) {
                    unsigned int c0, c1, c2, c3;
                    if (k == 0) { c0 = qperm<QBCAST(0)>(p0[e]); c1 = qperm<QBCAST(0)>(p1[e]); c2 = qperm<QBCAST(0)>(p2[e]); c3 = qperm<QBCAST(0)>(p3[e]); }
                    else if (k == 1) { c0 = qperm<QBCAST(1)>(p0[e]); c1 = qperm<QBCAST(1)>(p1[e]); c2 = qperm<QBCAST(1)>(p2[e]); c3 = qperm<QBCAST(1)>(p3[e]); }
                    else if (k == 2) { c0 = qperm<QBCAST(2)>(p0[e]); c1 = qperm<QBCAST(2)>(p1[e]); c2 = qperm<QBCAST(2)>(p2[e]); c3 = qperm<QBCAST(2)>(p3[e]); }
                    else { c0 = qperm<QBCAST(3)>(p0[e]); c1 = qperm<QBCAST(3)>(p1[e]); c2 = qperm<QBCAST(3)>(p2[e]); c3 = qperm<QBCAST(3)>(p3[e]); }
                    v[e] = (j == 0) ? c0 : (j == 1) ? c1 : (j == 2) ? c2 : c3;
                }
                mine[k] = v;
            }
            a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
        }
        const unsigned int h = a.x ^ b.y ^ c.z ^ d.w ^ a.w ^ d.x;
        acc += h;
        rec = (int)(((h ^ (unsigned)tid * 0x9E3779B9u) * 2654435761u + (unsigned)s * 40503u) & mask);   // dependent next index, per-thread
    }
    out[tid] = acc;
}

int main(int argc, char** argv)
{
    const size_t tableBytes = (size_t)(argc > 1 ? atol(argv[1]) : 16384) << 10;  // KB
    const int blocks = 8192, steps = 64;
    std::vector<unsigned int> h(tableBytes / 4);
    unsigned int x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    std::vector<int> idx(blocks * 256);
    for (auto& v : idx) { x = x * 1664525u + 1013904223u; v = (x >> 8) % (tableBytes / 64); }
    void* d_t; int* d_i; unsigned int* d_o;
    hipMalloc(&d_t, tableBytes); hipMalloc(&d_i, idx.size() * 4); hipMalloc(&d_o, idx.size() * 4);
    hipMemcpy(d_t, h.data(), tableBytes, hipMemcpyHostToDevice);
    hipMemcpy(d_i, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned int> o0(idx.size()), o1(idx.size());
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, d_t, (unsigned)tableBytes, d_i, steps, d_o);
            if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, d_t, (unsigned)tableBytes, d_i, steps, d_o);
            if (mode == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, d_t, (unsigned)tableBytes, d_i, steps, d_o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        if (mode == 0) hipMemcpy(o0.data(), d_o, o0.size() * 4, hipMemcpyDeviceToHost);
        if (mode == 1) hipMemcpy(o1.data(), d_o, o1.size() * 4, hipMemcpyDeviceToHost);
        const double recs = (double)blocks * 256 * steps;
        printf("table %zu KB mode %d: %.3f ms  %.1f Grec/s  %.2f TB/s (64B recs)\n", tableBytes >> 10, mode, best, recs / best / 1e6, recs * 64 / best / 1e9);
    }
    size_t bad = 0; for (size_t i = 0; i < o0.size(); i++) bad += o0[i] != o1[i];
    printf("mode0 vs mode1 mismatches: %zu\n", bad);
    return 0;
}
