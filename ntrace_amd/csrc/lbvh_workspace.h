// lbvh_workspace.h -- host-side helpers of the LBVH builder's driver (ntr_lbvh_build, lbvh_kernels.hip): phase events, the per-device
// grow-only scratch workspace, and its 256-byte-aligned carver.  Included once, by lbvh_kernels.hip.
#pragma once
#include "device_scratch.h"
namespace {
// Phase boundaries are recorded as events on the stream and read back after ONE synchronisation at
// the end of the build, so the timed build has no host round trips inside it.
struct PhaseEvents {
    enum { N = 7 };
    hipEvent_t ev[N] = {};
    hipStream_t s;
    explicit PhaseEvents(hipStream_t st) : s(st) { for (auto& e : ev) (void)hipEventCreate(&e); }
    ~PhaseEvents() { for (auto& e : ev) (void)hipEventDestroy(e); }
    void mark(int i) { (void)hipEventRecord(ev[i], s); }
    float ms(int a, int b) { float v = 0; (void)hipEventElapsedTime(&v, ev[a], ev[b]); return v; }
};

// Grow-only scratch memory of the builder, kept between builds (device_scratch.h): a rebuild per frame must not pay nine
// hipMalloc/hipFree pairs.
ntr::DeviceScratchPool g_ws;
int workspace_reserve(size_t bytes, void** out) { return g_ws.reserve(bytes, out); }
int workspace_release() { return g_ws.release(); }

struct DevMem {   // a temporary device allocation of the rare paths (hole compaction)
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
};

struct Carver {  // 256-byte aligned slices of the workspace
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};

}  // namespace
