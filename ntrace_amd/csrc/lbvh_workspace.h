// lbvh_workspace.h -- host-side helpers of the LBVH builder's driver (ntr_lbvh_build, lbvh_kernels.hip): phase events, the per-device
// grow-only scratch workspace, and its 256-byte-aligned carver.  Included once, by lbvh_kernels.hip.
#pragma once
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

// Grow-only scratch memory of the builder, kept between builds: a rebuild per frame must not pay nine
// hipMalloc/hipFree pairs.  One workspace PER DEVICE (one caller per device at a time, as the rest of the
// C-ABI; host threads driving different devices never touch each other's workspace).  A workspace is only
// regrown after the device has drained, so a build still in flight on another stream keeps its memory.
struct Workspace {
    void* p = nullptr;
    size_t bytes = 0;
};
constexpr int kMaxDevices = 64;
Workspace g_ws[kMaxDevices];
std::mutex g_wsMu;

int workspace_reserve(size_t bytes, void** out)
{
    int dev = 0;
    NTR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return set_error(NTR_ERR_INVALID, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_wsMu);
    Workspace& w = g_ws[dev];
    if (w.p && w.bytes < bytes) {
        NTR_HIP(hipDeviceSynchronize());
        NTR_HIP(hipFree(w.p));
        w.p = nullptr; w.bytes = 0;
    }
    if (!w.p) {
        NTR_HIP(hipMalloc(&w.p, bytes));
        w.bytes = bytes;
    }
    *out = w.p;
    return NTR_OK;
}

int workspace_release()
{
    int dev = 0;
    NTR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return set_error(NTR_ERR_INVALID, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_wsMu);
    Workspace& w = g_ws[dev];
    if (w.p) {
        NTR_HIP(hipDeviceSynchronize());
        NTR_HIP(hipFree(w.p));
        w.p = nullptr; w.bytes = 0;
    }
    return NTR_OK;
}

struct DevMem {   // a temporary device allocation of the rare paths (hole compaction)
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
};

struct Carver {  // 256-byte aligned slices of the workspace
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};

}  // namespace
