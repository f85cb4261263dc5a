// device_scratch.h -- a grow-only scratch allocation per device, kept between calls (the LBVH builder's workspace, the ray sort's
// temporaries): a rebuild or a sort per frame must not pay hipMalloc / hipFree pairs, each of which synchronises the device.
// One caller per device at a time, as everywhere in the C-ABI; host threads driving different devices never touch each other's memory.
// A pool is only regrown after the device has drained, so work still in flight on another stream keeps its memory.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "ntr_internal.h"

namespace ntr {

class DeviceScratchPool {
public:
    int reserve(size_t bytes, void** out)
    {
        int dev = 0;
        NTR_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= kMaxDevices) return set_error(NTR_ERR_INVALID, "device index %d out of range", dev);
        std::lock_guard<std::mutex> lk(mu_);
        Slot& w = slots_[dev];
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
    int release()   // the current device's allocation (waits for the device first)
    {
        int dev = 0;
        NTR_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= kMaxDevices) return set_error(NTR_ERR_INVALID, "device index %d out of range", dev);
        std::lock_guard<std::mutex> lk(mu_);
        Slot& w = slots_[dev];
        if (w.p) {
            NTR_HIP(hipDeviceSynchronize());
            NTR_HIP(hipFree(w.p));
            w.p = nullptr; w.bytes = 0;
        }
        return NTR_OK;
    }

private:
    static constexpr int kMaxDevices = 64;
    struct Slot { void* p = nullptr; size_t bytes = 0; };
    Slot slots_[kMaxDevices];
    std::mutex mu_;
};

}  // namespace ntr
