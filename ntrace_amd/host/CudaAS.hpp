// CudaAS.hpp -- acceleration-structure interface (src/rt/cuda/CudaAS.hpp:20-64).
// BVHLayout and KernelConfig keep the reference's names
// (src/rt/kernels/CudaTracerKernels.hpp:52-75) and the C-ABI's values.
#pragma once
#include <iosfwd>

#include "Buffer.hpp"
#include "RayBuffer.hpp"
#include "ntrace_amd.h"

namespace FW {

enum BVHLayout {
    BVHLayout_AOS_AOS = NTR_BVHLayout_AOS_AOS,
    BVHLayout_AOS_SOA = NTR_BVHLayout_AOS_SOA,
    BVHLayout_SOA_AOS = NTR_BVHLayout_SOA_AOS,
    BVHLayout_SOA_SOA = NTR_BVHLayout_SOA_SOA,
    BVHLayout_Compact = NTR_BVHLayout_Compact,
    BVHLayout_Compact2 = NTR_BVHLayout_Compact2,
    BVHLayout_CPU = NTR_BVHLayout_CPU,
    BVHLayout_Max = NTR_BVHLayout_Max
};

typedef NtrKernelConfig KernelConfig;

// Traversal counters of the host tracer (src/rt/bvh/BVH.hpp:44-70; filled at CudaBVH.cpp:746-749, 1107-1111).
struct RayStats {
    RayStats() { clear(); }
    void clear() { numRays = numTriangleTests = numNodeTests = numTreelets = 0; }
    S32 numRays;
    S32 numTriangleTests;
    S32 numNodeTests;
    S32 numTreelets;
};

class CudaAS {
public:
    virtual ~CudaAS(void) {}
    virtual Buffer&   getNodeBuffer(void) = 0;
    virtual Buffer&   getTriWoopBuffer(void) = 0;
    virtual Buffer&   getTriIndexBuffer(void) = 0;
    virtual BVHLayout getLayout(void) const = 0;
    virtual void      serialize(std::ostream& out) = 0;
    // The reference's HOST tracer (src/rt/cuda/CudaAS.hpp:62, CudaBVH.cpp:213-302): traces `rays` on the CPU
    // copy of the buffers and sets visibility[id] = 1 for every triangle hit (when `visibility` is not empty).
    // Never a fallback of the device tracer: CudaBVHTracer::traceBatch fails without a HIP device.
    virtual void      trace(RayBuffer& rays, Buffer& visibility) = 0;
};

}  // namespace FW
