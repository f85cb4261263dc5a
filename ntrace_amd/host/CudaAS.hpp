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

class CudaAS {
public:
    virtual ~CudaAS(void) {}
    virtual Buffer&   getNodeBuffer(void) = 0;
    virtual Buffer&   getTriWoopBuffer(void) = 0;
    virtual Buffer&   getTriIndexBuffer(void) = 0;
    virtual BVHLayout getLayout(void) const = 0;
    virtual void      serialize(std::ostream& out) = 0;
    // The reference's CudaAS::trace(RayBuffer&, Buffer& visibility) is its *CPU*
    // tracer (src/rt/cuda/CudaBVH.cpp:213-302).  This backend has no CPU trace
    // path by design; the CPU tracer is restated only as the test oracle (oracle/).
};

}  // namespace FW
