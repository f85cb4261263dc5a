// CudaBVHTracer.hpp -- BVH tracer launcher (src/rt/cuda/CudaBVHTracer.hpp:46-107).
// Same methods; runtime nvcc compilation, texrefs and g_config read-back are
// replaced by the C-ABI (ntr_query_config / ntr_trace_bvh).
#pragma once
#include "CudaBVH.hpp"
#include "CudaVirtualTracer.hpp"

namespace FW {

class CudaBVHTracer : public CudaVirtualTracer {
public:
    CudaBVHTracer(void);
    virtual ~CudaBVHTracer(void) {}

    virtual void      setMessageWindow(Window*) {}
    virtual void      setKernel(const String& kernelName);
    virtual BVHLayout getDesiredBVHLayout(void) const { return (BVHLayout)m_kernelConfig.bvhLayout; }
    virtual void      setBVH(CudaAS* bvh) { m_bvh = bvh; }
    virtual F32       traceBatch(RayBuffer& rays);
    // Multi-GPU extension (no counterpart in the reference): trace only the slots [first, first + count) of `rays` -- a rank's
    // screen-tile range of the primary batch (Renderer::setShard).
    F32               traceRange(RayBuffer& rays, S32 first, S32 count);
    // Dispatch hint for the next traceBatch / traceRange calls (ntr_trace_bvh_hinted; NULL = none).  No counterpart in the reference.
    void              setSchedHint(NtrSchedHint* hint) { m_hint = hint; }

    const KernelConfig& getKernelConfig(void) const { return m_kernelConfig; }

private:
    String       m_kernelName;
    KernelConfig m_kernelConfig;
    CudaAS*      m_bvh;
    NtrSchedHint* m_hint;
};

}  // namespace FW
