#include "CudaBVHTracer.hpp"

namespace FW {

CudaBVHTracer::CudaBVHTracer(void) : m_bvh(NULL), m_hint(NULL)
{
    m_scene = NULL;
    m_kernelConfig.bvhLayout = BVHLayout_Max;
    m_kernelConfig.blockWidth = m_kernelConfig.blockHeight = m_kernelConfig.usePersistentThreads = 0;
}

// CudaBVHTracer::setKernel (CudaBVHTracer.cpp:52-84).
void CudaBVHTracer::setKernel(const String& kernelName)
{
    if (m_kernelName == kernelName) return;
    m_kernelName = kernelName;
    if (ntr_query_config(kernelName.c_str(), &m_kernelConfig) != NTR_OK)
        fail("CudaBVHTracer: %s", ntr_last_error());
}

// CudaBVHTracer::traceBatch (CudaBVHTracer.cpp:88-168).
F32 CudaBVHTracer::traceBatch(RayBuffer& rays)
{
    return traceRange(rays, 0, rays.getSize());
}

F32 CudaBVHTracer::traceRange(RayBuffer& rays, S32 first, S32 count)
{
    if (first < 0 || count < 0 || first + count > rays.getSize()) fail("CudaBVHTracer: ray range out of bounds");
    int numRays = count;
    if (!numRays) return 0.0f;

    if (!m_bvh) fail("CudaBVHTracer: No BVH!");
    if (m_bvh->getLayout() != getDesiredBVHLayout()) fail("CudaBVHTracer: Incorrect BVH layout!");

    U32 flags = 0;
    if (CudaBVH* cb = dynamic_cast<CudaBVH*>(m_bvh)) flags = cb->getTraceFlags();

    float seconds = 0.0f;
    int rc = ntr_trace_bvh_hinted(m_kernelName.c_str(), numRays, rays.getNeedClosestHit() ? 0 : 1,
                           (const NtrRay*)rays.getRayBuffer().getCudaPtr() + first,
                           (NtrRayResult*)rays.getResultBuffer().getMutableCudaPtr() + first,
                           m_bvh->getNodeBuffer().getCudaPtr(), m_bvh->getNodeBuffer().getSize(),
                           m_bvh->getTriWoopBuffer().getCudaPtr(), m_bvh->getTriWoopBuffer().getSize(),
                           (const int32_t*)m_bvh->getTriIndexBuffer().getCudaPtr(), (int32_t)m_bvh->getLayout(),
                           flags, NULL, &seconds, m_hint);
    if (rc != NTR_OK) fail("CudaBVHTracer: %s", ntr_last_error());
    return seconds;
}

}  // namespace FW
