#include "RayBuffer.hpp"

#include "ntrace_amd.h"

namespace FW {

void RayBuffer::resize(S32 n)
{
    if (n < 0) fail("RayBuffer: negative size");
    if (n < m_size) {  // RayBuffer.cpp:41-45: shrinking keeps the allocations
        m_size = n;
        return;
    }
    m_size = n;
    m_rays.resize((S64)n * sizeof(Ray));
    m_results.resize((S64)n * sizeof(RayResult));
    m_IDToSlot.resize((S64)n * sizeof(S32));
    m_slotToID.resize((S64)n * sizeof(S32));
}

void RayBuffer::setRay(S32 slot, const Ray& ray, S32 id)
{
    ((Ray*)m_rays.getMutablePtr())[slot] = ray;
    ((S32*)m_IDToSlot.getMutablePtr())[id] = slot;
    ((S32*)m_slotToID.getMutablePtr())[slot] = id;
}

void RayBuffer::mortonSort()
{
    if (m_size == 0) return;
    Buffer oldRayBuffer(getRayBuffer());  // copies, like the reference's temporaries (RayBuffer.cpp:116-117)
    Buffer oldSlotToIDBuffer(getSlotToIDBuffer());
    int rc = ntr_ray_morton_sort(m_size, (const NtrRay*)oldRayBuffer.getCudaPtr(), (const int32_t*)oldSlotToIDBuffer.getCudaPtr(),
                                 (NtrRay*)getRayBuffer().getMutableCudaPtr(), (int32_t*)getIDToSlotBuffer().getMutableCudaPtr(),
                                 (int32_t*)getSlotToIDBuffer().getMutableCudaPtr(), NULL, NULL);
    if (rc != NTR_OK) fail("RayBuffer::mortonSort: %s", ntr_last_error());
}

}  // namespace FW
