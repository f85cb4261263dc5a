// RayBuffer.hpp -- ray batch container; same public surface as
// src/rt/ray/RayBuffer.hpp:38-194 (4 Buffers: rays 32 B, results 16 B, idToSlot, slotToID).
#pragma once
#include "Buffer.hpp"

namespace FW {

class RayBuffer {
public:
    RayBuffer(S32 n = 0, bool closestHit = true) : m_size(0), m_needClosestHit(closestHit) { resize(n); }

    S32  getSize() const { return m_size; }
    void resize(S32 n);  // never shrinks allocations (RayBuffer.cpp:38-52)

    void setRay(S32 slot, const Ray& ray) { setRay(slot, ray, slot); }
    void setRay(S32 slot, const Ray& ray, S32 id);
    void setResult(S32 slot, const RayResult& r) { getMutableResultForSlot(slot) = r; }

    const Ray&       getRayForSlot(S32 slot) const { return ((const Ray*)m_rays.getPtr())[slot]; }
    const Ray&       getRayForID(S32 id) const { return getRayForSlot(getSlotForID(id)); }
    const RayResult& getResultForSlot(S32 slot) const { return ((const RayResult*)m_results.getPtr())[slot]; }
    RayResult&       getMutableResultForSlot(S32 slot) { return ((RayResult*)m_results.getMutablePtr())[slot]; }
    const RayResult& getResultForID(S32 id) const { return getResultForSlot(getSlotForID(id)); }
    RayResult&       getMutableResultForID(S32 id) { return getMutableResultForSlot(getSlotForID(id)); }
    S32              getSlotForID(S32 id) const { return ((const S32*)m_IDToSlot.getPtr())[id]; }
    S32              getIDForSlot(S32 slot) const { return ((const S32*)m_slotToID.getPtr())[slot]; }

    // RayBuffer::mortonSort (RayBuffer.cpp:103-165): reorder by the 192-bit origin/direction key, on the device
    void mortonSort();

    void setNeedClosestHit(bool c) { m_needClosestHit = c; }
    bool getNeedClosestHit() const { return m_needClosestHit; }

    Buffer& getRayBuffer() { return m_rays; }
    Buffer& getResultBuffer() { return m_results; }
    Buffer& getIDToSlotBuffer() { return m_IDToSlot; }
    Buffer& getSlotToIDBuffer() { return m_slotToID; }

private:
    S32            m_size;
    mutable Buffer m_rays;
    mutable Buffer m_results;
    mutable Buffer m_IDToSlot;
    mutable Buffer m_slotToID;
    bool           m_needClosestHit;
};

}  // namespace FW
