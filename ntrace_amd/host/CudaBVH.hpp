// CudaBVH.hpp -- flattened GPU BVH (src/rt/cuda/CudaBVH.hpp:95-330).
//
// BVHLayout_Compact (CudaBVH.hpp:42-56):
//   nodes   [innerOfs +  0] = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)
//           [innerOfs + 16] = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//           [innerOfs + 32] = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
//           [innerOfs + 48] = (c0 byte offset or ~c0.triOfs, c1 ..., splitBits, 0)
//   triWoop [triOfs*16 + 0/16/32] = woopZ, woopU, woopV ; leaf terminator = 0x80000000
//   triIndex[triOfs] = original triangle id (parallel to triWoop float4 index)
#pragma once
#include "CudaAS.hpp"
#include "bvh/BVH.hpp"

namespace FW {

class CudaBVH : public CudaAS {
public:
    enum { Align = 4096 };

    explicit CudaBVH(const BVH& bvh, BVHLayout layout);         // CudaBVH.cpp:60-103
    explicit CudaBVH(BVHLayout layout) : m_layout(layout), m_flags(0), m_flagsValid(false) {}
    explicit CudaBVH(std::istream& in);                          // CudaBVH.cpp:105-108
    virtual ~CudaBVH(void) {}

    virtual BVHLayout getLayout(void) const { return m_layout; }
    virtual Buffer&   getNodeBuffer(void) { return m_nodes; }
    virtual Buffer&   getTriWoopBuffer(void) { return m_triWoop; }
    virtual Buffer&   getTriIndexBuffer(void) { return m_triIndex; }
    virtual void      serialize(std::ostream& out);             // CudaBVH.cpp:118-125
    virtual void      trace(RayBuffer& rays, Buffer& visibility) { trace(rays, visibility, NULL); }  // CudaBVH.cpp:213-302 (host tracer)
    void              trace(RayBuffer& rays, Buffer& visibility, RayStats* stats);

    // Hint flags for ntr_trace_bvh (NTR_BVH_FINITE), computed once on the device.
    U32 getTraceFlags(void);
    void invalidateTraceFlags(void) { m_flagsValid = false; }

protected:
    void createCompact(const BVH& bvh, int nodeOffsetSizeDiv);  // CudaBVH.cpp:579-664
    static void woopify(const Vec3i* triVtxIndex, const Vec3f* vtxPos, S32 tri, Vec4f (&out)[3]);  // CudaBVH.cpp:668-687

    BVHLayout m_layout;
    Buffer    m_nodes;
    Buffer    m_triWoop;
    Buffer    m_triIndex;
    U32       m_flags;
    bool      m_flagsValid;
};

}  // namespace FW
