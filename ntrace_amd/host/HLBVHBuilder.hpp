// HLBVHBuilder.hpp -- GPU LBVH builder object (src/rt/bvh/HLBVH/HLBVHBuilder.hpp): a CudaBVH that
// builds itself on the device.  Only the plain-LBVH path (buildLBVH, taken for !hlbvh ||
// hlbvhBits == 10, HLBVHBuilder.cpp:44-47) is provided; buildHLBVH's SAH top level is out of scope.
#pragma once
#include "CudaBVH.hpp"
#include "Scene.hpp"
#include "bvh/Platform.hpp"

namespace FW {

struct HLBVHParams {  // HLBVHBuilder.hpp
    bool hlbvh;
    S32  hlbvhBits;
    S32  leafSize;
    F32  epsilon;
    HLBVHParams(void) : hlbvh(false), hlbvhBits(4), leafSize(8), epsilon(0.001f) {}
};

class HLBVHBuilder : public CudaBVH {
public:
    HLBVHBuilder(Scene* scene, const Platform& platform, HLBVHParams params);
    virtual ~HLBVHBuilder(void) {}

    F32  getGPUTime(void) const { return m_gpuTime; }
    void getStats(U32& nodes, U32& leaves, U32& nodeTop) const { nodes = m_nodesCnt; leaves = m_leafs; nodeTop = m_nodesCnt; }
    const NtrLbvhResult& getBuildResult(void) const { return m_result; }

private:
    void buildLBVH(void);

    Scene*        m_scene;
    Platform      m_platform;
    HLBVHParams   m_params;
    F32           m_gpuTime;
    U32           m_nodesCnt, m_leafs;
    NtrLbvhResult m_result;
};

}  // namespace FW
