#include "HLBVHBuilder.hpp"

namespace FW {

HLBVHBuilder::HLBVHBuilder(Scene* scene, const Platform& platform, HLBVHParams params)
    : CudaBVH(BVHLayout_Compact), m_scene(scene), m_platform(platform), m_params(params), m_gpuTime(0.0f), m_nodesCnt(0), m_leafs(0)
{
    if (params.hlbvh && params.hlbvhBits != 10)
        fail("HLBVHBuilder: the HLBVH top-level SAH path (buildHLBVH) is not part of this backend; use hlbvh=false");
    buildLBVH();
}

// HLBVHBuilder::buildLBVH (HLBVHBuilder.cpp:451-593) through the C-ABI.
void HLBVHBuilder::buildLBVH(void)
{
    const int triCnt = m_scene->getNumTriangles();
    int64_t capN, capW, capI;
    if (ntr_lbvh_capacity(triCnt, &capN, &capW, &capI) != NTR_OK) fail("HLBVHBuilder: %s", ntr_last_error());
    m_nodes.resizeDiscard(capN);
    m_triWoop.resizeDiscard(capW);
    m_triIndex.resizeDiscard(capI);
    Vec3f lo, hi;
    m_scene->getBBox(lo, hi);
    const float mn[3] = {lo.x, lo.y, lo.z}, mx[3] = {hi.x, hi.y, hi.z};
    int rc = ntr_lbvh_build(triCnt, (const int32_t*)m_scene->getTriVtxIndexBuffer().getCudaPtr(), m_scene->getNumVertices(),
                            (const float*)m_scene->getVtxPosBuffer().getCudaPtr(), mn, mx, m_params.leafSize, m_params.epsilon,
                            m_nodes.getMutableCudaPtr(), capN, m_triWoop.getMutableCudaPtr(), capW,
                            (int32_t*)m_triIndex.getMutableCudaPtr(), capI, &m_result, NULL);
    if (rc != NTR_OK) fail("HLBVHBuilder: %s", ntr_last_error());
    // exact sizes (HLBVHBuilder.cpp:382-386)
    m_nodes.resize(m_result.nodesBytes);
    m_triWoop.resize(m_result.triWoopBytes);
    m_triIndex.resize(m_result.triIndexBytes);
    m_gpuTime = m_result.seconds;
    m_nodesCnt = (U32)m_result.numNodes;
    m_leafs = (U32)m_result.numLeaves;
    invalidateTraceFlags();
}

}  // namespace FW
