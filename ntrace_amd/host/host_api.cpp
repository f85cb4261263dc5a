// host_api.cpp -- C-ABI entry points that are pure host work (no device).
#include <new>

#include "CudaBVH.hpp"
#include "bvh/BVH.hpp"
#include "ntr_internal.h"

using namespace FW;

struct NtrHostBvh {
    Scene*   scene;
    CudaBVH* cbvh;
    BVH::Stats stats;
};

extern "C" {

int ntr_sah_build(int32_t numTris, const int32_t* triVtxIndex, int32_t numVerts, const float* vtxPos,
                  int32_t minLeafSize, int32_t maxLeafSize, NtrHostBvh** out)
{
    if (!out) return ntr::set_error(NTR_ERR_INVALID, "ntr_sah_build: null out");
    *out = nullptr;
    if (numTris < 0 || numVerts < 0 || (numTris && !triVtxIndex) || (numVerts && !vtxPos))
        return ntr::set_error(NTR_ERR_INVALID, "ntr_sah_build: bad geometry arguments");
    for (int64_t i = 0; i < (int64_t)numTris * 3; i++)
        if (triVtxIndex[i] < 0 || triVtxIndex[i] >= numVerts)
            return ntr::set_error(NTR_ERR_INVALID, "ntr_sah_build: vertex index out of range at triangle %lld", (long long)(i / 3));
    try {
        NtrHostBvh* h = new NtrHostBvh();
        h->scene = new Scene(numTris, (const Vec3i*)triVtxIndex, numVerts, (const Vec3f*)vtxPos);
        Platform platform("GPU");  // Renderer.cpp:88-89
        platform.setLeafPreferences(minLeafSize, maxLeafSize);
        BVH::BuildParams params;
        params.stats = &h->stats;
        BVH bvh(h->scene, platform, params);
        h->cbvh = new CudaBVH(bvh, BVHLayout_Compact);
        *out = h;
        return NTR_OK;
    } catch (const FatalError& e) {
        return ntr::set_error(NTR_ERR_INVALID, "%s", e.message.c_str());
    } catch (const std::bad_alloc&) {
        return ntr::set_error(NTR_ERR_NOMEM, "ntr_sah_build: out of host memory");
    }
}

int ntr_host_bvh_info(const NtrHostBvh* bvh, NtrHostBvhInfo* info)
{
    if (!bvh || !info) return ntr::set_error(NTR_ERR_INVALID, "ntr_host_bvh_info: null argument");
    CudaBVH* c = bvh->cbvh;
    info->nodes = c->getNodeBuffer().getPtr();
    info->nodesBytes = c->getNodeBuffer().getSize();
    info->triWoop = c->getTriWoopBuffer().getPtr();
    info->triWoopBytes = c->getTriWoopBuffer().getSize();
    info->triIndex = (const int32_t*)c->getTriIndexBuffer().getPtr();
    info->triIndexBytes = c->getTriIndexBuffer().getSize();
    info->layout = (int32_t)c->getLayout();
    info->numInnerNodes = bvh->stats.numInnerNodes;
    info->numLeafNodes = bvh->stats.numLeafNodes;
    info->maxDepth = bvh->stats.maxDepth;
    info->buildSeconds = bvh->stats.buildTime;
    return NTR_OK;
}

void ntr_host_bvh_free(NtrHostBvh* bvh)
{
    if (!bvh) return;
    delete bvh->cbvh;
    delete bvh->scene;
    delete bvh;
}

}  // extern "C"
