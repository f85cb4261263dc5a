// host_api.cpp -- C-ABI entry points that are pure host work (no device).
#include <cstring>
#include <new>
#include <vector>

#include <chrono>
#include <exception>
#include <cstdio>
#include <cstdlib>
#include "CameraControls.hpp"
#include "CudaBVH.hpp"
#include "MeshWavefrontIO.hpp"
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
        const bool timing = std::getenv("NTR_SAH_TIMING") != NULL;
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        NtrHostBvh* h = new NtrHostBvh();
        h->scene = new Scene(numTris, (const Vec3i*)triVtxIndex, numVerts, (const Vec3f*)vtxPos);
        Platform platform("GPU");  // Renderer.cpp:88-89
        platform.setLeafPreferences(minLeafSize, maxLeafSize);
        BVH::BuildParams params;
        params.stats = &h->stats;
        const double t1 = now();
        double t2, t3;
        {
        BVH bvh(h->scene, platform, params);
        t2 = now();
        h->cbvh = new CudaBVH(bvh, BVHLayout_Compact);
        t3 = now();
        }
        if (timing) std::fprintf(stderr, "ntr_sah_build: scene %.2f s, BVH (builder + stats) %.2f s, Compact conversion %.2f s, tree teardown %.2f s\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
        *out = h;
        return NTR_OK;
    } catch (const FatalError& e) {
        return ntr::set_error(NTR_ERR_INVALID, "%s", e.message.c_str());
    } catch (const std::bad_alloc&) {
        return ntr::set_error(NTR_ERR_NOMEM, "ntr_sah_build: out of host memory");
    } catch (const std::exception& e) {   // e.g. std::system_error from a thread constructor: nothing may escape the extern "C" boundary
        return ntr::set_error(NTR_ERR_INVALID, "%s: %s", __func__, e.what());
    } catch (...) {
        return ntr::set_error(NTR_ERR_INVALID, "%s: unknown exception", __func__);
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

int ntr_host_bvh_wrap(const void* nodes, int64_t nodesBytes, const void* triWoop, int64_t triWoopBytes, const int32_t* triIndex,
                      int64_t triIndexBytes, NtrHostBvh** out)
{
    if (!out) return ntr::set_error(NTR_ERR_INVALID, "ntr_host_bvh_wrap: null out");
    *out = nullptr;
    if (!nodes || !triWoop || !triIndex || nodesBytes < 64 || (nodesBytes % 64) != 0 || triWoopBytes < 16 || (triWoopBytes % 16) != 0 ||
        triIndexBytes * 4 < triWoopBytes)
        return ntr::set_error(NTR_ERR_INVALID, "ntr_host_bvh_wrap: bad buffer sizes");
    try {
        NtrHostBvh* h = new NtrHostBvh();
        h->scene = nullptr;
        h->cbvh = new CudaBVH(BVHLayout_Compact);
        h->cbvh->getNodeBuffer().set(nodes, nodesBytes);
        h->cbvh->getTriWoopBuffer().set(triWoop, triWoopBytes);
        h->cbvh->getTriIndexBuffer().set(triIndex, triIndexBytes);
        *out = h;
        return NTR_OK;
    } catch (const FatalError& e) {
        return ntr::set_error(NTR_ERR_INVALID, "%s", e.message.c_str());
    } catch (const std::bad_alloc&) {
        return ntr::set_error(NTR_ERR_NOMEM, "ntr_host_bvh_wrap: out of host memory");
    } catch (const std::exception& e) {   // e.g. std::system_error from a thread constructor: nothing may escape the extern "C" boundary
        return ntr::set_error(NTR_ERR_INVALID, "%s: %s", __func__, e.what());
    } catch (...) {
        return ntr::set_error(NTR_ERR_INVALID, "%s: unknown exception", __func__);
    }
}

int ntr_host_bvh_trace(const NtrHostBvh* bvh, int32_t numRays, int32_t anyHit, const NtrRay* rays, NtrRayResult* results,
                       int32_t* visibility, int32_t numVisibility, NtrTraceStats* stats)
{
    if (!bvh || numRays < 0 || (numRays && (!rays || !results)) || numVisibility < 0 || (numVisibility && !visibility))
        return ntr::set_error(NTR_ERR_INVALID, "ntr_host_bvh_trace: bad argument");
    try {
        RayBuffer rb(numRays, anyHit == 0);
        if (numRays) memcpy(rb.getRayBuffer().getMutablePtr(), rays, (size_t)numRays * sizeof(NtrRay));
        Buffer vis;
        if (numVisibility) vis.wrapCPU(visibility, (S64)numVisibility * (S64)sizeof(int32_t));
        RayStats rs;
        bvh->cbvh->trace(rb, vis, stats ? &rs : NULL);
        if (numRays) memcpy(results, rb.getResultBuffer().getPtr(), (size_t)numRays * sizeof(NtrRayResult));
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->numRays = rs.numRays;
            stats->numInnerVisits = rs.numNodeTests / 2;
            stats->numTriTests = rs.numTriangleTests;
        }
        return NTR_OK;
    } catch (const FatalError& e) {
        return ntr::set_error(NTR_ERR_INVALID, "%s", e.message.c_str());
    } catch (const std::bad_alloc&) {
        return ntr::set_error(NTR_ERR_NOMEM, "ntr_host_bvh_trace: out of host memory");
    } catch (const std::exception& e) {   // e.g. std::system_error from a thread constructor: nothing may escape the extern "C" boundary
        return ntr::set_error(NTR_ERR_INVALID, "%s: %s", __func__, e.what());
    } catch (...) {
        return ntr::set_error(NTR_ERR_INVALID, "%s: unknown exception", __func__);
    }
}

int ntr_camera_decode(const char* signature, float out[16])
{
    if (!signature || !out) return ntr::set_error(NTR_ERR_INVALID, "ntr_camera_decode: null argument");
    clearError();
    CameraControls c;
    c.decodeSignature(signature);
    if (hasError()) { int rc = ntr::set_error(NTR_ERR_INVALID, "%s", getError().c_str()); clearError(); return rc; }
    const float v[13] = {c.getPosition().x, c.getPosition().y, c.getPosition().z, c.getForward().x, c.getForward().y, c.getForward().z,
                         c.getUp().x, c.getUp().y, c.getUp().z, c.getSpeed(), c.getFOV(), c.getNear(), c.getFar()};
    memcpy(out, v, sizeof(v));
    out[13] = c.getKeepAligned() ? 1.0f : 0.0f;
    out[14] = out[15] = 0.0f;
    return NTR_OK;
}

int ntr_camera_reencode(const char* signature, char* out, int32_t outSize)
{
    if (!signature || !out || outSize < 2) return ntr::set_error(NTR_ERR_INVALID, "ntr_camera_reencode: bad argument");
    clearError();
    CameraControls c;
    c.decodeSignature(signature);
    if (hasError()) { int rc = ntr::set_error(NTR_ERR_INVALID, "%s", getError().c_str()); clearError(); return rc; }
    String s = c.encodeSignature();
    if ((int)s.size() + 1 > outSize) return ntr::set_error(NTR_ERR_INVALID, "ntr_camera_reencode: buffer too small");
    memcpy(out, s.c_str(), s.size() + 1);
    return NTR_OK;
}

int ntr_camera_nscreen_to_world(const char* signature, int32_t viewW, int32_t viewH, float matrix[16], float position[3], float* cameraFar)
{
    if (!signature || !matrix || !position || !cameraFar || viewW < 1 || viewH < 1)
        return ntr::set_error(NTR_ERR_INVALID, "ntr_camera_nscreen_to_world: bad argument");
    clearError();
    CameraControls c;
    c.decodeSignature(signature);
    if (hasError()) { int rc = ntr::set_error(NTR_ERR_INVALID, "%s", getError().c_str()); clearError(); return rc; }
    CameraView v = c.getView(viewW, viewH);
    memcpy(matrix, v.nscreenToWorld.m, sizeof(float) * 16);
    position[0] = v.position.x; position[1] = v.position.y; position[2] = v.position.z;
    *cameraFar = v.cameraFar;
    return NTR_OK;
}

struct NtrObjMesh {
    std::vector<Vec3i> tris;
    WavefrontMesh mesh;
};

int ntr_obj_load(const char* path, NtrObjMesh** out, int32_t* numTris, int32_t* numVerts, int32_t* numSubmeshes)
{
    if (!path || !out) return ntr::set_error(NTR_ERR_INVALID, "ntr_obj_load: null argument");
    *out = nullptr;
    clearError();
    NtrObjMesh* m = new NtrObjMesh();
    if (!importWavefrontMesh(m->mesh, path)) {
        int rc = ntr::set_error(NTR_ERR_INVALID, "%s", getError().c_str());
        clearError();
        delete m;
        return rc;
    }
    m->mesh.flatten(m->tris);
    if (numTris) *numTris = (int32_t)m->tris.size();
    if (numVerts) *numVerts = (int32_t)m->mesh.vertices.size();
    if (numSubmeshes) *numSubmeshes = (int32_t)m->mesh.submeshes.size();
    *out = m;
    return NTR_OK;
}

int ntr_obj_get(const NtrObjMesh* mesh, int32_t* triVtxIndex, float* vtxPos)
{
    if (!mesh) return ntr::set_error(NTR_ERR_INVALID, "ntr_obj_get: null mesh");
    if (triVtxIndex) memcpy(triVtxIndex, mesh->tris.data(), mesh->tris.size() * sizeof(Vec3i));
    if (vtxPos) memcpy(vtxPos, mesh->mesh.vertices.data(), mesh->mesh.vertices.size() * sizeof(Vec3f));
    return NTR_OK;
}

void ntr_obj_free(NtrObjMesh* mesh) { delete mesh; }

void ntr_host_bvh_free(NtrHostBvh* bvh)
{
    if (!bvh) return;
    delete bvh->cbvh;
    delete bvh->scene;
    delete bvh;
}

}  // extern "C"
