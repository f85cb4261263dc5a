#include "Scene.hpp"

#include <cmath>

namespace FW {

Scene::Scene(S32 numTris, const Vec3i* triVtxIndex, S32 numVerts, const Vec3f* vtxPos)
    : m_numTriangles(numTris), m_numVertices(numVerts)
{
    m_triVtxIndex.set(triVtxIndex, (S64)numTris * sizeof(Vec3i));
    m_vtxPos.set(vtxPos, (S64)numVerts * sizeof(Vec3f));
    m_triNormal.resizeDiscard((S64)numTris * sizeof(Vec3f));
    Vec3f* nrm = (Vec3f*)m_triNormal.getMutablePtr();

    // Scene.cpp:112-135: bbox over vertices, per-triangle geometric normal.
    AABB box;
    for (S32 i = 0; i < numVerts; i++) box.grow(vtxPos[i]);
    m_AABBMin = box.min();
    m_AABBMax = box.max();
    for (S32 i = 0; i < numTris; i++) {
        const Vec3f& a = vtxPos[triVtxIndex[i].x];
        const Vec3f& b = vtxPos[triVtxIndex[i].y];
        const Vec3f& c = vtxPos[triVtxIndex[i].z];
        Vec3f n = cross(b - a, c - a);
        F32 len = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
        nrm[i] = (len > 0.0f) ? n * (1.0f / len) : n;
    }
}

U32 Scene::hash(void)
{
    // FNV-1a over the geometry; only used to key BVH cache files.
    U32 h = 2166136261u;
    const U8* p = m_triVtxIndex.getPtr();
    for (S64 i = 0; i < m_triVtxIndex.getSize(); i++) h = (h ^ p[i]) * 16777619u;
    p = m_vtxPos.getPtr();
    for (S64 i = 0; i < m_vtxPos.getSize(); i++) h = (h ^ p[i]) * 16777619u;
    return h;
}

}  // namespace FW
