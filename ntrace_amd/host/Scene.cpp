#include "Scene.hpp"

#include "Hash.hpp"

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

// Scene::hash (src/rt/Scene.cpp:171-179): hashBits over the hashes of the scene's buffers.  The reference also hashes its per-triangle
// material / shaded colour buffers, which this mirror does not hold (shading is out of scope): their slots hash an empty buffer, so
// a cache file NAME matches the reference's only for scenes without them; the file FORMAT is the reference's either way.
U32 Scene::hash(void)
{
    const U32 empty = hashBuffer(NULL, 0);
    return hashBits(hashBuffer(m_triVtxIndex.getPtr(), m_triVtxIndex.getSize()), hashBuffer(m_triNormal.getPtr(), m_triNormal.getSize()), empty, empty,
                    hashBuffer(m_vtxPos.getPtr(), m_vtxPos.getSize()));
}

}  // namespace FW
