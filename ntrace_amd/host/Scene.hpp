// Scene.hpp -- minimal mirror of FW::Scene (src/rt/Scene.hpp, Scene.cpp:34-167):
// flat triangle/vertex buffers and the scene bounding box.  Materials, texture
// atlas and emissive lists are outside this backend's scope.
#pragma once
#include "Buffer.hpp"

namespace FW {

class Scene {
public:
    // triVtxIndex: numTris x Vec3i, vtxPos: numVerts x Vec3f (copied).
    Scene(S32 numTris, const Vec3i* triVtxIndex, S32 numVerts, const Vec3f* vtxPos);

    int     getNumTriangles(void) const { return m_numTriangles; }
    int     getNumVertices(void) const { return m_numVertices; }
    Buffer& getTriVtxIndexBuffer(void) { return m_triVtxIndex; }
    Buffer& getVtxPosBuffer(void) { return m_vtxPos; }
    Buffer& getTriNormalBuffer(void) { return m_triNormal; }
    void    getBBox(Vec3f& lo, Vec3f& hi) const { lo = m_AABBMin; hi = m_AABBMax; }
    U32     hash(void);

private:
    Scene(const Scene&);
    Scene& operator=(const Scene&);

    S32    m_numTriangles;
    S32    m_numVertices;
    Buffer m_triVtxIndex;  // Vec3i[numTris]
    Buffer m_triNormal;    // Vec3f[numTris]
    Buffer m_vtxPos;       // Vec3f[numVerts]
    Vec3f  m_AABBMin, m_AABBMax;
};

}  // namespace FW
