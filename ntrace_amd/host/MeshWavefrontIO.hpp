// MeshWavefrontIO.hpp -- minimal Wavefront OBJ importer producing Scene buffers with the SAME triangle ids as
// the reference (src/framework/io/MeshWavefrontIO.cpp:412-485 + src/rt/Scene.cpp:101-136):
//   * a vertex is a unique (position, texcoord, normal) index triple, numbered in first-seen order;
//   * polygons are fan-triangulated (v0, v[i-1], v[i]);
//   * one submesh per material in first-`usemtl` order -- only for materials defined by the mtllib
//     (`newmtl`); faces before any (known) usemtl go to a default submesh created on first use;
//   * scene triangle id = submeshes concatenated in creation order.
// Materials, texture coordinates and normals are otherwise ignored (shading is out of scope).
#pragma once
#include <vector>

#include "Scene.hpp"

namespace FW {

struct WavefrontMesh {
    std::vector<Vec3f> vertices;             // unique (p,t,n) vertices, position only
    std::vector<std::vector<Vec3i> > submeshes;
    std::vector<String> submeshMaterial;     // "" for the default submesh

    int numTriangles(void) const;
    void flatten(std::vector<Vec3i>& tris) const;  // submeshes concatenated (Scene.cpp:101-136)
    Scene* createScene(void) const;
};

// Returns false (and sets the sticky error) when the file cannot be read.
bool importWavefrontMesh(WavefrontMesh& mesh, const String& fileName);
bool parseWavefrontMesh(WavefrontMesh& mesh, const String& objText, const std::vector<String>& knownMaterials);

}  // namespace FW
