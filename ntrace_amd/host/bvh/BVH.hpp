// BVH.hpp -- host BVH container + builder dispatch (src/rt/bvh/BVH.hpp:74-190,
// BVH.cpp:36-88).  The reference reads the builder name from the Environment
// singleton ("Renderer.builder"); here it is a BuildParams field.
#pragma once
#include <vector>

#include "../Hash.hpp"
#include "../Scene.hpp"
#include "BVHNode.hpp"
#include "Platform.hpp"

namespace FW {

class BVH {
public:
    struct Stats {
        Stats() { clear(); }
        void clear() { memset(this, 0, sizeof(Stats)); }
        void print() const;
        F32 SAHCost;
        S32 branchingFactor;
        S32 maxDepth;
        S32 numInnerNodes;
        S32 numLeafNodes;
        S32 numChildNodes;
        S32 numTris;
        F32 buildTime;
    };

    struct BuildParams {
        Stats* stats;
        bool   enablePrints;
        F32    splitAlpha;
        String builder;  // "SAHBVH" (Renderer.builder in config.conf)
        BuildParams(void) : stats(NULL), enablePrints(false), splitAlpha(1.0e-5f), builder("SAHBVH") {}
        U32 computeHash(void) const { return hashBits(floatToBits(splitAlpha)); }  // src/rt/bvh/BVH.hpp:143
    };

    BVH(Scene* scene, const Platform& platform, const BuildParams& params);
    ~BVH(void) { if (m_root) m_root->deleteSubtree(); }

    Scene*                  getScene(void) const { return m_scene; }
    const Platform&         getPlatform(void) const { return m_platform; }
    BVHNode*                getRoot(void) const { return m_root; }
    std::vector<S32>&       getTriIndices(void) { return m_triIndices; }
    const std::vector<S32>& getTriIndices(void) const { return m_triIndices; }

private:
    BVH(const BVH&);
    BVH& operator=(const BVH&);

    Scene*           m_scene;
    Platform         m_platform;
    BVHNode*         m_root;
    std::vector<S32> m_triIndices;
};

}  // namespace FW
