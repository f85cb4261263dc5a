#include "BVH.hpp"

#include <chrono>
#include <cstdio>

#include "SAHBVHBuilder.hpp"

namespace FW {

void BVHNode::deleteSubtree()
{
    // iterative: SAH trees may be 64 deep but LBVH-imported ones can be deeper
    std::vector<BVHNode*> stack(1, this);
    while (!stack.empty()) {
        BVHNode* n = stack.back();
        stack.pop_back();
        for (int i = 0; i < n->getNumChildNodes(); i++) stack.push_back(n->getChildNode(i));
        delete n;
    }
}

S32 BVHNode::getSubtreeDepth() const
{
    S32 best = 0;
    std::vector<std::pair<const BVHNode*, S32> > stack(1, std::make_pair(this, 1));
    while (!stack.empty()) {
        std::pair<const BVHNode*, S32> e = stack.back();
        stack.pop_back();
        best = FW::max(best, e.second);
        for (int i = 0; i < e.first->getNumChildNodes(); i++) stack.push_back(std::make_pair(e.first->getChildNode(i), e.second + 1));
    }
    return best;
}

S32 BVHNode::countNodes(bool inner) const
{
    S32 cnt = 0;
    std::vector<const BVHNode*> stack(1, this);
    while (!stack.empty()) {
        const BVHNode* n = stack.back();
        stack.pop_back();
        if (n->isLeaf() != inner) cnt++;
        for (int i = 0; i < n->getNumChildNodes(); i++) stack.push_back(n->getChildNode(i));
    }
    return cnt;
}

void BVH::Stats::print() const
{
    printf("Tree stats: [bfactor=%d] %d nodes (%d+%d), %.2f SAHCost, %.1f children/inner, %.1f tris/leaf\n", branchingFactor,
           numLeafNodes + numInnerNodes, numLeafNodes, numInnerNodes, SAHCost, 1.f * numChildNodes / FW::max(numInnerNodes, 1),
           1.f * numTris / FW::max(numLeafNodes, 1));
}

// BVH.cpp:36-88: dispatch on the builder name, time the build, fill Stats.
BVH::BVH(Scene* scene, const Platform& platform, const BuildParams& params)
    : m_scene(scene), m_platform(platform), m_root(NULL)
{
    if (params.enablePrints)
        printf("BVH builder: %d tris, %d vertices\n", scene->getNumTriangles(), scene->getNumVertices());

    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    if (params.builder == "SAHBVH")
        m_root = SAHBVHBuilder(*this, params).run();
    else
        fail("Unsupported BVH builder %s\n", params.builder.c_str());
    F32 time = std::chrono::duration<F32>(std::chrono::steady_clock::now() - t0).count();

    if (params.stats) {
        params.stats->branchingFactor = 2;
        params.stats->maxDepth = m_root->getSubtreeDepth();
        params.stats->numLeafNodes = m_root->countNodes(false);
        params.stats->numInnerNodes = m_root->countNodes(true);
        params.stats->numChildNodes = params.stats->numInnerNodes * 2;
        params.stats->numTris = (S32)m_triIndices.size();
        params.stats->buildTime = time;
    }
}

}  // namespace FW
