// SAHBVHBuilder.hpp -- full-sweep SAH object-split builder
// (src/rt/bvh/SAHBVHBuilder.hpp, SAHBVHBuilder.cpp:51-254).  Host-side "prebuilt
// BVH" producer for the Cornell/Sponza/Conference configurations.
#pragma once
#include <vector>

#include "BVH.hpp"

namespace FW {

class SAHBVHBuilder {
public:
    SAHBVHBuilder(BVH& bvh, const BVH::BuildParams& params);
    BVHNode* run(void);

private:
    enum { MaxDepth = 64 };  // SAHBVHBuilder.hpp:48

    struct Reference {
        S32  triIdx;
        AABB bounds;
        Reference(void) : triIdx(-1) {}
    };
    struct NodeSpec {
        S32  numRef;
        AABB bounds;
        NodeSpec(void) : numRef(0) {}
    };
    struct ObjectSplit {
        F32  sah;
        S32  sortDim;
        S32  numLeft;
        AABB leftBounds;
        AABB rightBounds;
        ObjectSplit(void) : sah(FW_F32_MAX), sortDim(0), numLeft(0) {}
    };

    BVHNode*    buildNode(NodeSpec& spec, int level);
    BVHNode*    createLeaf(const NodeSpec& spec);
    ObjectSplit findObjectSplit(const NodeSpec& spec, F32 nodeSAH);
    void        performObjectSplit(NodeSpec& left, NodeSpec& right, const NodeSpec& spec, const ObjectSplit& split);
    void        sortTop(int numRef, int dim);

    BVH&                   m_bvh;
    const Platform&        m_platform;
    BVH::BuildParams       m_params;
    std::vector<Reference> m_refStack;
    std::vector<AABB>      m_rightBounds;
};

}  // namespace FW
