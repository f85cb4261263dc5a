// SAHBVHBuilder.hpp -- full-sweep SAH object-split builder: the trees of src/rt/bvh/SAHBVHBuilder.cpp:51-254,
// built in O(n log n) from three presorted primitive orders, subtrees in parallel.  Host-side "prebuilt BVH"
// producer for the Cornell / Sponza / Conference configurations.
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

    struct Job {       // one node to build: the same primitives occupy [begin, end) of all three orders
        S32  begin, end;
        S32  level;
        S32  triBase;   // where this subtree's triangles start in BVH::getTriIndices()
        S32  order;     // the order (0..2) the primitives were last arranged by when this node is reached
        AABB bounds;
    };
    struct Split {
        F32 sah;
        S32 dim;
        S32 numLeft;
        Split(void) : sah(FW_F32_MAX), dim(0), numLeft(0) {}
    };
    struct Scratch {   // per thread
        std::vector<F32> rightArea;
        std::vector<S32> tmp;
    };

    BVHNode* build(const Job& job, Scratch& scratch, int spawnDepth);
    BVHNode* leaf(const Job& job, int order);
    Split    bestSplit(const Job& job, F32 nodeSAH, Scratch& scratch, bool threaded) const;

    BVH&              m_bvh;
    const Platform&   m_platform;
    BVH::BuildParams  m_params;
    std::vector<AABB> m_box;       // per triangle
    std::vector<F32>  m_key[3];    // per triangle and axis: box.min + box.max, the reference's sort key
    std::vector<S32>  m_order[3];  // the live triangles sorted by (key[axis], triangle index)
    std::vector<U8>   m_side;      // per triangle: goes to the left child of the node being split
};

}  // namespace FW
