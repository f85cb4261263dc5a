// SAHBVHBuilder.cpp -- behaviour follows src/rt/bvh/SAHBVHBuilder.cpp:51-254:
// same degenerate filter, leaf conditions, SAH cost model, sweep, tie-breaks and
// right-before-left recursion over a reference stack, so triangle ids land in
// leaves in the same order as the reference builder produces them.
#include "SAHBVHBuilder.hpp"

#include <algorithm>

namespace FW {

SAHBVHBuilder::SAHBVHBuilder(BVH& bvh, const BVH::BuildParams& params)
    : m_bvh(bvh), m_platform(bvh.getPlatform()), m_params(params)
{
}

BVHNode* SAHBVHBuilder::run(void)
{
    const Vec3i* tris = (const Vec3i*)m_bvh.getScene()->getTriVtxIndexBuffer().getPtr();
    const Vec3f* verts = (const Vec3f*)m_bvh.getScene()->getVtxPosBuffer().getPtr();

    NodeSpec rootSpec;
    rootSpec.numRef = m_bvh.getScene()->getNumTriangles();
    m_refStack.resize(rootSpec.numRef);
    for (int i = 0; i < rootSpec.numRef; i++) {
        m_refStack[i].triIdx = i;
        for (int j = 0; j < 3; j++) m_refStack[i].bounds.grow(verts[tris[i][j]]);
        rootSpec.bounds.grow(m_refStack[i].bounds);
    }
    m_rightBounds.assign(std::max(rootSpec.numRef, 1), AABB());
    m_bvh.getTriIndices().clear();
    return buildNode(rootSpec, 0);
}

// Order of SAHBVHBuilder::sortCompare (:106-115): centroid (min+max) on the sort
// axis, ties by triangle index -- a strict total order, so any sort gives the same
// permutation as the reference's FW::sort.
void SAHBVHBuilder::sortTop(int numRef, int dim)
{
    std::sort(m_refStack.end() - numRef, m_refStack.end(), [dim](const Reference& ra, const Reference& rb) {
        F32 ca = ra.bounds.min()[dim] + ra.bounds.max()[dim];
        F32 cb = rb.bounds.min()[dim] + rb.bounds.max()[dim];
        return (ca < cb || (ca == cb && ra.triIdx < rb.triIdx));
    });
}

BVHNode* SAHBVHBuilder::buildNode(NodeSpec& spec, int level)
{
    // Remove degenerates (:141-151): negative extent, or at most one non-zero extent.
    {
        int firstRef = (int)m_refStack.size() - spec.numRef;
        for (int i = (int)m_refStack.size() - 1; i >= firstRef; i--) {
            Vec3f size = m_refStack[i].bounds.max() - m_refStack[i].bounds.min();
            if (size.min() < 0.0f || size.sum() == size.max()) {
                m_refStack[i] = m_refStack.back();  // Array::removeSwap
                m_refStack.pop_back();
            }
        }
        spec.numRef = (int)m_refStack.size() - firstRef;
    }

    // Small enough or too deep => leaf; the root is never a leaf (:155-156).
    if ((level != 0 && spec.numRef <= m_platform.getMinLeafSize()) || level >= MaxDepth)
        return createLeaf(spec);

    F32 area = spec.bounds.area();
    F32 leafSAH = area * m_platform.getTriangleCost(spec.numRef);
    F32 nodeSAH = area * m_platform.getNodeCost(2);
    ObjectSplit object = findObjectSplit(spec, nodeSAH);

    F32 minSAH = FW::min(leafSAH, object.sah);
    if (level != 0 && minSAH == leafSAH && spec.numRef <= m_platform.getMaxLeafSize())
        return createLeaf(spec);

    NodeSpec left, right;
    performObjectSplit(left, right, spec, object);

    // The right half sits on top of the reference stack: build it first (:187-188).
    BVHNode* rightNode = buildNode(right, level + 1);
    BVHNode* leftNode = buildNode(left, level + 1);
    return new InnerNode(spec.bounds, leftNode, rightNode, object.sortDim, SplitInfo::SAH, false);
}

BVHNode* SAHBVHBuilder::createLeaf(const NodeSpec& spec)
{
    std::vector<S32>& tris = m_bvh.getTriIndices();
    for (int i = 0; i < spec.numRef; i++) {
        tris.push_back(m_refStack.back().triIdx);
        m_refStack.pop_back();
    }
    return new LeafNode(spec.bounds, (int)tris.size() - spec.numRef, (int)tris.size());
}

SAHBVHBuilder::ObjectSplit SAHBVHBuilder::findObjectSplit(const NodeSpec& spec, F32 nodeSAH)
{
    ObjectSplit split;
    F32 bestTieBreak = FW_F32_MAX;

    for (int dim = 0; dim < 3; dim++) {
        sortTop(spec.numRef, dim);
        const Reference* refPtr = m_refStack.data() + (m_refStack.size() - spec.numRef);

        AABB rightBounds;
        for (int i = spec.numRef - 1; i > 0; i--) {
            rightBounds.grow(refPtr[i].bounds);
            m_rightBounds[i - 1] = rightBounds;
        }

        AABB leftBounds;
        for (int i = 1; i < spec.numRef; i++) {
            leftBounds.grow(refPtr[i - 1].bounds);
            F32 sah = nodeSAH + leftBounds.area() * m_platform.getTriangleCost(i) +
                      m_rightBounds[i - 1].area() * m_platform.getTriangleCost(spec.numRef - i);
            F32 fi = (F32)i, fr = (F32)(spec.numRef - i);
            F32 tieBreak = fi * fi + fr * fr;
            if (sah < split.sah || (sah == split.sah && tieBreak < bestTieBreak)) {
                split.sah = sah;
                split.sortDim = dim;
                split.numLeft = i;
                split.leftBounds = leftBounds;
                split.rightBounds = m_rightBounds[i - 1];
                bestTieBreak = tieBreak;
            }
        }
    }
    return split;
}

void SAHBVHBuilder::performObjectSplit(NodeSpec& left, NodeSpec& right, const NodeSpec& spec, const ObjectSplit& split)
{
    sortTop(spec.numRef, split.sortDim);
    left.numRef = split.numLeft;
    left.bounds = split.leftBounds;
    right.numRef = spec.numRef - split.numLeft;
    right.bounds = split.rightBounds;
}

}  // namespace FW
