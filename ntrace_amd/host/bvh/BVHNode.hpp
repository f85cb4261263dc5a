// BVHNode.hpp -- host BVH node types (src/rt/bvh/BVHNode.hpp: SplitInfo :55-120,
// BVHNode/InnerNode/LeafNode :130-260), reduced to what the builders and the
// Compact flattener use.
#pragma once
#include "../Defs.hpp"

namespace FW {

class SplitInfo {
public:
    enum SplitType { SAH, SBVH, OSAH };
    enum SplitAxis { SPLIT_X, SPLIT_Y, SPLIT_Z };
    SplitInfo() : m_code(0) {}
    SplitInfo(S32 axis, SplitType splitType, bool osahTested) : m_code(0)
    {
        m_code |= ((unsigned long)osahTested << 31);
        m_code |= (unsigned long)splitType << 2;
        m_code |= (unsigned long)axis;
    }
    explicit SplitInfo(unsigned long bitCode) : m_code(bitCode) {}
    SplitType     getType() const { return (SplitType)((m_code & 0xC) >> 2); }
    SplitAxis     getAxis() const { return (SplitAxis)(m_code & 0x3); }
    unsigned long getBitCode() const { return m_code; }

private:
    unsigned long m_code;
};

class BVHNode {
public:
    BVHNode() {}
    virtual ~BVHNode() {}
    virtual bool     isLeaf() const = 0;
    virtual S32      getNumChildNodes() const = 0;
    virtual BVHNode* getChildNode(S32 i) const = 0;
    virtual S32      getNumTriangles() const { return 0; }
    void             deleteSubtree();
    S32              getSubtreeDepth() const;
    S32              countNodes(bool inner) const;

    AABB m_bounds;
};

class InnerNode : public BVHNode {
public:
    InnerNode(const AABB& bounds, BVHNode* child0, BVHNode* child1, S32 axis, SplitInfo::SplitType splitType, bool osahTested)
        : m_splitInfo(axis, splitType, osahTested)
    {
        m_bounds = bounds;
        m_children[0] = child0;
        m_children[1] = child1;
    }
    bool             isLeaf() const { return false; }
    S32              getNumChildNodes() const { return 2; }
    BVHNode*         getChildNode(S32 i) const { return m_children[i]; }
    const SplitInfo& getSplitInfo() const { return m_splitInfo; }

    BVHNode*  m_children[2];
    SplitInfo m_splitInfo;
};

class LeafNode : public BVHNode {
public:
    LeafNode(const AABB& bounds, int lo, int hi) : m_lo(lo), m_hi(hi) { m_bounds = bounds; }
    bool     isLeaf() const { return true; }
    S32      getNumChildNodes() const { return 0; }
    BVHNode* getChildNode(S32) const { return NULL; }
    S32      getNumTriangles() const { return m_hi - m_lo; }

    S32 m_lo;
    S32 m_hi;
};

}  // namespace FW
