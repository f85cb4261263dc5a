// Platform.hpp -- SAH cost / batching / leaf-size settings (src/rt/bvh/Platform.hpp:45-160).
#pragma once
#include "../Defs.hpp"
#include "../Hash.hpp"

namespace FW {

class Platform {
public:
    Platform() : m_name("Default"), m_SAHNodeCost(1.f), m_SAHTriangleCost(1.f), m_triBatchSize(1), m_nodeBatchSize(1), m_minLeafSize(1), m_maxLeafSize(0x7FFFFFF) {}
    Platform(const String& name, float nodeCost = 1.f, float triCost = 1.f, S32 nodeBatchSize = 1, S32 triBatchSize = 1)
        : m_name(name), m_SAHNodeCost(nodeCost), m_SAHTriangleCost(triCost), m_triBatchSize(triBatchSize), m_nodeBatchSize(nodeBatchSize), m_minLeafSize(1), m_maxLeafSize(0x7FFFFFF) {}

    const String& getName() const { return m_name; }
    float getSAHTriangleCost() const { return m_SAHTriangleCost; }
    float getSAHNodeCost() const { return m_SAHNodeCost; }
    float getCost(int numChildNodes, int numTris) const { return getNodeCost(numChildNodes) + getTriangleCost(numTris); }
    float getTriangleCost(S32 n) const { return roundToTriangleBatchSize(n) * m_SAHTriangleCost; }
    float getNodeCost(S32 n) const { return roundToNodeBatchSize(n) * m_SAHNodeCost; }
    S32   getTriangleBatchSize() const { return m_triBatchSize; }
    S32   getNodeBatchSize() const { return m_nodeBatchSize; }
    void  setTriangleBatchSize(S32 b) { m_triBatchSize = b; }
    void  setNodeBatchSize(S32 b) { m_nodeBatchSize = b; }
    S32   roundToTriangleBatchSize(S32 n) const { return ((n + m_triBatchSize - 1) / m_triBatchSize) * m_triBatchSize; }
    S32   roundToNodeBatchSize(S32 n) const { return ((n + m_nodeBatchSize - 1) / m_nodeBatchSize) * m_nodeBatchSize; }
    void  setLeafPreferences(S32 minSize, S32 maxSize) { m_minLeafSize = minSize; m_maxLeafSize = maxSize; }
    S32   getMinLeafSize() const { return m_minLeafSize; }
    S32   getMaxLeafSize() const { return m_maxLeafSize; }
    U32   computeHash() const;

private:
    String m_name;
    float  m_SAHNodeCost;
    float  m_SAHTriangleCost;
    S32    m_triBatchSize;
    S32    m_nodeBatchSize;
    S32    m_minLeafSize;
    S32    m_maxLeafSize;
};

// src/rt/bvh/Platform.hpp:162
inline U32 Platform::computeHash() const
{
    return hashBits(hashString(m_name), floatToBits(m_SAHNodeCost), floatToBits(m_SAHTriangleCost),
                    hashBits((U32)m_triBatchSize, (U32)m_nodeBatchSize, (U32)m_minLeafSize, (U32)m_maxLeafSize));
}

}  // namespace FW
