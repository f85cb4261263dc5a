// SAHBVHBuilder.cpp -- SAH object-split builder that yields, triangle for triangle, the trees of the reference's
// SAHBVHBuilder (src/rt/bvh/SAHBVHBuilder.cpp:51-254), by another route.
//
// The reference keeps a stack of references and, at every node, sorts the node's references three times (once per
// axis) plus once more for the chosen axis: O(n log^2 n), single-threaded.  Its results depend only on
//   * the order (box.min + box.max on the axis, ties by triangle index) -- a strict total order (:106-115), so a
//     node's sorted sequence is unique however it is obtained;
//   * the sweep expression  nodeSAH + area(left) * triCost(i) + area(right) * triCost(n - i)  and the tie rules
//     "smaller sah, then smaller i^2 + (n-i)^2, then earlier axis, then smaller i" (:221-232);
//   * the leaf rules (:155-156, :163-165) and the order in which a leaf pops its references off the stack
//     (:193-203): back to front of whatever order the node's range was last sorted by -- axis 2 once the split
//     search has run, the parent's split axis when the node became a leaf before searching;
//   * right subtree first (:187-188), which fixes every leaf's position in the triangle index array.
// Here the three orders are sorted ONCE; a split marks the triangles of the left side and stable-partitions the
// other two orders, which keeps every child's three sequences sorted.  A node of m triangles costs O(m), the tree
// O(n log n).  Because the right-first rule gives every subtree a known slice of the triangle index array, large
// subtrees are built by separate threads.  Box unions are exact (min / max), so sweeping in another grouping changes
// no area and no cost; the Compact buffers are byte-identical to the reference-order builder's
// (tests/test_sah_builder_cpu.py pins their hashes).
#include "SAHBVHBuilder.hpp"

#include <algorithm>
#include <thread>

namespace FW {

SAHBVHBuilder::SAHBVHBuilder(BVH& bvh, const BVH::BuildParams& params)
    : m_bvh(bvh), m_platform(bvh.getPlatform()), m_params(params)
{
}

BVHNode* SAHBVHBuilder::run(void)
{
    const Scene* scene = m_bvh.getScene();
    const Vec3i* tris = (const Vec3i*)m_bvh.getScene()->getTriVtxIndexBuffer().getPtr();
    const Vec3f* verts = (const Vec3f*)m_bvh.getScene()->getVtxPosBuffer().getPtr();
    const S32 numTris = scene->getNumTriangles();

    // Boxes, sort keys; the root's box covers every triangle, the degenerate ones too (:70-84).  Triangles whose box
    // has a negative extent or at most one non-zero extent never reach a leaf (:141-151).
    Job root;
    root.level = 0;
    root.triBase = 0;
    root.order = 2;
    m_box.resize(numTris);
    for (int d = 0; d < 3; d++) m_key[d].resize(numTris);
    std::vector<S32> live;
    live.reserve(numTris);
    for (S32 t = 0; t < numTris; t++) {
        AABB b;
        for (int j = 0; j < 3; j++) b.grow(verts[tris[t][j]]);
        m_box[t] = b;
        root.bounds.grow(b);
        for (int d = 0; d < 3; d++) m_key[d][t] = b.min()[d] + b.max()[d];
        const Vec3f size = b.max() - b.min();
        if (!(size.min() < 0.0f || size.sum() == size.max())) live.push_back(t);
    }
    const S32 n = (S32)live.size();
    root.begin = 0;
    root.end = n;
    m_side.assign(numTris, 0);
    m_bvh.getTriIndices().assign(n, 0);

    auto sortAxis = [&](int d) {
        m_order[d] = live;
        const F32* key = m_key[d].data();
        std::sort(m_order[d].begin(), m_order[d].end(), [key](S32 a, S32 b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
    };
    if (n > 100000) {
        std::thread t0(sortAxis, 0), t1(sortAxis, 1);
        sortAxis(2);
        t0.join();
        t1.join();
    } else {
        for (int d = 0; d < 3; d++) sortAxis(d);
    }

    // how many levels of the tree may hand their right child to a new thread
    int spawnDepth = 0;
    if (n > 200000) {
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 1;
        if (hw > 256) hw = 256;
        while ((1u << spawnDepth) < hw) spawnDepth++;
        spawnDepth += 2;  // a few more tasks than threads: SAH splits are uneven
    }
    Scratch scratch;
    BVHNode* node = build(root, scratch, spawnDepth);
    for (int d = 0; d < 3; d++) { std::vector<S32>().swap(m_order[d]); std::vector<F32>().swap(m_key[d]); }
    std::vector<AABB>().swap(m_box);
    return node;
}

// A leaf takes its triangles back to front of the sequence its range was last arranged by.
BVHNode* SAHBVHBuilder::leaf(const Job& job, int order)
{
    S32* out = m_bvh.getTriIndices().data() + job.triBase;
    const S32* seq = m_order[order].data();
    for (S32 i = job.end; i-- > job.begin;) *out++ = seq[i];
    return new LeafNode(job.bounds, job.triBase, job.triBase + (job.end - job.begin));
}

// The sweep of findObjectSplit (:206-243) over the three presorted sequences of the node.
SAHBVHBuilder::Split SAHBVHBuilder::bestSplit(const Job& job, F32 nodeSAH, Scratch& scratch) const
{
    const S32 m = job.end - job.begin;
    Split best;
    F32 bestBalance = FW_F32_MAX;
    if ((S32)scratch.rightArea.size() < m) scratch.rightArea.resize(m);
    F32* rightArea = scratch.rightArea.data();
    for (int d = 0; d < 3; d++) {
        const S32* seq = m_order[d].data() + job.begin;
        AABB acc;
        for (S32 i = m - 1; i > 0; i--) {
            acc.grow(m_box[seq[i]]);
            rightArea[i - 1] = acc.area();
        }
        AABB left;
        for (S32 i = 1; i < m; i++) {
            left.grow(m_box[seq[i - 1]]);
            const F32 sah = nodeSAH + left.area() * m_platform.getTriangleCost(i) + rightArea[i - 1] * m_platform.getTriangleCost(m - i);
            const F32 fl = (F32)i, fr = (F32)(m - i);
            const F32 balance = fl * fl + fr * fr;
            if (sah < best.sah || (sah == best.sah && balance < bestBalance)) {
                best.sah = sah;
                best.dim = d;
                best.numLeft = i;
                bestBalance = balance;
            }
        }
    }
    return best;
}

BVHNode* SAHBVHBuilder::build(const Job& job, Scratch& scratch, int spawnDepth)
{
    const S32 m = job.end - job.begin;
    // small enough or too deep: a leaf, in the order the parent left; the root is never a leaf (:155-156)
    if ((job.level != 0 && m <= m_platform.getMinLeafSize()) || job.level >= MaxDepth) return leaf(job, job.order);

    const F32 area = job.bounds.area();
    const F32 leafSAH = area * m_platform.getTriangleCost(m);
    const F32 nodeSAH = area * m_platform.getNodeCost(2);
    const Split split = bestSplit(job, nodeSAH, scratch);
    const F32 minSAH = FW::min(leafSAH, split.sah);
    // the search has run: the reference's range is now sorted by the last axis (:163-165)
    if (job.level != 0 && minSAH == leafSAH && m <= m_platform.getMaxLeafSize()) return leaf(job, 2);

    // children: first numLeft of the chosen sequence go left; their boxes are the sweep's boxes at the split
    Job left, right;
    const S32* chosen = m_order[split.dim].data() + job.begin;
    for (S32 i = 0; i < split.numLeft; i++) { left.bounds.grow(m_box[chosen[i]]); m_side[chosen[i]] = 1; }
    for (S32 i = split.numLeft; i < m; i++) { right.bounds.grow(m_box[chosen[i]]); m_side[chosen[i]] = 0; }
    if ((S32)scratch.tmp.size() < m) scratch.tmp.resize(m);
    for (int d = 0; d < 3; d++) {
        if (d == split.dim) continue;
        S32* seq = m_order[d].data() + job.begin;
        S32* tmp = scratch.tmp.data();
        S32 nl = 0, nr = 0;
        for (S32 i = 0; i < m; i++) {
            const S32 t = seq[i];
            if (m_side[t]) seq[nl++] = t;  // nl <= i: in place
            else tmp[nr++] = t;
        }
        std::copy(tmp, tmp + nr, seq + nl);
    }
    left.begin = job.begin;
    left.end = job.begin + split.numLeft;
    right.begin = left.end;
    right.end = job.end;
    left.level = right.level = job.level + 1;
    left.order = right.order = split.dim;
    // the right subtree is built first in the reference (:187-188): its triangles come first in the index array
    right.triBase = job.triBase;
    left.triBase = job.triBase + (right.end - right.begin);

    BVHNode* rightNode = NULL;
    BVHNode* leftNode = NULL;
    if (spawnDepth > 0 && m > 65536) {
        std::thread other([&]() { Scratch own; rightNode = build(right, own, spawnDepth - 1); });
        leftNode = build(left, scratch, spawnDepth - 1);
        other.join();
    } else {
        rightNode = build(right, scratch, 0);
        leftNode = build(left, scratch, 0);
    }
    return new InnerNode(job.bounds, leftNode, rightNode, split.dim, SplitInfo::SAH, false);
}

}  // namespace FW
