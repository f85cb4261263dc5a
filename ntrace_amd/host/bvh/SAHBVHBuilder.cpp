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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../Threads.hpp"

namespace FW {

SAHBVHBuilder::SAHBVHBuilder(BVH& bvh, const BVH::BuildParams& params)
    : m_bvh(bvh), m_platform(bvh.getPlatform()), m_params(params)
{
}

BVHNode* SAHBVHBuilder::run(void)
{
    const bool timing = std::getenv("NTR_SAH_TIMING") != NULL;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
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
    unsigned hwThreads = std::thread::hardware_concurrency();
    if (hwThreads == 0) hwThreads = 1;
    if (hwThreads > 64) hwThreads = 64;
    const int prepThreads = numTris > 200000 ? (int)hwThreads : 1;
    std::vector<S32> live;
    {   // slices of the triangle range, one per thread; the live list keeps ascending triangle order, unions are exact
        std::vector<std::vector<S32> > liveOf(prepThreads);
        std::vector<AABB> boundsOf(prepThreads);
        auto prep = [&](int p) {
            const S32 t0 = (S32)((S64)numTris * p / prepThreads), t1 = (S32)((S64)numTris * (p + 1) / prepThreads);
            liveOf[p].reserve(t1 - t0);
            for (S32 t = t0; t < t1; t++) {
                AABB b;
                for (int j = 0; j < 3; j++) b.grow(verts[tris[t][j]]);
                m_box[t] = b;
                boundsOf[p].grow(b);
                for (int d = 0; d < 3; d++) m_key[d][t] = b.min()[d] + b.max()[d];
                const Vec3f size = b.max() - b.min();
                if (!(size.min() < 0.0f || size.sum() == size.max())) liveOf[p].push_back(t);
            }
        };
        ThreadGroup pool;
        for (int p = 1; p < prepThreads; p++) pool.spawn([&prep, p]() { prep(p); });
        pool.run([&prep]() { prep(0); });
        pool.join();
        size_t total = 0;
        for (int p = 0; p < prepThreads; p++) total += liveOf[p].size();
        live.reserve(total);
        for (int p = 0; p < prepThreads; p++) {
            live.insert(live.end(), liveOf[p].begin(), liveOf[p].end());
            if (numTris > 0) root.bounds.grow(boundsOf[p]);
        }
    }
    const S32 n = (S32)live.size();
    root.begin = 0;
    root.end = n;
    m_side.assign(numTris, 0);
    m_bvh.getTriIndices().assign(n, 0);

    const double t1 = now();
    // The order is a strict total order, so the sorted sequence is unique however it is produced: chunks sorted by separate threads,
    // then merged pairwise (the merges of one round run in parallel).
    const int chunksPerAxis = n > 200000 ? (int)std::max(1u, std::min(hwThreads / 3u, 16u)) : 1;
    auto sortAxis = [&](int d) {
        m_order[d] = live;
        const F32* key = m_key[d].data();
        auto less = [key](S32 a, S32 b) { return key[a] < key[b] || (key[a] == key[b] && a < b); };
        S32* base = m_order[d].data();
        std::vector<S32> bounds(chunksPerAxis + 1);
        for (int c = 0; c <= chunksPerAxis; c++) bounds[c] = (S32)((S64)n * c / chunksPerAxis);
        {
            ThreadGroup pool;
            for (int c = 1; c < chunksPerAxis; c++) pool.spawn([=]() { std::sort(base + bounds[c], base + bounds[c + 1], less); });
            pool.run([=]() { std::sort(base + bounds[0], base + bounds[1], less); });
            pool.join();
        }
        for (int width = 1; width < chunksPerAxis; width *= 2) {
            ThreadGroup pool;
            for (int c = 0; c + width < chunksPerAxis; c += 2 * width) {
                const S32 lo = bounds[c], mid = bounds[c + width], hi = bounds[std::min(c + 2 * width, chunksPerAxis)];
                pool.spawn([=]() { std::inplace_merge(base + lo, base + mid, base + hi, less); });
            }
            pool.join();
        }
    };
    if (n > 100000) {
        ThreadGroup axes;
        axes.spawn([&sortAxis]() { sortAxis(0); });
        axes.spawn([&sortAxis]() { sortAxis(1); });
        axes.run([&sortAxis]() { sortAxis(2); });
        axes.join();
    } else {
        for (int d = 0; d < 3; d++) sortAxis(d);
    }

    const double t2 = now();
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
    if (timing) std::fprintf(stderr, "SAHBVHBuilder: boxes %.2f s, presort %.2f s, build %.2f s (%d triangles, spawn depth %d)\n", t1 - t0, t2 - t1, now() - t2, (int)n, spawnDepth);
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

// The sweep of findObjectSplit (:206-243) over ONE presorted sequence of the node: its best split under the reference's rule
// "smaller sah, then smaller i^2 + (n-i)^2, then smaller i" (the first minimum met wins).
static void sweepAxis(const S32* seq, S32 m, const AABB* boxes, const Platform& platform, F32 nodeSAH, F32* rightArea, F32& bestSah,
                      F32& bestBalance, S32& bestLeft)
{
    AABB acc;
    for (S32 i = m - 1; i > 0; i--) {
        acc.grow(boxes[seq[i]]);
        rightArea[i - 1] = acc.area();
    }
    AABB left;
    for (S32 i = 1; i < m; i++) {
        left.grow(boxes[seq[i - 1]]);
        const F32 sah = nodeSAH + left.area() * platform.getTriangleCost(i) + rightArea[i - 1] * platform.getTriangleCost(m - i);
        const F32 fl = (F32)i, fr = (F32)(m - i);
        const F32 balance = fl * fl + fr * fr;
        if (sah < bestSah || (sah == bestSah && balance < bestBalance)) {
            bestSah = sah;
            bestLeft = i;
            bestBalance = balance;
        }
    }
}

// The three sweeps; axes are compared in order with the same rule, so an earlier axis wins ties (:221-232).  The sweeps of a large
// node near the root -- where the tree offers no other parallelism yet -- run on three threads.
SAHBVHBuilder::Split SAHBVHBuilder::bestSplit(const Job& job, F32 nodeSAH, Scratch& scratch, bool threaded) const
{
    const S32 m = job.end - job.begin;
    F32 sah[3] = {FW_F32_MAX, FW_F32_MAX, FW_F32_MAX}, balance[3] = {FW_F32_MAX, FW_F32_MAX, FW_F32_MAX};
    S32 numLeft[3] = {0, 0, 0};
    if (threaded) {
        std::vector<F32> area[3];
        auto one = [&](int d) {
            area[d].resize(m);
            sweepAxis(m_order[d].data() + job.begin, m, m_box.data(), m_platform, nodeSAH, area[d].data(), sah[d], balance[d], numLeft[d]);
        };
        ThreadGroup axes;
        axes.spawn([&one]() { one(0); });
        axes.spawn([&one]() { one(1); });
        axes.run([&one]() { one(2); });
        axes.join();
    } else {
        if ((S32)scratch.rightArea.size() < m) scratch.rightArea.resize(m);
        for (int d = 0; d < 3; d++)
            sweepAxis(m_order[d].data() + job.begin, m, m_box.data(), m_platform, nodeSAH, scratch.rightArea.data(), sah[d], balance[d], numLeft[d]);
    }
    Split best;
    F32 bestBalance = FW_F32_MAX;
    for (int d = 0; d < 3; d++) {
        if (sah[d] < best.sah || (sah[d] == best.sah && balance[d] < bestBalance)) {
            best.sah = sah[d];
            best.dim = d;
            best.numLeft = numLeft[d];
            bestBalance = balance[d];
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
    const bool threaded = spawnDepth > 0 && m > 400000;   // near the root: the node itself is the only work there is
    const Split split = bestSplit(job, nodeSAH, scratch, threaded);
    const F32 minSAH = FW::min(leafSAH, split.sah);
    // the search has run: the reference's range is now sorted by the last axis (:163-165)
    if (job.level != 0 && minSAH == leafSAH && m <= m_platform.getMaxLeafSize()) return leaf(job, 2);

    // children: first numLeft of the chosen sequence go left; their boxes are the sweep's boxes at the split
    Job left, right;
    const S32* chosen = m_order[split.dim].data() + job.begin;
    for (S32 i = 0; i < split.numLeft; i++) { left.bounds.grow(m_box[chosen[i]]); m_side[chosen[i]] = 1; }
    for (S32 i = split.numLeft; i < m; i++) { right.bounds.grow(m_box[chosen[i]]); m_side[chosen[i]] = 0; }
    auto partition = [&](int d, S32* tmp) {
        S32* seq = m_order[d].data() + job.begin;
        S32 nl = 0, nr = 0;
        for (S32 i = 0; i < m; i++) {
            const S32 t = seq[i];
            if (m_side[t]) seq[nl++] = t;  // nl <= i: in place
            else tmp[nr++] = t;
        }
        std::copy(tmp, tmp + nr, seq + nl);
    };
    const int da = (split.dim + 1) % 3, db = (split.dim + 2) % 3;
    if ((S32)scratch.tmp.size() < m) scratch.tmp.resize(m);
    if (threaded) {
        std::vector<S32> tmp2(m);
        ThreadGroup pair;
        S32* tmp2p = tmp2.data();
        pair.spawn([&partition, da, tmp2p]() { partition(da, tmp2p); });
        pair.run([&]() { partition(db, scratch.tmp.data()); });
        pair.join();
    } else {
        partition(da, scratch.tmp.data());
        partition(db, scratch.tmp.data());
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
        ThreadGroup pair;   // an exception in either half (bad_alloc, FatalError) is rethrown here, after both have ended
        pair.spawn([&]() { Scratch own; rightNode = build(right, own, spawnDepth - 1); });
        pair.run([&]() { leftNode = build(left, scratch, spawnDepth - 1); });
        try {
            pair.join();
        } catch (...) {
            if (rightNode) rightNode->deleteSubtree();
            if (leftNode) leftNode->deleteSubtree();
            throw;
        }
    } else {
        rightNode = build(right, scratch, 0);
        leftNode = build(left, scratch, 0);
    }
    return new InnerNode(job.bounds, leftNode, rightNode, split.dim, SplitInfo::SAH, false);
}

}  // namespace FW
