// CudaBVH.cpp -- host BVH -> BVHLayout_Compact buffers (+ bvhcache (de)serialisation).
#include "CudaBVH.hpp"

#include <cstring>
#include <istream>
#include <ostream>
#include <thread>

#include "Threads.hpp"
#include <utility>
#include <vector>

namespace FW {

CudaBVH::CudaBVH(const BVH& bvh, BVHLayout layout) : m_layout(layout), m_flags(0), m_flagsValid(false)
{
    // This fork builds Compact only (CudaBVH.cpp:65-82 asserts on anything else).
    // The reference additionally permutes non-root node slots at random
    // (METHOD RND, CudaBVH.cpp:71-72, seeded from the wall clock); hit records do
    // not depend on node numbering, so the deterministic DFS emission order is kept.
    if (layout != BVHLayout_Compact) fail("CudaBVH: only BVHLayout_Compact is supported");
    createCompact(bvh, 1);
}

CudaBVH::CudaBVH(std::istream& in) : m_flags(0), m_flagsValid(false)
{
    S32 layout = 0;
    in.read((char*)&layout, sizeof(layout));
    m_layout = (BVHLayout)layout;
    m_nodes.readFromStream(in);
    m_triWoop.readFromStream(in);
    m_triIndex.readFromStream(in);
    if (!in) setError("CudaBVH: truncated stream");
}

void CudaBVH::serialize(std::ostream& out)
{
    // S32 layout + 3 x (S64 size + bytes), little endian -- the reference's
    // bvhcache/*.dat format (CudaBVH.cpp:118-125, Buffer.cpp:365-381).
    S32 layout = (S32)m_layout;
    out.write((const char*)&layout, sizeof(layout));
    m_nodes.writeToStream(out);
    m_triWoop.writeToStream(out);
    m_triIndex.writeToStream(out);
}

U32 CudaBVH::getTraceFlags(void)
{
    if (!m_flagsValid) {
        uint32_t f = 0;
        if (m_nodes.getSize() >= 64 &&
            ntr_bvh_validate(m_nodes.getCudaPtr(), m_nodes.getSize(), &f, NULL) != NTR_OK)
            fail("CudaBVH: ntr_bvh_validate failed: %s", ntr_last_error());
        m_flags = f;
        m_flagsValid = true;
    }
    return m_flags;
}

// Emission order of CudaBVH::createCompact (CudaBVH.cpp:594-652): explicit stack,
// pop -> for child 0 then child 1: an inner child takes the next 64-B slot at once
// (so siblings are adjacent) and is pushed; a leaf child appends its triangles and
// a terminator.  The layout (which slot every node and every leaf gets) is fixed by that
// walk, which only chases pointers: it is done first and records where everything goes;
// the contents -- child boxes, Woop rows (a 4x4 cofactor inverse per triangle), indices --
// are then written by all host threads, every record being independent of the others.
void CudaBVH::createCompact(const BVH& bvh, int nodeOffsetSizeDiv)
{
    struct Int4 { S32 x, y, z, w; };
    struct InnerRec { const BVHNode* node; S32 idx, c0, c1; };  // idx = index of the node's first int4
    struct LeafRec { const LeafNode* leaf; S32 ofs; };           // ofs = float4 index of the leaf's first Woop row

    // ---- 1. layout ---------------------------------------------------------------------------------------------
    std::vector<InnerRec> inner;
    std::vector<LeafRec> leaves;
    std::vector<std::pair<const BVHNode*, S32> > stack(1, std::make_pair((const BVHNode*)bvh.getRoot(), (S32)0));
    S64 numNodeInt4 = 4, numWoop = 0;
    while (!stack.empty()) {
        const std::pair<const BVHNode*, S32> e = stack.back();
        stack.pop_back();
        if (e.first->getNumChildNodes() != 2) fail("CudaBVH::createCompact: inner node without 2 children");
        InnerRec rec = {e.first, e.second, 0, 0};
        for (int i = 0; i < 2; i++) {
            const BVHNode* child = e.first->getChildNode(i);
            S32 cidx;
            if (!child->isLeaf()) {
                cidx = (S32)((numNodeInt4 * (S64)sizeof(Int4)) / nodeOffsetSizeDiv);
                stack.push_back(std::make_pair(child, (S32)numNodeInt4));
                numNodeInt4 += 4;
            } else {
                const LeafNode* leaf = static_cast<const LeafNode*>(child);
                cidx = ~(S32)numWoop;
                leaves.push_back(LeafRec{leaf, (S32)numWoop});
                numWoop += (S64)(leaf->m_hi - leaf->m_lo) * 3 + 1;  // three rows per triangle + the terminator
            }
            (i ? rec.c1 : rec.c0) = cidx;
        }
        inner.push_back(rec);
        if (numNodeInt4 * (S64)sizeof(Int4) > 0x7FFFFFFFll || numWoop > 0x7FFFFFFFll) fail("CudaBVH::createCompact: BVH too large for 32-bit offsets");
    }

    // ---- 2. contents, in parallel ------------------------------------------------------------------------------
    // straight into the three Buffers' host memory: no second copy of 1.4 GB for a 10 M-triangle scene
    m_nodes.resizeDiscard(numNodeInt4 * (S64)sizeof(Int4));
    m_triWoop.resizeDiscard(numWoop * (S64)sizeof(Int4));
    m_triIndex.resizeDiscard(numWoop * (S64)sizeof(S32));
    Int4* nodeData = (Int4*)m_nodes.getMutablePtr();
    Int4* triWoopData = (Int4*)m_triWoop.getMutablePtr();
    S32* triIndexData = (S32*)m_triIndex.getMutablePtr();
    const Vec3i* triVtxIndex = (const Vec3i*)bvh.getScene()->getTriVtxIndexBuffer().getPtr();
    const Vec3f* vtxPos = (const Vec3f*)bvh.getScene()->getVtxPosBuffer().getPtr();
    const std::vector<S32>& triIndices = bvh.getTriIndices();

    auto fillInner = [&](size_t b, size_t e) {
        for (size_t k = b; k < e; k++) {
            const InnerRec& r = inner[k];
            const AABB& b0 = r.node->getChildNode(0)->m_bounds;
            const AABB& b1 = r.node->getChildNode(1)->m_bounds;
            Int4* dst = &nodeData[(size_t)r.idx];
            dst[0] = Int4{(S32)floatToBits(b0.min().x), (S32)floatToBits(b0.max().x), (S32)floatToBits(b0.min().y), (S32)floatToBits(b0.max().y)};
            dst[1] = Int4{(S32)floatToBits(b1.min().x), (S32)floatToBits(b1.max().x), (S32)floatToBits(b1.min().y), (S32)floatToBits(b1.max().y)};
            dst[2] = Int4{(S32)floatToBits(b0.min().z), (S32)floatToBits(b0.max().z), (S32)floatToBits(b1.min().z), (S32)floatToBits(b1.max().z)};
            dst[3] = Int4{r.c0, r.c1, (S32)static_cast<const InnerNode*>(r.node)->getSplitInfo().getBitCode(), 0};
        }
    };
    auto fillLeaves = [&](size_t b, size_t e) {
        for (size_t k = b; k < e; k++) {
            const LeafRec& r = leaves[k];
            S32 o = r.ofs;
            for (int j = r.leaf->m_lo; j < r.leaf->m_hi; j++, o += 3) {
                Vec4f w[3];
                woopify(triVtxIndex, vtxPos, triIndices[j], w);
                if (w[0].x == 0.0f) w[0].x = 0.0f;  // -0 would alias the terminator (:627-628)
                memcpy(&triWoopData[(size_t)o], w, sizeof(w));
                triIndexData[(size_t)o] = triIndices[j];
                triIndexData[(size_t)o + 1] = 0;
                triIndexData[(size_t)o + 2] = 0;
            }
            const S32 nz = (S32)0x80000000;
            triWoopData[(size_t)o] = Int4{nz, nz, nz, nz};  // Array<Vec4i>::add(0x80000000) -> Vec4i(a) sets all four (:641)
            triIndexData[(size_t)o] = 0;
        }
    };
    unsigned threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    if (threads > 64) threads = 64;
    if (inner.size() < 50000) threads = 1;
    if (threads == 1) {
        fillInner(0, inner.size());
        fillLeaves(0, leaves.size());
    } else {
        ThreadGroup pool;
        for (unsigned t = 0; t < threads; t++)
            pool.spawn([&, t]() {
                fillInner(inner.size() * t / threads, inner.size() * (t + 1) / threads);
                fillLeaves(leaves.size() * t / threads, leaves.size() * (t + 1) / threads);
            });
        pool.join();
    }

    m_flagsValid = false;
}

// ---- woopifyTri (CudaBVH.cpp:668-687) ---------------------------------------------
// Inverse of the 4x4 matrix with columns (v0-v2,0) (v1-v2,0) (n,0) (v2,1) by
// cofactors, following MatrixBase::inverted / detImpl<3> (Math.hpp:993-1046):
// r(i,j) = det(minor without row j, col i) * sign ; d = sum r(i,j)*m(j,i) over all
// i,j (= 4*det) ; result = r * (1/d) * 4.
namespace {
struct M4 {
    F32 m[4][4];  // m[row][col]
};

F32 det3(const F32 v[3][3])
{
    return v[0][0] * v[1][1] * v[2][2] - v[0][0] * v[1][2] * v[2][1] + v[1][0] * v[2][1] * v[0][2] -
           v[1][0] * v[2][2] * v[0][1] + v[2][0] * v[0][1] * v[1][2] - v[2][0] * v[0][2] * v[1][1];
}

M4 inverted(const M4& a)
{
    M4 r;
    F32 d = 0.0f;
    F32 si = 1.0f;
    for (int i = 0; i < 4; i++) {
        F32 sj = si;
        for (int j = 0; j < 4; j++) {
            F32 sub[3][3];
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 3; l++) sub[k][l] = a.m[(k < j) ? k : k + 1][(l < i) ? l : l + 1];
            F32 dd = det3(sub) * sj;
            r.m[i][j] = dd;
            d += dd * a.m[j][i];
            sj = -sj;
        }
        si = -si;
    }
    F32 rd = 1.0f / d;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = r.m[i][j] * rd * 4.0f;
    return r;
}
}  // namespace

// Woop rows of scene triangle `tri` (CudaBVH::woopifyTri, CudaBVH.cpp:668-687).
void CudaBVH::woopify(const Vec3i* triVtxIndex, const Vec3f* vtxPos, S32 tri, Vec4f (&out)[3])
{
    const Vec3i& inds = triVtxIndex[tri];
    const Vec3f& v0 = vtxPos[inds.x];
    const Vec3f& v1 = vtxPos[inds.y];
    const Vec3f& v2 = vtxPos[inds.z];

    const Vec3f c0 = v0 - v2, c1 = v1 - v2, c2 = cross(v0 - v2, v1 - v2);
    M4 mtx;
    const Vec3f cols[4] = {c0, c1, c2, v2};
    for (int c = 0; c < 4; c++) {
        mtx.m[0][c] = cols[c].x;
        mtx.m[1][c] = cols[c].y;
        mtx.m[2][c] = cols[c].z;
        mtx.m[3][c] = (c == 3) ? 1.0f : 0.0f;
    }
    mtx = inverted(mtx);

    out[0] = Vec4f(mtx.m[2][0], mtx.m[2][1], mtx.m[2][2], -mtx.m[2][3]);
    out[1] = Vec4f(mtx.m[0][0], mtx.m[0][1], mtx.m[0][2], mtx.m[0][3]);
    out[2] = Vec4f(mtx.m[1][0], mtx.m[1][1], mtx.m[1][2], mtx.m[1][3]);
}

}  // namespace FW
