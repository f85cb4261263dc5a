// CudaBVH.cpp -- host BVH -> BVHLayout_Compact buffers (+ bvhcache (de)serialisation).
#include "CudaBVH.hpp"

#include <istream>
#include <ostream>
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
// a terminator.
void CudaBVH::createCompact(const BVH& bvh, int nodeOffsetSizeDiv)
{
    struct StackEntry {
        const BVHNode* node;
        S32            idx;  // index of the node's first int4
    };
    struct Int4 { S32 x, y, z, w; };

    std::vector<Int4> nodeData(4);
    std::vector<Int4> triWoopData;
    std::vector<S32>  triIndexData;
    std::vector<StackEntry> stack(1, StackEntry{bvh.getRoot(), 0});

    while (!stack.empty()) {
        StackEntry e = stack.back();
        stack.pop_back();
        if (e.node->getNumChildNodes() != 2) fail("CudaBVH::createCompact: inner node without 2 children");
        const AABB* cbox[2];
        int cidx[2];

        for (int i = 0; i < 2; i++) {
            const BVHNode* child = e.node->getChildNode(i);
            cbox[i] = &child->m_bounds;
            if (!child->isLeaf()) {
                cidx[i] = (int)(nodeData.size() * sizeof(Int4)) / nodeOffsetSizeDiv;
                stack.push_back(StackEntry{child, (S32)nodeData.size()});
                nodeData.resize(nodeData.size() + 4);
                continue;
            }
            const LeafNode* leaf = static_cast<const LeafNode*>(child);
            cidx[i] = ~(int)triWoopData.size();
            for (int j = leaf->m_lo; j < leaf->m_hi; j++) {
                woopifyTri(bvh, j);
                if (m_woop[0].x == 0.0f) m_woop[0].x = 0.0f;  // -0 would alias the terminator (:627-628)
                Int4 w[3];
                memcpy(w, m_woop, sizeof(w));
                triWoopData.insert(triWoopData.end(), w, w + 3);
                triIndexData.push_back(bvh.getTriIndices()[j]);
                triIndexData.push_back(0);
                triIndexData.push_back(0);
            }
            const S32 nz = (S32)0x80000000;
            triWoopData.push_back(Int4{nz, nz, nz, nz});  // Array<Vec4i>::add(0x80000000) -> Vec4i(a) sets all four (:641)
            triIndexData.push_back(0);
        }

        const InnerNode* eN = static_cast<const InnerNode*>(e.node);
        Int4* dst = &nodeData[e.idx];
        dst[0] = Int4{(S32)floatToBits(cbox[0]->min().x), (S32)floatToBits(cbox[0]->max().x), (S32)floatToBits(cbox[0]->min().y), (S32)floatToBits(cbox[0]->max().y)};
        dst[1] = Int4{(S32)floatToBits(cbox[1]->min().x), (S32)floatToBits(cbox[1]->max().x), (S32)floatToBits(cbox[1]->min().y), (S32)floatToBits(cbox[1]->max().y)};
        dst[2] = Int4{(S32)floatToBits(cbox[0]->min().z), (S32)floatToBits(cbox[0]->max().z), (S32)floatToBits(cbox[1]->min().z), (S32)floatToBits(cbox[1]->max().z)};
        dst[3] = Int4{cidx[0], cidx[1], (S32)eN->getSplitInfo().getBitCode(), 0};
    }

    m_nodes.set(nodeData.data(), (S64)(nodeData.size() * sizeof(Int4)));
    m_triWoop.set(triWoopData.data(), (S64)(triWoopData.size() * sizeof(Int4)));
    m_triIndex.set(triIndexData.data(), (S64)(triIndexData.size() * sizeof(S32)));
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

void CudaBVH::woopifyTri(const BVH& bvh, int idx)
{
    const Vec3i* triVtxIndex = (const Vec3i*)bvh.getScene()->getTriVtxIndexBuffer().getPtr();
    const Vec3f* vtxPos = (const Vec3f*)bvh.getScene()->getVtxPosBuffer().getPtr();
    const Vec3i& inds = triVtxIndex[bvh.getTriIndices()[idx]];
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

    m_woop[0] = Vec4f(mtx.m[2][0], mtx.m[2][1], mtx.m[2][2], -mtx.m[2][3]);
    m_woop[1] = Vec4f(mtx.m[0][0], mtx.m[0][1], mtx.m[0][2], mtx.m[0][3]);
    m_woop[2] = Vec4f(mtx.m[1][0], mtx.m[1][1], mtx.m[1][2], mtx.m[1][3]);
}

}  // namespace FW
