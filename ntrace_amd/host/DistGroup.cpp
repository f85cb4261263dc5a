#include "DistGroup.hpp"

#include "ntrace_amd.h"

namespace FW {

static void check(int rc, const char* what)
{
    if (rc != NTR_OK) fail("DistGroup: %s failed: %s", what, ntr_last_error());
}

void DistGroup::uniqueId(char id[IdBytes]) { check(ntr_dist_unique_id(id), "ntr_dist_unique_id"); }

DistGroup::DistGroup(const char id[IdBytes], int rank, int world) : m_h(NULL), m_rank(rank), m_world(world)
{
    check(ntr_dist_init(id, rank, world, &m_h), "ntr_dist_init");
}

std::vector<DistGroup*> DistGroup::createAll(int numDevices)
{
    std::vector<NtrDist*> hs((size_t)numDevices, (NtrDist*)NULL);
    check(ntr_dist_init_all(numDevices, NULL, hs.data()), "ntr_dist_init_all");
    std::vector<DistGroup*> out;
    for (int i = 0; i < numDevices; i++) out.push_back(new DistGroup(hs[(size_t)i], i, numDevices));
    return out;
}

DistGroup::~DistGroup(void) { ntr_dist_destroy(m_h); }

void DistGroup::broadcastBVH(CudaBVH& bvh, int root)
{
    // header: layout + the three sizes, through a small device buffer
    Buffer hdr;
    hdr.resizeDiscard(4 * (S64)sizeof(S64));
    if (m_rank == root) {
        S64* h = (S64*)hdr.getMutablePtr();
        h[0] = (S64)bvh.getLayout();
        h[1] = bvh.getNodeBuffer().getSize();
        h[2] = bvh.getTriWoopBuffer().getSize();
        h[3] = bvh.getTriIndexBuffer().getSize();
    }
    check(ntr_dist_broadcast(m_h, hdr.getMutableCudaPtr(), hdr.getSize(), root, NULL), "ntr_dist_broadcast");
    check(ntr_stream_synchronize(NULL), "sync");
    const S64* h = (const S64*)hdr.getPtr();
    if (m_rank != root) {
        if ((S64)bvh.getLayout() != h[0]) fail("DistGroup::broadcastBVH: the receiving CudaBVH has layout %d, the root's has %d", (int)bvh.getLayout(), (int)h[0]);
        bvh.getNodeBuffer().resizeDiscard(h[1]);
        bvh.getTriWoopBuffer().resizeDiscard(h[2]);
        bvh.getTriIndexBuffer().resizeDiscard(h[3]);
        bvh.invalidateTraceFlags();
    }
    check(ntr_dist_broadcast_bvh(m_h, bvh.getNodeBuffer().getMutableCudaPtr(), h[1], bvh.getTriWoopBuffer().getMutableCudaPtr(), h[2],
                                 (int32_t*)bvh.getTriIndexBuffer().getMutableCudaPtr(), h[3], root, NULL), "ntr_dist_broadcast_bvh");
    check(ntr_stream_synchronize(NULL), "sync");
}

void DistGroup::gatherRecords(Renderer& renderer, Buffer& fullRecords, int root)
{
    RayBuffer& prim = renderer.getPrimaryRays();
    const S32 n = prim.getSize();
    if (m_rank == root) fullRecords.resizeDiscard((S64)n * (S64)sizeof(RayResult));
    check(ntr_dist_gather_records(m_h, (const NtrRayResult*)prim.getResultBuffer().getCudaPtr() + renderer.getShardLo(), n, 64,
                                  m_rank == root ? (NtrRayResult*)fullRecords.getMutableCudaPtr() : NULL, root, NULL), "ntr_dist_gather_records");
    check(ntr_stream_synchronize(NULL), "sync");
}

void DistGroup::gatherPixels(Renderer& renderer, Buffer& ownPixels, Buffer& fullPixels, int root)
{
    const S32 n = renderer.getPrimaryRays().getSize();
    if (m_scratch.getSize() < (S64)n * 4) m_scratch.resizeDiscard((S64)n * 4);
    if (m_rank == root) fullPixels.resizeDiscard((S64)n * 4);
    check(ntr_dist_gather_pixels(m_h, (const uint32_t*)ownPixels.getCudaPtr(),
                                 (const int32_t*)renderer.getRayGen().getPixelTable().getIndexToPixel().getCudaPtr(), n, 64,
                                 m_rank == root ? (uint32_t*)fullPixels.getMutableCudaPtr() : NULL, (uint32_t*)m_scratch.getMutableCudaPtr(), root, NULL),
          "ntr_dist_gather_pixels");
    check(ntr_stream_synchronize(NULL), "sync");
}

}  // namespace FW
