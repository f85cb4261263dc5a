// DistGroup.hpp -- the multi-GPU group of one rank; no counterpart in the reference, which drives one device with
// synchronous launches (src/framework/gpu/CudaKernel.cpp:188-221).  One DistGroup per GPU: in its own process (uniqueId() on the
// root, the 128 bytes handed to every rank, then the constructor on every rank) or in its own host thread of one process
// (createAll()).  Forwards to the C-ABI (ntr_dist_*), which binds RCCL at run time.
//
//   rank r:  ntr_set_device(r); DistGroup g(id, r, world);
//            Renderer ren(builder); ren.setScene(&scene); ren.setShard(r, world);
//            if (r == 0) g.broadcastBVH(*ren.getCudaBVH()); else { CudaBVH* b = new CudaBVH(layout); g.broadcastBVH(*b); ren.adoptCudaBVH(b); }
//            per frame: ren.beginFrame(cam); while (ren.nextBatch()) { ren.traceBatch(); ren.updateResult(pixels, ...); }
//                       g.gatherPixels(ren, pixels, fullPixels);      // the frame's one collective (or gatherRecords)
#pragma once
#include <vector>

#include "Renderer.hpp"

struct NtrDist;

namespace FW {

class DistGroup {
public:
    enum { IdBytes = 128 };
    static void uniqueId(char id[IdBytes]);                                   // root only (ncclGetUniqueId)
    DistGroup(const char id[IdBytes], int rank, int world);                   // collective; on the calling thread's current device
    static std::vector<DistGroup*> createAll(int numDevices);                 // one process, one thread per GPU (ncclCommInitAll)
    ~DistGroup(void);

    int getRank(void) const { return m_rank; }
    int getWorld(void) const { return m_world; }

    // The root's BVH to every rank: sizes first, then the three Compact buffers.  Non-roots pass an empty CudaBVH of the root's layout.
    void broadcastBVH(CudaBVH& bvh, int root = 0);
    // The frame's one collective: every rank's hit records of its primary range / its pixels -> the root's full-frame buffer
    // (resized on the root; untouched elsewhere).  Blocks until the gather has completed.
    void gatherRecords(Renderer& renderer, Buffer& fullRecords, int root = 0);
    void gatherPixels(Renderer& renderer, Buffer& ownPixels, Buffer& fullPixels, int root = 0);

private:
    explicit DistGroup(NtrDist* h, int rank, int world) : m_h(h), m_rank(rank), m_world(world) {}
    DistGroup(const DistGroup&);
    DistGroup& operator=(const DistGroup&);

    NtrDist* m_h;
    int      m_rank, m_world;
    Buffer   m_scratch;
};

}  // namespace FW
