// ntr_dist.cpp -- native multi-GPU part of the C-ABI (include/ntrace_amd.h, "multi-GPU"): one process per GPU or one host
// thread per GPU, the BVH replicated by broadcast, a frame's rays sharded by screen tile, and ONE collective per frame -- the gather
// of the ranks' hit records or pixels to the root -- over RCCL (point-to-point sends over xGMI, grouped).  The reference has no
// multi-GPU code (its launches are synchronous on one context, src/framework/gpu/CudaKernel.cpp:188-221): this is new design
// (SURVEY.md 8(e)).  The partition arithmetic (ntr_frame_shard / ntr_frame_ao_batches) is shared with the Python plumbing
// (ntrace_amd/dist.py), which the gloo tests cover on the CPU.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first ntr_dist_* call): the library itself has no link dependency on it,
// loads on a box without it, and a process that already holds an RCCL (PyTorch) shares that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>

#include "ntr_internal.h"
#include "ntrace_amd.h"

static_assert(NTR_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "NtrDist unique id = ncclUniqueId");

namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommInitAll) commInitAll = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
};
Rccl g_rccl;
std::mutex g_rcclMu;

int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rcclMu);
    if (g_rccl.so) return NTR_OK;
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so) return ntr::set_error(NTR_ERR_HIP, "ntr_dist: cannot load librccl.so.1 (%s)", dlerror());
    Rccl r;
    r.so = so;
#define NTR_SYM(field, name) r.field = (decltype(r.field))dlsym(so, name); if (!r.field) return ntr::set_error(NTR_ERR_HIP, "ntr_dist: librccl has no %s", name)
    NTR_SYM(getUniqueId, "ncclGetUniqueId");
    NTR_SYM(commInitRank, "ncclCommInitRank");
    NTR_SYM(commInitAll, "ncclCommInitAll");
    NTR_SYM(commDestroy, "ncclCommDestroy");
    NTR_SYM(broadcast, "ncclBroadcast");
    NTR_SYM(send, "ncclSend");
    NTR_SYM(recv, "ncclRecv");
    NTR_SYM(groupStart, "ncclGroupStart");
    NTR_SYM(groupEnd, "ncclGroupEnd");
    NTR_SYM(errorString, "ncclGetErrorString");
#undef NTR_SYM
    g_rccl = r;
    return NTR_OK;
}

int rccl_fail(ncclResult_t e, const char* what)
{
    return ntr::set_error(NTR_ERR_HIP, "%s: %s", what, g_rccl.errorString ? g_rccl.errorString(e) : "RCCL error");
}
#define NTR_RCCL(call) do { ncclResult_t _e = (call); if (_e != ncclSuccess) return rccl_fail(_e, #call); } while (0)

}  // namespace

struct NtrDist {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

// ---- partition arithmetic (no device, no RCCL): what ntrace_amd/dist.py's FramePlan computes, in one place ----------------------
extern "C" int ntr_frame_shard(int32_t numPrimary, int32_t rank, int32_t world, int32_t align, int32_t* lo, int32_t* hi)
{
    if (!lo || !hi || numPrimary < 0 || world < 1 || rank < 0 || rank >= world || align < 1)
        return ntr::set_error(NTR_ERR_INVALID, "ntr_frame_shard: bad argument");
    // the rank-th of `world` contiguous ranges of whole `align`-ray blocks; the first (blocks % world) ranks get one block more
    const int64_t blocks = ((int64_t)numPrimary + align - 1) / align;
    const int64_t per = blocks / world, extra = blocks % world;
    const int64_t loB = (int64_t)rank * per + (rank < extra ? rank : extra);
    const int64_t hiB = loB + per + (rank < extra ? 1 : 0);
    const int64_t l = loB * align, h = hiB * align;
    *lo = (int32_t)(l < numPrimary ? l : numPrimary);
    *hi = (int32_t)(h < numPrimary ? h : numPrimary);
    return NTR_OK;
}

extern "C" int ntr_frame_ao_batches(int32_t lo, int32_t hi, int32_t samples, int32_t maxBatchRays, int32_t* first, int32_t* count, int32_t capacity,
                                    int32_t* numBatches)
{
    if (!numBatches || lo < 0 || hi < lo || samples < 0 || maxBatchRays < 1 || capacity < 0 || (capacity > 0 && (!first || !count)))
        return ntr::set_error(NTR_ERR_INVALID, "ntr_frame_ao_batches: bad argument");
    *numBatches = 0;
    if (samples == 0) return NTR_OK;
    // RayGen::batching (src/rt/ray/RayGen.cpp:582-602) over the rank's own input rays: at most maxBatchRays output rays per batch
    int32_t per = maxBatchRays / samples;
    if (per < 1) per = 1;
    int32_t n = 0;
    for (int64_t f = lo; f < hi; f += per, n++) {
        if (n < capacity) { first[n] = (int32_t)f; count[n] = (int32_t)((hi - f) < per ? (hi - f) : per); }
    }
    *numBatches = n;
    return NTR_OK;
}

extern "C" int ntr_dist_unique_id(char id[NTR_DIST_ID_BYTES])
{
    if (!id) return ntr::set_error(NTR_ERR_INVALID, "ntr_dist_unique_id: null argument");
    const int rc = rccl_load();
    if (rc != NTR_OK) return rc;
    ncclUniqueId u;
    NTR_RCCL(g_rccl.getUniqueId(&u));
    memcpy(id, u.internal, NTR_DIST_ID_BYTES);
    return NTR_OK;
}

extern "C" int ntr_dist_init(const char id[NTR_DIST_ID_BYTES], int32_t rank, int32_t world, NtrDist** out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return ntr::set_error(NTR_ERR_INVALID, "ntr_dist_init: bad argument");
    *out = nullptr;
    const int rc = rccl_load();
    if (rc != NTR_OK) return rc;
    NtrDist* d = new (std::nothrow) NtrDist();
    if (!d) return ntr::set_error(NTR_ERR_NOMEM, "ntr_dist_init: out of memory");
    d->rank = rank; d->world = world;
    hipError_t he = hipGetDevice(&d->device);
    if (he != hipSuccess) { delete d; return ntr::hip_fail(he, "hipGetDevice"); }
    ncclUniqueId u;
    memcpy(u.internal, id, NTR_DIST_ID_BYTES);
    const ncclResult_t e = g_rccl.commInitRank(&d->comm, world, u, rank);   // collective: every rank of the group calls it (own process or own thread)
    if (e != ncclSuccess) { delete d; return rccl_fail(e, "ncclCommInitRank"); }
    *out = d;
    return NTR_OK;
}

extern "C" int ntr_dist_init_all(int32_t numDevices, const int32_t* devices, NtrDist** out)
{
    if (numDevices < 1 || numDevices > 64 || !out) return ntr::set_error(NTR_ERR_INVALID, "ntr_dist_init_all: bad argument");
    for (int i = 0; i < numDevices; i++) out[i] = nullptr;
    const int rc = rccl_load();
    if (rc != NTR_OK) return rc;
    int prev = 0;
    NTR_HIP(hipGetDevice(&prev));
    ncclComm_t comms[64];
    int devs[64];
    for (int i = 0; i < numDevices; i++) devs[i] = devices ? devices[i] : i;
    {   // one process: a communicator per device, used by one host thread each (ncclCommInitAll visits the devices: put the caller's back either way)
        const ncclResult_t ie = g_rccl.commInitAll(comms, numDevices, devs);
        if (ie != ncclSuccess) { (void)hipSetDevice(prev); return rccl_fail(ie, "ncclCommInitAll"); }
    }
    for (int i = 0; i < numDevices; i++) {
        NtrDist* d = new (std::nothrow) NtrDist();
        if (!d) {   // give everything back: the groups made so far (with their communicators) and the communicators not yet wrapped
            for (int j = 0; j < i; j++) { (void)ntr_dist_destroy(out[j]); out[j] = nullptr; }
            for (int j = i; j < numDevices; j++) (void)g_rccl.commDestroy(comms[j]);
            (void)hipSetDevice(prev);
            return ntr::set_error(NTR_ERR_NOMEM, "ntr_dist_init_all: out of memory");
        }
        d->comm = comms[i]; d->rank = i; d->world = numDevices; d->device = devs[i];
        out[i] = d;
    }
    (void)hipSetDevice(prev);
    return NTR_OK;
}

extern "C" int ntr_dist_destroy(NtrDist* d)
{
    if (!d) return NTR_OK;
    if (d->comm && g_rccl.commDestroy) (void)g_rccl.commDestroy(d->comm);
    delete d;
    return NTR_OK;
}

extern "C" int ntr_dist_info(const NtrDist* d, int32_t* rank, int32_t* world)
{
    if (!d) return ntr::set_error(NTR_ERR_INVALID, "ntr_dist_info: null group");
    if (rank) *rank = d->rank;
    if (world) *world = d->world;
    return NTR_OK;
}

extern "C" int ntr_dist_broadcast(NtrDist* d, void* d_buf, int64_t bytes, int32_t root, void* stream)
{
    if (!d || bytes < 0 || root < 0 || root >= d->world || (bytes > 0 && !d_buf)) return ntr::set_error(NTR_ERR_INVALID, "ntr_dist_broadcast: bad argument");
    if (bytes == 0) return NTR_OK;
    NTR_RCCL(g_rccl.broadcast(d_buf, d_buf, (size_t)bytes, ncclUint8, root, d->comm, (hipStream_t)stream));
    return NTR_OK;
}

extern "C" int ntr_dist_broadcast_bvh(NtrDist* d, void* d_nodes, int64_t nodesBytes, void* d_triWoop, int64_t triWoopBytes, int32_t* d_triIndex,
                                      int64_t triIndexBytes, int32_t root, void* stream)
{
    // the three Compact buffers of the BVH built once on `root` (CudaAS::getNodeBuffer / getTriWoopBuffer / getTriIndexBuffer); every rank
    // passes buffers of the root's sizes (a rank learns them from ntr_dist_broadcast of a small header, or from its own build parameters)
    int rc = ntr_dist_broadcast(d, d_nodes, nodesBytes, root, stream);
    if (rc == NTR_OK) rc = ntr_dist_broadcast(d, d_triWoop, triWoopBytes, root, stream);
    if (rc == NTR_OK) rc = ntr_dist_broadcast(d, d_triIndex, triIndexBytes, root, stream);
    return rc;
}

// every rank's slice [lo_r, hi_r) x elemBytes of a frame-sized array -> the root's full array, at the slices' own offsets.
// Like every collective: ALL ranks of the group must make the matching call.  A rank that returns an argument error here (a null slice,
// a null destination on the root) has posted nothing, and its peers wait in ncclSend / ncclRecv for it: argument errors are programming
// errors that must be uniform across the ranks (the sizes and offsets are, by construction: ntr_frame_shard of the same numPrimary).
// `cuts` (world + 1 non-decreasing slot indices from 0 to numPrimary, or null): the ranks' ranges when a host cut the frame itself -- ranges of
// equal predicted cost (ntr_predict_block_costs) instead of equal ray counts; the same table on every rank.
static int gather_slices(NtrDist* d, const void* d_own, int32_t numPrimary, int32_t align, int32_t elemBytes, void* d_full, int32_t root, void* stream,
                         const char* who, const int32_t* cuts = nullptr)
{
    if (!d || numPrimary < 0 || align < 1 || root < 0 || root >= d->world) return ntr::set_error(NTR_ERR_INVALID, "%s: bad argument", who);
    if (cuts) {
        if (cuts[0] != 0 || cuts[d->world] != numPrimary) return ntr::set_error(NTR_ERR_INVALID, "%s: the cut table must run from 0 to numPrimary", who);
        for (int r = 0; r < d->world; r++)
            if (cuts[r] > cuts[r + 1]) return ntr::set_error(NTR_ERR_INVALID, "%s: the cut table must be non-decreasing", who);
    }
    auto range_of = [&](int r, int32_t* l, int32_t* h) -> int {
        if (cuts) { *l = cuts[r]; *h = cuts[r + 1]; return NTR_OK; }
        return ntr_frame_shard(numPrimary, r, d->world, align, l, h);
    };
    int32_t lo = 0, hi = 0;
    int rc = range_of(d->rank, &lo, &hi);
    if (rc != NTR_OK) return rc;
    if (hi > lo && !d_own) return ntr::set_error(NTR_ERR_INVALID, "%s: null slice", who);
    if (d->rank == root && numPrimary > 0 && !d_full) return ntr::set_error(NTR_ERR_INVALID, "%s: null destination on the root", who);
    hipStream_t s = (hipStream_t)stream;
    if (d->rank == root) {
        if (hi > lo && (const char*)d_own != (const char*)d_full + (size_t)lo * elemBytes)
            NTR_HIP(hipMemcpyAsync((char*)d_full + (size_t)lo * elemBytes, d_own, (size_t)(hi - lo) * elemBytes, hipMemcpyDeviceToDevice, s));
        NTR_RCCL(g_rccl.groupStart());
        ncclResult_t re = ncclSuccess;   // (a group once started is always ended)
        for (int r = 0; r < d->world && re == ncclSuccess; r++) {
            if (r == root) continue;
            int32_t l = 0, h = 0;
            (void)range_of(r, &l, &h);
            if (h > l) re = g_rccl.recv((char*)d_full + (size_t)l * elemBytes, (size_t)(h - l) * elemBytes, ncclUint8, r, d->comm, s);
        }
        const ncclResult_t ge = g_rccl.groupEnd();
        if (re != ncclSuccess) return rccl_fail(re, "ncclRecv");
        if (ge != ncclSuccess) return rccl_fail(ge, "ncclGroupEnd");
    } else if (hi > lo) {
        NTR_RCCL(g_rccl.send(d_own, (size_t)(hi - lo) * elemBytes, ncclUint8, root, d->comm, s));
    }
    return NTR_OK;
}

extern "C" int ntr_dist_gather_records(NtrDist* d, const NtrRayResult* d_ownRecords, int32_t numPrimary, int32_t align, NtrRayResult* d_fullRecords,
                                       int32_t root, void* stream)
{
    return gather_slices(d, d_ownRecords, numPrimary, align, (int32_t)sizeof(NtrRayResult), d_fullRecords, root, stream, "ntr_dist_gather_records");
}

extern "C" int ntr_dist_gather_records_cuts(NtrDist* d, const NtrRayResult* d_ownRecords, const int32_t* cuts, NtrRayResult* d_fullRecords, int32_t root,
                                            void* stream)
{
    if (!d || !cuts) return ntr::set_error(NTR_ERR_INVALID, "ntr_dist_gather_records_cuts: bad argument");
    return gather_slices(d, d_ownRecords, cuts[d->world], 1, (int32_t)sizeof(NtrRayResult), d_fullRecords, root, stream, "ntr_dist_gather_records_cuts", cuts);
}

extern "C" hipError_t ntr_launch_pixels_pack(const uint32_t* d_pixels, const int32_t* d_slotToPixel, int first, int count, uint32_t* d_out, hipStream_t s);
extern "C" hipError_t ntr_launch_pixels_unpack(const uint32_t* d_bySlot, const int32_t* d_slotToPixel, int count, uint32_t* d_pixels, hipStream_t s);

extern "C" int ntr_dist_gather_pixels(NtrDist* d, const uint32_t* d_ownPixels, const int32_t* d_slotToPixel, int32_t numPrimary, int32_t align,
                                      uint32_t* d_fullPixels, uint32_t* d_scratch, int32_t root, void* stream)
{
    // A rank's pixels are whole 8 x 8 tiles scattered over the image (its slice of the PixelTable index space): they travel packed in slot
    // order (4 bytes per primary ray: 8.3 MB for a whole 1080p frame) and are scattered to their pixel positions on the root.
    if (!d || !d_slotToPixel || !d_scratch || numPrimary < 0) return ntr::set_error(NTR_ERR_INVALID, "ntr_dist_gather_pixels: bad argument");
    int32_t lo = 0, hi = 0;
    int rc = ntr_frame_shard(numPrimary, d->rank, d->world, align, &lo, &hi);
    if (rc != NTR_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (hi > lo) {
        if (!d_ownPixels) return ntr::set_error(NTR_ERR_INVALID, "ntr_dist_gather_pixels: null framebuffer");
        const hipError_t e = ntr_launch_pixels_pack(d_ownPixels, d_slotToPixel, lo, hi - lo, d_scratch + lo, s);
        if (e != hipSuccess) return ntr::hip_fail(e, "pixels_pack launch");
    }
    rc = gather_slices(d, d_scratch + lo, numPrimary, align, 4, d_scratch, root, stream, "ntr_dist_gather_pixels");
    if (rc != NTR_OK) return rc;
    if (d->rank == root && numPrimary > 0) {
        if (!d_fullPixels) return ntr::set_error(NTR_ERR_INVALID, "ntr_dist_gather_pixels: null destination on the root");
        const hipError_t e = ntr_launch_pixels_unpack(d_scratch, d_slotToPixel, numPrimary, d_fullPixels, s);
        if (e != hipSuccess) return ntr::hip_fail(e, "pixels_unpack launch");
    }
    return NTR_OK;
}
