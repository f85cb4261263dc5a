#include "RayGen.hpp"

#include "Random.hpp"
#include "ntrace_amd.h"

namespace FW {

static void check(int rc, const char* what)
{
    if (rc != NTR_OK) fail("RayGen: %s failed: %s", what, ntr_last_error());
}

void PixelTable::setSize(const Vec2i& size)
{
    if (size.x == m_size.x && size.y == m_size.y) return;
    m_size = size;
    m_indexToPixel.resizeDiscard((S64)size.x * size.y * sizeof(S32));
    m_pixelToIndex.resizeDiscard((S64)size.x * size.y * sizeof(S32));
    check(ntr_pixel_table(size.x, size.y, (int32_t*)m_indexToPixel.getMutableCudaPtr(),
                          (int32_t*)m_pixelToIndex.getMutableCudaPtr(), NULL), "ntr_pixel_table");
}

void RayGen::primary(RayBuffer& orays, const Vec3f& origin, const Mat4f& nscreenToWorld, S32 w, S32 h, float maxDist, U32 randomSeed)
{
    m_pixelTable.setSize(Vec2i(w, h));
    orays.resize(w * h);
    orays.setNeedClosestHit(true);
    const float o[3] = {origin.x, origin.y, origin.z};
    const U32 kernelSeed = (randomSeed != 0) ? Random(randomSeed).getU32() : 0;  // RayGen.cpp:66
    check(ntr_raygen_primary((NtrRay*)orays.getRayBuffer().getMutableCudaPtr(),
                             (int32_t*)orays.getIDToSlotBuffer().getMutableCudaPtr(),
                             (int32_t*)orays.getSlotToIDBuffer().getMutableCudaPtr(),
                             (const int32_t*)m_pixelTable.getIndexToPixel().getCudaPtr(), o, nscreenToWorld.m, w, h, maxDist,
                             kernelSeed, NULL), "ntr_raygen_primary");
    check(ntr_stream_synchronize(NULL), "sync");
}

bool RayGen::ao(RayBuffer& orays, RayBuffer& irays, Scene& scene, int numSamples, float maxDist, bool& newBatch, U32 randomSeed)
{
    S32 lo, hi;
    if (!batching(irays.getSize(), numSamples, m_aoStartIdx, newBatch, lo, hi)) return false;
    orays.resize((hi - lo) * numSamples);
    orays.setNeedClosestHit(false);
    check(ntr_raygen_ao((NtrRay*)orays.getRayBuffer().getMutableCudaPtr(), (int32_t*)orays.getIDToSlotBuffer().getMutableCudaPtr(),
                        (int32_t*)orays.getSlotToIDBuffer().getMutableCudaPtr(), (const NtrRay*)irays.getRayBuffer().getCudaPtr(),
                        (const NtrRayResult*)irays.getResultBuffer().getCudaPtr(),
                        (const float*)scene.getTriNormalBuffer().getCudaPtr(), lo, hi - lo, numSamples, maxDist,
                        Random(randomSeed).getU32() /* RayGen.cpp:220 */, NULL), "ntr_raygen_ao");
    check(ntr_stream_synchronize(NULL), "sync");
    return true;
}

bool RayGen::shadow(RayBuffer& orays, RayBuffer& irays, int numSamples, const Vec3f& lightPos, float lightRadius, bool& newBatch, U32 randomSeed)
{
    S32 lo, hi;
    if (!batching(irays.getSize(), numSamples, m_shadowStartIdx, newBatch, lo, hi)) return false;
    orays.resize((hi - lo) * numSamples);
    orays.setNeedClosestHit(false);
    const float lp[3] = {lightPos.x, lightPos.y, lightPos.z};
    check(ntr_raygen_shadow((NtrRay*)orays.getRayBuffer().getMutableCudaPtr(), (int32_t*)orays.getIDToSlotBuffer().getMutableCudaPtr(),
                            (int32_t*)orays.getSlotToIDBuffer().getMutableCudaPtr(), (const NtrRay*)irays.getRayBuffer().getCudaPtr(),
                            (const NtrRayResult*)irays.getResultBuffer().getCudaPtr(), lo, hi - lo, numSamples, lp, lightRadius,
                            Random(randomSeed).getU32() /* RayGen.cpp:139 */, NULL), "ntr_raygen_shadow");
    check(ntr_stream_synchronize(NULL), "sync");
    return true;
}

bool RayGen::batching(S32 numInputRays, S32 numSamples, S32& startIdx, bool& newBatch, S32& lo, S32& hi)
{
    const S32 end = (m_inHi >= 0) ? FW::min(m_inHi, numInputRays) : numInputRays;   // (a rank's own input range, or everything)
    if (newBatch) {
        newBatch = false;
        startIdx = FW::min(m_inLo, end);
    }
    if (startIdx >= end) return false;
    lo = startIdx;
    hi = FW::min(end, lo + m_maxBatchSize / numSamples);
    startIdx = hi;
    return true;
}

}  // namespace FW
