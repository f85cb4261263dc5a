// RayGen.hpp -- ray production front end (src/rt/ray/RayGen.hpp, RayGen.cpp:45-74, 198-232, 582-602)
// and PixelTable (src/rt/ray/PixelTable.hpp).  Kernels live behind the C-ABI.
#pragma once
#include "RayBuffer.hpp"
#include "Scene.hpp"

namespace FW {

struct Mat4f {
    F32 m[16];  // row-major
};

// What beginFrame needs from CameraControls / GLContext in the reference (Renderer.cpp:473-477).
struct CameraView {
    Vec3f position;
    Mat4f nscreenToWorld;  // invert(fitToView * worldToClip)
    F32   cameraFar;
    S32   width, height;
};

class PixelTable {
public:
    PixelTable(void) : m_size(0, 0) {}
    void    setSize(const Vec2i& size);
    const Vec2i& getSize(void) const { return m_size; }
    Buffer& getIndexToPixel(void) { return m_indexToPixel; }
    Buffer& getPixelToIndex(void) { return m_pixelToIndex; }

private:
    Vec2i  m_size;
    Buffer m_indexToPixel;
    Buffer m_pixelToIndex;
};

class RayGen {
public:
    explicit RayGen(S32 maxBatchSize = 8 * 1024 * 1024) : m_maxBatchSize(maxBatchSize), m_aoStartIdx(0), m_shadowStartIdx(0), m_inLo(0), m_inHi(-1) {}

    // Multi-GPU extension: secondary rays are generated from the input slots [lo, hi) only (a rank's own primary hits); hi < 0 = all.
    void setInputRange(S32 lo, S32 hi) { m_inLo = lo; m_inHi = hi; }
    PixelTable& getPixelTable(void) { return m_pixelTable; }

    // RayGen::primary (RayGen.cpp:45-74): no batching
    void primary(RayBuffer& orays, const Vec3f& origin, const Mat4f& nscreenToWorld, S32 w, S32 h, float maxDist, U32 randomSeed = 0);
    // RayGen::ao (RayGen.cpp:198-232): false when all input rays have been consumed
    bool ao(RayBuffer& orays, RayBuffer& irays, Scene& scene, int numSamples, float maxDist, bool& newBatch, U32 randomSeed = 0);
    // RayGen::shadow (src/rt/ray/RayGen.cpp:114-150): numSamples any-hit rays per input ray towards the area light
    bool shadow(RayBuffer& orays, RayBuffer& irays, int numSamples, const Vec3f& lightPos, float lightRadius, bool& newBatch, U32 randomSeed = 0);

private:
    bool batching(S32 numInputRays, S32 numSamples, S32& startIdx, bool& newBatch, S32& lo, S32& hi);  // RayGen.cpp:582-602

    S32        m_maxBatchSize;
    PixelTable m_pixelTable;
    S32        m_aoStartIdx;
    S32        m_shadowStartIdx;
    S32        m_inLo, m_inHi;
};

}  // namespace FW
