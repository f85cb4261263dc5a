#include "Renderer.hpp"

#include <sys/stat.h>

#include <cstdio>
#include <fstream>

#include "Hash.hpp"

namespace FW {

// Renderer::Renderer (Renderer.cpp:44-94): m_raygen(1 << 20), Platform("GPU") with leaf preferences (1,1).
Renderer::Renderer(const String& builder)
    : m_builder(builder), m_raygen(1 << 20), m_enableRandom(false), m_scene(NULL), m_mesh(NULL), m_ownsScene(false), m_cameraFar(0.0f), m_newBatch(true),
      m_batchRays(NULL), m_batchStart(0), m_accelStruct(NULL), m_cachePath("bvhcache"), m_cacheDataStructure(false),
      m_predictSecondary(true), m_leafDepthOf(NULL), m_secondaryHint(NULL), m_shardRank(0), m_shardWorld(1), m_shardLo(0), m_shardHi(0)
{
    m_cudaTracer = new CudaBVHTracer();
    m_cudaTracer->setScene(NULL);
    m_platform = Platform("GPU");
    m_platform.setLeafPreferences(1, 1);
    m_buildParams.builder = (builder == "HLBVH") ? "SAHBVH" : builder;
}

Renderer::~Renderer(void)
{
    static_cast<CudaBVHTracer*>(m_cudaTracer)->setSchedHint(NULL);
    if (m_secondaryHint) (void)ntr_sched_hint_destroy(m_secondaryHint);
    delete m_accelStruct;
    delete m_cudaTracer;
    if (m_ownsScene) delete m_scene;    // (the reference's destructor calls setMesh(NULL), Renderer.cpp:96-98)
}

void Renderer::setScene(Scene* scene)
{
    if (scene == m_scene) return;
    invalidateBVH();
    if (m_ownsScene) { delete m_scene; m_ownsScene = false; m_mesh = NULL; }
    m_scene = scene;
    m_cudaTracer->setScene(scene);
}

void Renderer::setMesh(const WavefrontMesh* mesh)  // Renderer.cpp:98-132
{
    if (mesh == m_mesh) return;         // same mesh => done
    if (m_ownsScene) delete m_scene;    // deinit scene and BVH
    m_scene = NULL;
    m_ownsScene = false;
    invalidateBVH();
    m_mesh = mesh;
    if (mesh) {                         // create scene
        m_scene = mesh->createScene();
        m_ownsScene = true;
    }
    m_cudaTracer->setScene(m_scene);
}

void Renderer::setShard(int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) fail("Renderer::setShard: rank %d of %d", rank, world);
    m_shardRank = rank;
    m_shardWorld = world;
}

void Renderer::setParams(const Params& params)  // Renderer.cpp:138-143
{
    m_params = params;
    m_cudaTracer->setKernel(params.kernelName);
}

// "<cachePath>/<hash>_<builder>.dat" with the hash over scene, platform, build parameters, layout and data structure name
// (Renderer.cpp:173-180); this backend's only data structure is "BVH".
String Renderer::getCacheFileName(void)
{
    char name[64];
    const U32 h = hashBits(m_scene ? m_scene->hash() : 0u, m_platform.computeHash(), m_buildParams.computeHash(),
                           (U32)m_cudaTracer->getDesiredBVHLayout(), hashString("BVH"));
    snprintf(name, sizeof(name), "/%08x_", h);
    return m_cachePath + name + m_builder + ".dat";
}

// Renderer::getCudaBVH (Renderer.cpp:147-305) without the OcclusionBVH branch.
CudaAS* Renderer::getCudaBVH(void)
{
    BVHLayout layout = m_cudaTracer->getDesiredBVHLayout();
    if (!m_scene || (m_accelStruct && m_accelStruct->getLayout() == layout)) return m_accelStruct;
    delete m_accelStruct;
    m_accelStruct = NULL;
    const String cacheFile = getCacheFileName();
    if (m_cacheDataStructure) {  // cache file exists => import (:184-191)
        std::ifstream in(cacheFile.c_str(), std::ios::binary);
        if (in) {
            CudaBVH* cached = new CudaBVH(in);
            if (!hasError() && cached->getLayout() == layout) { m_accelStruct = cached; return m_accelStruct; }
            clearError();
            delete cached;
        }
    }
    if (m_builder == "HLBVH") {
        HLBVHParams params;  // Renderer.cpp:203-207 asks for hlbvh = true, hlbvhBits = 4; this backend
        params.hlbvh = false;  // provides the plain LBVH pipeline of the same builder
        params.leafSize = 8;
        params.epsilon = 0.001f;
        m_accelStruct = new HLBVHBuilder(m_scene, m_platform, params);
    } else {
        BVH bvh(m_scene, m_platform, m_buildParams);
        m_accelStruct = new CudaBVH(bvh, layout);
        failIfError();
    }
    if (m_cacheDataStructure) {  // write to cache (:293-299); a failure to write is not an error
        ::mkdir(m_cachePath.c_str(), 0777);
        std::ofstream out(cacheFile.c_str(), std::ios::binary);
        if (out) m_accelStruct->serialize(out);
    }
    return m_accelStruct;
}

// Renderer::beginFrame (Renderer.cpp:405-497)
void Renderer::beginFrame(const CameraView& camera)
{
    if (!m_scene) fail("Renderer: no scene");
    m_cudaTracer->setBVH(getCudaBVH());
    m_raygen.primary(m_primaryRays, camera.position, camera.nscreenToWorld, camera.width, camera.height, camera.cameraFar, 0);
    // this rank's screen tiles: a contiguous 64-aligned range of the primary slots (the whole frame for one rank)
    int32_t lo = 0, hi = 0;
    if (ntr_frame_shard(m_primaryRays.getSize(), m_shardRank, m_shardWorld, 64, &lo, &hi) != NTR_OK) fail("Renderer: %s", ntr_last_error());
    m_shardLo = lo;
    m_shardHi = hi;
    m_raygen.setInputRange(lo, hi);
    if (m_params.rayType != RayType_Primary)  // :482-488
        static_cast<CudaBVHTracer*>(m_cudaTracer)->traceRange(m_primaryRays, lo, hi - lo);
    m_cameraFar = camera.cameraFar;
    m_newBatch = true;
    m_batchRays = NULL;
    m_batchStart = 0;
}

// Renderer::nextBatch (Renderer.cpp:501-564)
bool Renderer::nextBatch(void)
{
    if (m_batchRays) m_batchStart += (m_batchRays == &m_primaryRays) ? (m_shardHi - m_shardLo) : m_batchRays->getSize();
    m_batchRays = NULL;
    switch (m_params.rayType) {
    case RayType_Primary:
        if (!m_newBatch) return false;
        m_newBatch = false;
        m_batchRays = &m_primaryRays;
        break;
    case RayType_AO:
        if (!m_raygen.ao(m_secondaryRays, m_primaryRays, *m_scene, m_params.numSamples, m_params.aoRadius, m_newBatch, 0)) return false;
        m_batchRays = &m_secondaryRays;
        break;
    case RayType_Diffuse:
        if (!m_raygen.ao(m_secondaryRays, m_primaryRays, *m_scene, m_params.numSamples, m_cameraFar, m_newBatch, 0)) return false;
        m_secondaryRays.setNeedClosestHit(true);
        m_batchRays = &m_secondaryRays;
        break;
    default:
        return false;
    }
    // Renderer.cpp:559-563
    if (m_params.sortSecondary && m_params.rayType != RayType_Primary) m_batchRays->mortonSort();
    // AO batches: a dispatch hint from the tree (not for sorted batches: their blocks no longer follow the input rays)
    if (m_params.rayType == RayType_AO && m_predictSecondary && !m_params.sortSecondary && m_batchRays->getSize() > 0) predictSecondaryOrder();
    return true;
}

void Renderer::predictSecondaryOrder(void)
{
    CudaAS* as = getCudaBVH();
    const int numTris = m_scene->getNumTriangles();
    if (numTris < 1) return;
    if (m_leafDepthOf != as || m_leafDepth.getSize() != (S64)numTris * 4) {   // once per BVH
        m_leafDepth.resizeDiscard((S64)numTris * 4);
        int rc = ntr_bvh_leaf_depths(as->getNodeBuffer().getCudaPtr(), as->getNodeBuffer().getSize(), as->getTriWoopBuffer().getCudaPtr(),
                                     as->getTriWoopBuffer().getSize(), (const int32_t*)as->getTriIndexBuffer().getCudaPtr(), numTris,
                                     (int32_t*)m_leafDepth.getMutableCudaPtr(), NULL, NULL);
        if (rc != NTR_OK) fail("Renderer: %s", ntr_last_error());
        m_leafDepthOf = as;
    }
    const int ns = m_params.numSamples;
    const int first = m_shardLo + m_batchStart / ns, count = (int)(m_batchRays->getSize() / ns);
    const int blocks = (int)((m_batchRays->getSize() + 255) / 256);
    if (m_blockCost.getSize() < (S64)blocks * 4) m_blockCost.resizeDiscard((S64)blocks * 4);   // grows only: a frame's short last batch reallocates nothing
    int rc = ntr_secondary_block_costs((const NtrRayResult*)m_primaryRays.getResultBuffer().getCudaPtr(), first, count, ns,
                                       (const int32_t*)m_leafDepth.getCudaPtr(), numTris, (uint32_t*)m_blockCost.getMutableCudaPtr(), NULL);
    if (rc == NTR_OK && !m_secondaryHint) rc = ntr_sched_hint_create(&m_secondaryHint);
    if (rc == NTR_OK) rc = ntr_sched_hint_predict(m_secondaryHint, (const uint32_t*)m_blockCost.getCudaPtr(), blocks, NULL);
    if (rc != NTR_OK) fail("Renderer: %s", ntr_last_error());
}

void Renderer::updateResult(Buffer& pixels, Buffer& triMaterialColor, Buffer& triShadedColor)
{
    if (!m_batchRays) fail("Renderer::updateResult: no batch");
    const int perPrimary = (m_params.rayType == RayType_Primary) ? 1 : m_params.numSamples;
    // (the batch's first input slot counts from the start of this rank's range)
    const int numInputs = (m_batchRays == &m_primaryRays) ? (m_shardHi - m_shardLo) : m_batchRays->getSize() / perPrimary;
    int rc = ntr_reconstruct((int)m_params.rayType, perPrimary, m_shardLo + m_batchStart / perPrimary, numInputs,
                             (const int32_t*)m_primaryRays.getSlotToIDBuffer().getCudaPtr(),
                             (const NtrRayResult*)m_primaryRays.getResultBuffer().getCudaPtr(),
                             (const int32_t*)m_batchRays->getIDToSlotBuffer().getCudaPtr(),
                             (const NtrRayResult*)m_batchRays->getResultBuffer().getCudaPtr(),
                             (const uint32_t*)triMaterialColor.getCudaPtr(), (const uint32_t*)triShadedColor.getCudaPtr(),
                             (uint32_t*)pixels.getMutableCudaPtr(), NULL);
    if (rc != NTR_OK) fail("Renderer::updateResult: %s", ntr_last_error());
    if (ntr_stream_synchronize(NULL) != NTR_OK) fail("Renderer::updateResult: %s", ntr_last_error());
}

F32 Renderer::traceBatch(void)  // Renderer.cpp:568-579
{
    if (!m_batchRays) fail("Renderer::traceBatch: no batch");
    if (m_batchRays == &m_primaryRays)   // the primary batch: this rank's range of it
        return static_cast<CudaBVHTracer*>(m_cudaTracer)->traceRange(m_primaryRays, m_shardLo, m_shardHi - m_shardLo);
    CudaBVHTracer* tracer = static_cast<CudaBVHTracer*>(m_cudaTracer);
    tracer->setSchedHint((m_params.rayType == RayType_AO && m_predictSecondary && !m_params.sortSecondary) ? m_secondaryHint : NULL);
    const F32 sec = m_cudaTracer->traceBatch(*m_batchRays);
    tracer->setSchedHint(NULL);
    return sec;
}

int Renderer::getTotalNumRays(void)  // Renderer.cpp:676-710
{
    if (m_params.rayType == RayType_Primary) return m_shardHi - m_shardLo;
    int32_t hits = 0;
    if (ntr_count_hits((const NtrRayResult*)m_primaryRays.getResultBuffer().getCudaPtr() + m_shardLo, m_shardHi - m_shardLo, &hits, NULL) != NTR_OK)
        fail("Renderer: %s", ntr_last_error());
    return hits * m_params.numSamples;
}

}  // namespace FW
