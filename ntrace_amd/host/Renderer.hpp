// Renderer.hpp -- frame driver with the reference's batching semantics
// (src/rt/cuda/Renderer.hpp:78-115; Renderer.cpp:44-94, 147-305, 405-497, 501-579, 676-710).
// Kept: setMesh / setScene / setParams / getCudaBVH / beginFrame / nextBatch / traceBatch / updateResult (countHits + reconstruct:
// SURVEY 8(f-2)) / getTotalNumRays for the primary, AO and diffuse ray types; setShard for the multi-GPU extension.
// Out of scope: GL display, visualisation, VPL, kd-tree.
#pragma once
#include "MeshWavefrontIO.hpp"
#include "CudaBVHTracer.hpp"
#include "HLBVHBuilder.hpp"
#include "RayGen.hpp"

namespace FW {

class Renderer {
public:
    enum RayType { RayType_Primary = 0, RayType_AO, RayType_Diffuse, RayType_Max };

    struct Params {
        String  kernelName;
        RayType rayType;
        F32     aoRadius;
        S32     numSamples;
        bool    sortSecondary;
        Params(void) : kernelName(""), rayType(RayType_Primary), aoRadius(1.0f), numSamples(32), sortSecondary(false) {}
    };

    // builder: "SAHBVH" (host, leaf preferences (1,1)) or "HLBVH" (device LBVH) -- Renderer.builder in config.conf
    explicit Renderer(const String& builder = "SAHBVH");
    ~Renderer(void);

    void   setScene(Scene* scene);  // a Scene the caller built and owns
    // Renderer::setMesh (src/rt/cuda/Renderer.cpp:98-132): the same mesh -> nothing; otherwise the Renderer's own Scene of the previous
    // mesh goes, the BVH is invalidated, and a Scene is made from the mesh (Scene(const MeshBase&), Scene.cpp:101-136: here the OBJ
    // importer's WavefrontMesh, whose triangle ids are the reference's).  NULL releases the mesh.
    void   setMesh(const WavefrontMesh* mesh);
    Scene* getScene(void) const { return m_scene; }
    void   setBuildParams(const BVH::BuildParams& params) { invalidateBVH(); m_buildParams = params; }
    BVH::BuildParams& getBuildParams(void) { return m_buildParams; }
    void   invalidateBVH(void) { delete m_accelStruct; m_accelStruct = NULL; m_leafDepthOf = NULL; }
    void   setParams(const Params& params);
    void   setEnableRandom(bool enable) { m_enableRandom = enable; }
    CudaVirtualTracer& getCudaTracer(void) { return *m_cudaTracer; }
    CudaAS* getCudaBVH(void);
    // Multi-GPU (no counterpart in the reference, which is single-device).  setShard: this Renderer traces the rank-th of
    // `world` screen-tile ranges of every frame -- contiguous 64-aligned ranges of the PixelTable index space (ntr_frame_shard) -- and
    // the AO / diffuse rays of its own primary hits; (0, 1) = the whole frame.  adoptCudaBVH: use a BVH built elsewhere (the root's,
    // replicated by DistGroup::broadcastBVH) instead of building one; the Renderer owns it from then on.
    void   setShard(int rank, int world);
    S32    getShardLo(void) const { return m_shardLo; }
    S32    getShardHi(void) const { return m_shardHi; }
    void   adoptCudaBVH(CudaAS* as) { delete m_accelStruct; m_accelStruct = as; m_leafDepthOf = NULL; }
    RayGen& getRayGen(void) { return m_raygen; }
    // Dispatch hint of the AO batches (no counterpart in the reference): beside generating a batch the Renderer predicts the cost of its
    // 256-ray blocks from the depth of the leaves its pixels' primary rays hit (ntr_bvh_leaf_depths, ntr_secondary_block_costs) and hands
    // the tracer a hint that starts from that order (ntr_sched_hint_predict).  On by default; hit records do not depend on it.
    void   setPredictSecondaryOrder(bool enable) { m_predictSecondary = enable; }
    // BVH cache files, "<cachePath>/<hash>_<builder>.dat" (Renderer.cpp:173-191, 293-299; format of CudaBVH::serialize).  Off by
    // default, like Renderer.cacheDataStructure in the reference's environment.
    void   setCachePath(const String& path) { m_cachePath = path; }
    void   setCacheDataStructure(bool enable) { m_cacheDataStructure = enable; }
    String getCacheFileName(void);

    void beginFrame(const CameraView& camera);
    bool nextBatch(void);
    F32  traceBatch(void);       // launch time in seconds
    int  getTotalNumRays(void);  // for the selected ray type, excluding degenerates
    // Renderer::updateResult (Renderer.cpp:583-659): current batch -> ABGR8 pixels (width*height, by pixel id).
    // Per-triangle colours replace Scene::getTriMaterialColorBuffer / getTriShadedColorBuffer.
    void updateResult(Buffer& pixels, Buffer& triMaterialColor, Buffer& triShadedColor);
    RayBuffer& getPrimaryRays(void) { return m_primaryRays; }
    RayBuffer* getBatchRays(void) { return m_batchRays; }

private:
    void predictSecondaryOrder(void);
    Renderer(const Renderer&);
    Renderer& operator=(const Renderer&);

    String             m_builder;
    Platform           m_platform;
    BVH::BuildParams   m_buildParams;
    RayGen             m_raygen;
    Params             m_params;
    bool               m_enableRandom;
    Scene*             m_scene;
    const WavefrontMesh* m_mesh;            // setMesh: the mesh m_scene was made from (the Renderer then owns m_scene)
    bool               m_ownsScene;
    F32                m_cameraFar;
    RayBuffer          m_primaryRays;
    RayBuffer          m_secondaryRays;
    bool               m_newBatch;
    RayBuffer*         m_batchRays;
    S32                m_batchStart;
    CudaAS*            m_accelStruct;
    String             m_cachePath;
    bool               m_cacheDataStructure;
    CudaVirtualTracer* m_cudaTracer;
    bool               m_predictSecondary;
    Buffer             m_leafDepth;            // S32 per triangle: depth of its leaf in m_leafDepthOf
    CudaAS*            m_leafDepthOf;
    Buffer             m_blockCost;            // U32 per 256-ray block of the current secondary batch
    NtrSchedHint*      m_secondaryHint;
    int                m_shardRank, m_shardWorld;
    S32                m_shardLo, m_shardHi;   // this rank's range of the current frame's primary slots
};

}  // namespace FW
