// host_test.cpp -- exercises the C++ host mirror (ntrace_amd/host) the way NTrace's own code uses
// the reference classes.  `host_test cpu` needs no GPU; `host_test gpu` runs the device paths.
// Compiled with plain g++ against libntrace_amd.so: the mirror's headers contain no HIP types.
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

#include "DistGroup.hpp"
#include <thread>
#include <string>
#include "HLBVHBuilder.hpp"
#include "Random.hpp"
#include "Renderer.hpp"

using namespace FW;

static int g_failed = 0;
#define CHECK(X) do { if (!(X)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #X); g_failed++; } } while (0)

// A room of 6 tessellated walls with a few boxes: nTess^2 * 2 triangles per face.
static void makeScene(std::vector<Vec3i>& tris, std::vector<Vec3f>& verts, int nTess)
{
    auto quad = [&](Vec3f p0, Vec3f du, Vec3f dv) {
        int base = (int)verts.size();
        for (int i = 0; i <= nTess; i++)
            for (int j = 0; j <= nTess; j++) verts.push_back(p0 + du * ((F32)i / nTess) + dv * ((F32)j / nTess));
        for (int i = 0; i < nTess; i++)
            for (int j = 0; j < nTess; j++) {
                int a = base + i * (nTess + 1) + j, b = a + nTess + 1;
                tris.push_back(Vec3i(a, b, b + 1));
                tris.push_back(Vec3i(a, b + 1, a + 1));
            }
    };
    auto box = [&](Vec3f lo, Vec3f hi) {
        Vec3f d = hi - lo;
        quad(lo, Vec3f(d.x, 0, 0), Vec3f(0, d.y, 0));
        quad(Vec3f(lo.x, lo.y, hi.z), Vec3f(d.x, 0, 0), Vec3f(0, d.y, 0));
        quad(lo, Vec3f(d.x, 0, 0), Vec3f(0, 0, d.z));
        quad(Vec3f(lo.x, hi.y, lo.z), Vec3f(d.x, 0, 0), Vec3f(0, 0, d.z));
        quad(lo, Vec3f(0, d.y, 0), Vec3f(0, 0, d.z));
        quad(Vec3f(hi.x, lo.y, lo.z), Vec3f(0, d.y, 0), Vec3f(0, 0, d.z));
    };
    box(Vec3f(-10.5f, -10.25f, -10.75f), Vec3f(10.25f, 10.5f, 10.125f));
    box(Vec3f(-3.5f, -10.25f, 1.5f), Vec3f(0.5f, -4.0f, 5.25f));
    box(Vec3f(2.25f, -10.25f, -2.0f), Vec3f(5.0f, -1.5f, 1.75f));
}

static CameraView makeCamera(int w, int h)
{
    CameraView c;
    c.position = Vec3f(0.3f, 0.7f, -9.0f);
    const float th = std::tan(0.5f), aspect = (float)w / h;
    // pinhole looking down +z: world = position + (0,0,1) + nx*th*aspect*(1,0,0) - ny*th*(0,1,0)
    const float m[16] = {th * aspect, 0, 0, c.position.x, 0, -th, 0, c.position.y, 0, 0, 0, c.position.z + 1.0f, 0, 0, 0, 1};
    std::memcpy(c.nscreenToWorld.m, m, sizeof(m));
    c.cameraFar = 100.0f;
    c.width = w;
    c.height = h;
    return c;
}

static void cpuTests()
{
    // Buffer: CPU-only use never touches the device
    Buffer b;
    b.resize(64);
    std::memset(b.getMutablePtr(), 7, 64);
    b.resize(128);  // keeps contents
    CHECK(b.getSize() == 128 && b.getPtr()[63] == 7);
    b.resizeDiscard(16);
    CHECK(b.getSize() == 16 && b.getOwner() == Buffer::Module_None);
    int ext[4] = {1, 2, 3, 4};
    Buffer wrap;
    wrap.wrapCPU(ext, sizeof(ext));
    ((int*)wrap.getMutablePtr())[2] = 9;
    CHECK(ext[2] == 9);

    // RayBuffer: resize never shrinks allocations; id <-> slot maps (RayBuffer.cpp:38-65)
    RayBuffer rb(8);
    Ray r;
    r.origin = Vec3f(1, 2, 3);
    r.tmax = 5.0f;
    rb.setRay(3, r, 5);
    CHECK(rb.getSlotForID(5) == 3 && rb.getIDForSlot(3) == 5 && rb.getRayForID(5).tmax == 5.0f);
    rb.resize(4);
    CHECK(rb.getSize() == 4 && rb.getRayBuffer().getSize() == 8 * (S64)sizeof(Ray));
    Ray d;
    d.tmin = 2.0f;
    d.degenerate();
    CHECK(d.tmax == 1.0f);
    RayResult rr;
    CHECK(!rr.hit());

    // RANROT-A: deterministic, seed 0 == seed 0xFFFFFFFF (Random.cpp:61-62)
    CHECK(Random(0).getU32() == Random(0xFFFFFFFFu).getU32());
    CHECK(Random(1).getU32() != Random(2).getU32());
    Random a(42), a2(42);
    bool same = true;
    for (int i = 0; i < 100; i++) same = same && (a.getU32() == a2.getU32());
    CHECK(same);

    // host SAH build -> Compact, serialisation round trip in the bvhcache format
    std::vector<Vec3i> tris;
    std::vector<Vec3f> verts;
    makeScene(tris, verts, 4);
    Scene scene((S32)tris.size(), tris.data(), (S32)verts.size(), verts.data());
    Platform platform("GPU");
    platform.setLeafPreferences(1, 1);
    BVH::Stats stats;
    BVH::BuildParams params;
    params.stats = &stats;
    BVH bvh(&scene, platform, params);
    CHECK(stats.numLeafNodes == (S32)tris.size() && stats.numInnerNodes == stats.numLeafNodes - 1 && stats.maxDepth <= 64);
    CudaBVH cbvh(bvh, BVHLayout_Compact);
    CHECK(cbvh.getNodeBuffer().getSize() == (S64)stats.numInnerNodes * 64);
    CHECK(cbvh.getTriWoopBuffer().getSize() == (S64)(tris.size() * 3 + stats.numLeafNodes) * 16);
    CHECK(cbvh.getTriIndexBuffer().getSize() * 4 == cbvh.getTriWoopBuffer().getSize());
    std::stringstream ss;
    cbvh.serialize(ss);
    CudaBVH back(ss);
    CHECK(!hasError() && back.getLayout() == BVHLayout_Compact);
    CHECK(back.getNodeBuffer().getSize() == cbvh.getNodeBuffer().getSize() &&
          std::memcmp(back.getNodeBuffer().getPtr(), cbvh.getNodeBuffer().getPtr(), (size_t)cbvh.getNodeBuffer().getSize()) == 0);
    CHECK(std::memcmp(back.getTriWoopBuffer().getPtr(), cbvh.getTriWoopBuffer().getPtr(), (size_t)cbvh.getTriWoopBuffer().getSize()) == 0);
    bool threw = false;
    try { BVH::BuildParams p2; p2.builder = "Nope"; BVH bad(&scene, platform, p2); } catch (const FatalError&) { threw = true; }
    CHECK(threw);

    // CudaAS::trace, the host tracer (CudaBVH.cpp:213-302): rays down the z axis of the closed room hit the far wall
    // (z = 10.125) or a box in front of it, at exactly representable distances; visibility marks the triangles hit
    {
        CudaAS& as = cbvh;
        RayBuffer rb(4, true);
        Ray q;
        q.direction = Vec3f(0.0f, 0.0f, 1.0f);
        q.tmin = 0.0f;
        q.tmax = 100.0f;
        q.origin = Vec3f(-8.0f, 8.0f, -9.0f);  rb.setRay(0, q);   // free path to the far wall: t = 19.125
        q.origin = Vec3f(-1.5f, -8.0f, -9.0f); rb.setRay(1, q);   // box [-3.5,0.5]x[-10.25,-4]x[1.5,5.25]: t = 10.5
        q.origin = Vec3f(3.0f, -5.0f, -9.0f);  rb.setRay(2, q);   // box [2.25,5]x[-10.25,-1.5]x[-2,1.75]: t = 7
        q.tmax = 5.0f;                         rb.setRay(3, q);   // same ray cut short: miss, t = tmax
        Buffer vis;
        vis.resize((S64)tris.size() * 4);
        vis.clear(0);
        as.trace(rb, vis);
        CHECK(rb.getResultForSlot(0).hit() && rb.getResultForSlot(0).t == 19.125f);
        CHECK(rb.getResultForSlot(1).hit() && rb.getResultForSlot(1).t == 10.5f);
        CHECK(rb.getResultForSlot(2).hit() && rb.getResultForSlot(2).t == 7.0f);
        CHECK(!rb.getResultForSlot(3).hit() && rb.getResultForSlot(3).t == 5.0f);
        int marked = 0;
        for (size_t i = 0; i < tris.size(); i++) marked += ((const S32*)vis.getPtr())[i];
        CHECK(marked >= 1 && marked <= 3 && ((const S32*)vis.getPtr())[rb.getResultForSlot(0).id] == 1);
        rb.setNeedClosestHit(false);  // any hit: same rays still hit / miss
        Buffer none;
        as.trace(rb, none);
        CHECK(rb.getResultForSlot(0).hit() && rb.getResultForSlot(2).hit() && !rb.getResultForSlot(3).hit());
    }

    // tracer front end: argument checks that precede device work (CudaBVHTracer.cpp:92-100)
    CudaBVHTracer tracer;
    tracer.setKernel("kepler_dynamic_fetch");
    CHECK(tracer.getDesiredBVHLayout() == BVHLayout_Compact && tracer.getKernelConfig().usePersistentThreads == 1);
    RayBuffer empty(0);
    CHECK(tracer.traceBatch(empty) == 0.0f);
    threw = false;
    try { RayBuffer one(1); tracer.traceBatch(one); } catch (const FatalError& e) { threw = e.message.find("No BVH") != std::string::npos; }
    CHECK(threw);
    threw = false;
    try { tracer.setKernel("no_such_kernel"); } catch (const FatalError&) { threw = true; }
    CHECK(threw);
    // BVH cache files (Renderer.cpp:173-191, 293-299): "<path>/<hash>_<builder>.dat"; the second Renderer imports what the first wrote
    {
        CHECK(hashBits(1u) != hashBits(2u) && hashBuffer("abc", 3) != hashBuffer("abd", 3));
        CHECK(hashBuffer("0123456789abcdef", 16) != hashBuffer("0123456789abcdeg", 16));
        std::vector<Vec3i> tris;
        std::vector<Vec3f> verts;
        makeScene(tris, verts, 6);
        Renderer::Params params;
        params.kernelName = "fermi_speculative_while_while";
        Scene scene((S32)tris.size(), tris.data(), (S32)verts.size(), verts.data());
        char dir[] = "/tmp/ntr_bvhcache_XXXXXX";
        CHECK(mkdtemp(dir) != NULL);
        Renderer first("SAHBVH");
        first.setScene(&scene);
        first.setParams(params);
        first.setCachePath(dir);
        first.setCacheDataStructure(true);
        const String name = first.getCacheFileName();
        CHECK(name.find(String(dir) + "/") == 0 && name.size() == std::strlen(dir) + 1 + 8 + 1 + 6 + 4 && name.find("_SAHBVH.dat") != String::npos);
        CudaAS* built = first.getCudaBVH();
        std::ifstream probe(name.c_str(), std::ios::binary);
        CHECK(built && probe.good());
        // poison the builder name: a second Renderer can only get a BVH out of the cache file
        Renderer second("SAHBVH");
        second.setScene(&scene);
        second.setParams(params);
        second.setCachePath(dir);
        second.setCacheDataStructure(true);
        CHECK(second.getCacheFileName() == name);
        second.getBuildParams().builder = "no_such_builder";
        CudaAS* cached = second.getCudaBVH();
        CHECK(cached && cached->getNodeBuffer().getSize() == built->getNodeBuffer().getSize() &&
              std::memcmp(cached->getNodeBuffer().getPtr(), built->getNodeBuffer().getPtr(), (size_t)built->getNodeBuffer().getSize()) == 0);
        CHECK(cached && cached->getTriWoopBuffer().getSize() == built->getTriWoopBuffer().getSize() &&
              std::memcmp(cached->getTriIndexBuffer().getPtr(), built->getTriIndexBuffer().getPtr(), (size_t)built->getTriIndexBuffer().getSize()) == 0);
        // a different scene or build parameter names a different file
        Renderer third("SAHBVH");
        third.setScene(&scene);
        third.setParams(params);
        third.setCachePath(dir);
        third.getBuildParams().splitAlpha = 0.5f;
        CHECK(third.getCacheFileName() != name);
        verts[0].x += 1.0f;
        Scene moved((S32)tris.size(), tris.data(), (S32)verts.size(), verts.data());
        third.getBuildParams().splitAlpha = first.getBuildParams().splitAlpha;
        third.setScene(&moved);
        CHECK(third.getCacheFileName() != name);
        std::remove(name.c_str());
        rmdir(dir);
    }
    // Renderer::setMesh (Renderer.cpp:98-132): the Renderer makes -- and owns -- the Scene of a mesh; the same mesh again changes nothing;
    // another mesh or NULL releases it; triangle ids are the importer's (submeshes concatenated, polygons fan-triangulated)
    {
        const char* obj = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nf 1 2 3 4\nf 1 2 5\n";
        WavefrontMesh mesh, other;
        CHECK(parseWavefrontMesh(mesh, obj, std::vector<String>()) && mesh.numTriangles() == 3);
        CHECK(parseWavefrontMesh(other, "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", std::vector<String>()) && other.numTriangles() == 1);
        Renderer r("SAHBVH");
        CHECK(r.getScene() == NULL);
        r.setMesh(&mesh);
        Scene* s1 = r.getScene();
        CHECK(s1 != NULL && s1->getNumTriangles() == 3 && s1->getNumVertices() == 5);
        r.setMesh(&mesh);
        CHECK(r.getScene() == s1);
        r.setMesh(&other);
        CHECK(r.getScene() != NULL && r.getScene()->getNumTriangles() == 1);
        r.setMesh(NULL);
        CHECK(r.getScene() == NULL);
    }
    // multi-GPU partition arithmetic (ntr_frame_shard / ntr_frame_ao_batches): contiguous, 64-aligned ranges that cover the frame; AO
    // batches of a range as RayGen::batching cuts them
    for (int n : {0, 1, 63, 64, 65, 1000, 1920 * 1080}) {
        for (int world : {1, 2, 3, 8}) {
            int32_t prev = 0;
            for (int r = 0; r < world; r++) {
                int32_t lo = -1, hi = -1;
                CHECK(ntr_frame_shard(n, r, world, 64, &lo, &hi) == NTR_OK && lo == prev && hi >= lo && (lo % 64 == 0 || lo == n));
                prev = hi;
                int32_t first[64], count[64], nb = 0;
                CHECK(ntr_frame_ao_batches(lo, hi, 8, 1 << 20, first, count, 64, &nb) == NTR_OK);
                int32_t covered = 0;
                for (int b = 0; b < nb; b++) { CHECK(first[b] == lo + covered && count[b] * 8 <= (1 << 20) && count[b] > 0); covered += count[b]; }
                CHECK(covered == hi - lo);
            }
            CHECK(prev == n);
        }
    }
    {
        int32_t lo, hi;
        CHECK(ntr_frame_shard(1920 * 1080, 3, 8, 64, &lo, &hi) == NTR_OK && hi - lo == 259200 && lo % 64 == 0);
        CHECK(ntr_frame_shard(100, 2, 2, 64, &lo, &hi) != NTR_OK);
    }
    // sticky error model (Defs.hpp:142-151)
    clearError();
    setError("first %d", 1);
    setError("second");
    CHECK(hasError() && getError() == "first 1");
    clearError();
    CHECK(!hasError());
}

static void gpuTests()
{
    std::vector<Vec3i> tris;
    std::vector<Vec3f> verts;
    makeScene(tris, verts, 24);
    Scene scene((S32)tris.size(), tris.data(), (S32)verts.size(), verts.data());
    const int W = 320, H = 200;
    CameraView cam = makeCamera(W, H);

    std::vector<RayResult> prim[2];
    int b = 0;
    for (const char* builder : {"SAHBVH", "HLBVH"}) {
        Renderer renderer(builder);
        renderer.setScene(&scene);
        for (const char* kernel : {"fermi_speculative_while_while", "kepler_dynamic_fetch"}) {
            Renderer::Params p;
            p.kernelName = kernel;
            p.rayType = Renderer::RayType_Primary;
            renderer.setParams(p);
            renderer.beginFrame(cam);
            CHECK(renderer.getTotalNumRays() == W * H);
            int batches = 0;
            F32 sec = 0.0f;
            while (renderer.nextBatch()) { sec += renderer.traceBatch(); batches++; }
            CHECK(batches == 1 && sec > 0.0f);
            RayBuffer& rays = renderer.getPrimaryRays();
            int hits = 0;
            prim[b].resize(W * H);
            for (int i = 0; i < W * H; i++) {
                prim[b][i] = rays.getResultForSlot(i);  // D->H migration through Buffer
                hits += prim[b][i].hit();
            }
            CHECK(hits == W * H);  // closed room: every primary ray hits
            CHECK(rays.getIDForSlot(rays.getSlotForID(1234)) == 1234);

            // AO: 8 samples, <= 2^20 rays per batch (Renderer.cpp:45), any-hit
            p.rayType = Renderer::RayType_AO;
            p.numSamples = 8;
            p.aoRadius = 2.0f;
            renderer.setParams(p);
            renderer.beginFrame(cam);
            CHECK(renderer.getTotalNumRays() == W * H * 8);
            S64 traced = 0;
            batches = 0;
            while (renderer.nextBatch()) {
                CHECK(!renderer.getBatchRays()->getNeedClosestHit() && renderer.getBatchRays()->getSize() <= (1 << 20));
                renderer.traceBatch();
                traced += renderer.getBatchRays()->getSize();
                batches++;
            }
            CHECK(traced == (S64)W * H * 8 && batches == 1);
            // the AO batch was dispatched by the Renderer's leaf-depth hint; without it the records must be the same
            {
                std::vector<U8> withHint, without;
                for (int pass = 0; pass < 2; pass++) {
                    renderer.setPredictSecondaryOrder(pass == 0);
                    renderer.beginFrame(cam);
                    std::vector<U8>& dst = pass == 0 ? withHint : without;
                    while (renderer.nextBatch()) {
                        renderer.traceBatch();
                        Buffer& rb = renderer.getBatchRays()->getResultBuffer();
                        const size_t at = dst.size();
                        dst.resize(at + (size_t)rb.getSize());
                        std::memcpy(dst.data() + at, rb.getPtr(), (size_t)rb.getSize());
                    }
                }
                renderer.setPredictSecondaryOrder(true);
                CHECK(withHint.size() == (size_t)W * H * 8 * 16 && withHint == without);
            }
            // sorted AO batches + image reconstruction: AO image = fraction of unoccluded samples
            {
                p.sortSecondary = true;
                renderer.setParams(p);
                renderer.beginFrame(cam);
                Buffer pixels, matCol, shCol;
                pixels.resizeDiscard((S64)W * H * 4);
                pixels.clear(0);
                std::vector<U32> cols(tris.size(), 0xFF8090A0u);
                matCol.set(cols.data(), (S64)cols.size() * 4);
                shCol.set(cols.data(), (S64)cols.size() * 4);
                while (renderer.nextBatch()) { renderer.traceBatch(); renderer.updateResult(pixels, matCol, shCol); }
                const U32* px = (const U32*)pixels.getPtr();
                int grey = 0;
                for (int i = 0; i < W * H; i++) {
                    const U32 r = px[i] & 0xFF, g = (px[i] >> 8) & 0xFF, b2 = (px[i] >> 16) & 0xFF;
                    grey += (r == g && g == b2 && (px[i] >> 24) == 0xFF);  // closed room: every pixel is an AO grey
                }
                CHECK(grey == W * H);
                p.sortSecondary = false;
            }
            // diffuse: same generator, closest hit, camera-far length (Renderer.cpp:533-537)
            p.rayType = Renderer::RayType_Diffuse;
            renderer.setParams(p);
            renderer.beginFrame(cam);
            while (renderer.nextBatch()) {
                CHECK(renderer.getBatchRays()->getNeedClosestHit());
                renderer.traceBatch();
            }
        }
        b++;
    }
    // the two builders give the same visible triangle almost everywhere (Woop rows differ in the last bits)
    int sameId = 0;
    double maxRel = 0;
    for (int i = 0; i < W * H; i++) {
        sameId += prim[0][i].id == prim[1][i].id;
        maxRel = std::fmax(maxRel, std::fabs(prim[0][i].t - prim[1][i].t) / prim[0][i].t);
    }
    CHECK(sameId > W * H * 0.995 && maxRel < 1e-4);

    // multi-GPU path: Renderer::setShard + DistGroup.  This box has one GPU: the RCCL group has world size 1 (every call
    // goes through RCCL: id, communicator, broadcast, the grouped gather), and the 3-way sharding is exercised by three Renderers that
    // each trace their own range of the same frame on this GPU, compared with the unsharded frame.
    {
        char id[DistGroup::IdBytes];
        DistGroup::uniqueId(id);
        DistGroup group(id, 0, 1);
        CHECK(group.getRank() == 0 && group.getWorld() == 1);
        Renderer whole("SAHBVH");
        whole.setScene(&scene);
        Renderer::Params p;
        p.kernelName = "fermi_speculative_while_while";
        p.rayType = Renderer::RayType_AO;
        p.numSamples = 8;
        p.aoRadius = 2.0f;
        whole.setParams(p);
        whole.beginFrame(cam);
        const int totalAO = whole.getTotalNumRays();
        Buffer pixels, matCol, shCol, fullPixels, fullRecords;
        pixels.resizeDiscard((S64)W * H * 4);
        pixels.clear(0);
        std::vector<U32> cols(tris.size(), 0xFF8090A0u);
        matCol.set(cols.data(), (S64)cols.size() * 4);
        shCol.set(cols.data(), (S64)cols.size() * 4);
        while (whole.nextBatch()) { whole.traceBatch(); whole.updateResult(pixels, matCol, shCol); }
        // BVH replication: a second BVH object receives the root's buffers through RCCL (root == the only rank: the buffers come back as sent)
        CudaBVH* rootBvh = dynamic_cast<CudaBVH*>(whole.getCudaBVH());
        CHECK(rootBvh != NULL);
        const S64 nodesBefore = rootBvh->getNodeBuffer().getSize();
        std::vector<U8> nodesCopy((size_t)nodesBefore);
        std::memcpy(nodesCopy.data(), rootBvh->getNodeBuffer().getPtr(), (size_t)nodesBefore);
        group.broadcastBVH(*rootBvh, 0);
        CHECK(rootBvh->getNodeBuffer().getSize() == nodesBefore && std::memcmp(rootBvh->getNodeBuffer().getPtr(), nodesCopy.data(), (size_t)nodesBefore) == 0);
        // the frame's collective at world size 1: records and pixels arrive as traced
        group.gatherRecords(whole, fullRecords, 0);
        CHECK(fullRecords.getSize() == (S64)W * H * 16 &&
              std::memcmp(fullRecords.getPtr(), whole.getPrimaryRays().getResultBuffer().getPtr(), (size_t)W * H * 16) == 0);
        group.gatherPixels(whole, pixels, fullPixels, 0);
        CHECK(fullPixels.getSize() == (S64)W * H * 4 && std::memcmp(fullPixels.getPtr(), pixels.getPtr(), (size_t)W * H * 4) == 0);
        // the whole frame's primary-type image (AO rays depend on where a batch starts -- the rotation hash takes the batch-local index,
        // RayGenKernels.cu:179 -- so only the primary image is partition-invariant)
        Buffer primPixels;
        primPixels.resizeDiscard((S64)W * H * 4);
        primPixels.clear(0);
        {
            Renderer::Params pp = p;
            pp.rayType = Renderer::RayType_Primary;
            whole.setParams(pp);
            whole.beginFrame(cam);
            while (whole.nextBatch()) { whole.traceBatch(); whole.updateResult(primPixels, matCol, shCol); }
            whole.setParams(p);
        }
        // three ranks' worth of sharded Renderers on this one GPU: ranges partition the frame, records and pixels equal the whole frame's
        int sumAO = 0, covered = 0;
        std::vector<U32> assembled((size_t)W * H, 0u);
        std::vector<U32> tilesAO((size_t)W * H, 0u);
        for (int r = 0; r < 3; r++) {
            Renderer part("SAHBVH");
            part.setScene(&scene);
            part.setShard(r, 3);
            part.setParams(p);
            part.beginFrame(cam);
            CHECK(part.getShardLo() == covered && part.getShardLo() % 64 == 0);
            covered = part.getShardHi();
            sumAO += part.getTotalNumRays();
            Buffer px;
            px.resizeDiscard((S64)W * H * 4);
            px.clear(0);
            S64 traced = 0;
            while (part.nextBatch()) { part.traceBatch(); part.updateResult(px, matCol, shCol); traced += part.getBatchRays()->getSize(); }
            CHECK(traced == (S64)(part.getShardHi() - part.getShardLo()) * 8);
            const RayResult* own = (const RayResult*)part.getPrimaryRays().getResultBuffer().getPtr();
            const RayResult* ref = (const RayResult*)whole.getPrimaryRays().getResultBuffer().getPtr();
            int same = 0;
            for (int i = part.getShardLo(); i < part.getShardHi(); i++) same += (own[i].id == ref[i].id && own[i].t == ref[i].t);
            CHECK(same == part.getShardHi() - part.getShardLo());
            const U32* ppx = (const U32*)px.getPtr();
            int overlap = 0;
            for (int i = 0; i < W * H; i++) if (ppx[i]) { overlap += tilesAO[(size_t)i] != 0u; tilesAO[(size_t)i] = ppx[i]; }
            CHECK(overlap == 0);
            // the primary ray type on a shard: one batch of the range's size
            Renderer::Params pp = p;
            pp.rayType = Renderer::RayType_Primary;
            part.setParams(pp);
            part.beginFrame(cam);
            CHECK(part.getTotalNumRays() == part.getShardHi() - part.getShardLo());
            int nb = 0;
            Buffer px2;
            px2.resizeDiscard((S64)W * H * 4);
            px2.clear(0);
            while (part.nextBatch()) { CHECK(part.traceBatch() > 0.0f); part.updateResult(px2, matCol, shCol); nb++; }
            CHECK(nb == 1);
            const U32* p2 = (const U32*)px2.getPtr();
            for (int i = 0; i < W * H; i++) if (p2[i]) assembled[(size_t)i] = p2[i];
        }
        CHECK(covered == W * H && sumAO == totalAO);
        int written = 0;
        for (int i = 0; i < W * H; i++) written += tilesAO[(size_t)i] != 0u;
        CHECK(written == W * H);   // the ranks' AO tiles cover the image exactly once
        CHECK(std::memcmp(assembled.data(), primPixels.getPtr(), (size_t)W * H * 4) == 0);
    }

    // Thread-per-GPU mode on ONE device: N host threads, each with its own sharded Renderer tracing its range of the
    // same frame CONCURRENTLY through the one library (per-thread error state, per-device scheduling state under its own mutex), the
    // gather stubbed by a device copy of every rank's slice of hit records into the assembled frame -- which must equal the unsharded
    // frame bit for bit.  (The RCCL transport needs one GPU per rank and stays unexecuted at N > 1 on this box.)
    {
        const int N = 3;
        Renderer whole("SAHBVH");
        whole.setScene(&scene);
        Renderer::Params p;
        p.kernelName = "fermi_speculative_while_while";
        p.rayType = Renderer::RayType_AO;
        p.numSamples = 8;
        p.aoRadius = 2.0f;
        whole.setParams(p);
        whole.beginFrame(cam);
        const int totalAO = whole.getTotalNumRays();
        while (whole.nextBatch()) whole.traceBatch();
        Buffer assembled;
        assembled.resizeDiscard((S64)W * H * 16);
        assembled.clear(0xEE);
        void* d_assembled = (void*)assembled.getMutableCudaPtr();
        std::vector<int> rangeLo(N), rangeHi(N), aoRays(N), ok(N, 0);
        std::vector<std::string> errors(N);
        std::vector<std::thread> threads;
        for (int r = 0; r < N; r++)
            threads.emplace_back([&, r]() {
                try {
                    Renderer part("SAHBVH");
                    part.setScene(&scene);
                    part.setShard(r, N);
                    part.setParams(p);
                    for (int frame = 0; frame < 3; frame++) {   // re-traced frames: the automatic scheduling feedback of every thread's batches is live
                        part.beginFrame(cam);
                        while (part.nextBatch()) part.traceBatch();
                    }
                    rangeLo[r] = part.getShardLo(); rangeHi[r] = part.getShardHi(); aoRays[r] = part.getTotalNumRays();
                    // the gather, stubbed: this rank's slice of records -> its place in the assembled frame (device to device)
                    const char* own = (const char*)part.getPrimaryRays().getResultBuffer().getCudaPtr() + (S64)rangeLo[r] * 16;
                    if (ntr_memcpy_d2d((char*)d_assembled + (S64)rangeLo[r] * 16, own, (size_t)(rangeHi[r] - rangeLo[r]) * 16, NULL) != NTR_OK ||
                        ntr_stream_synchronize(NULL) != NTR_OK) throw FatalError{ntr_last_error()};
                    ok[r] = 1;
                } catch (const FatalError& e) { errors[r] = e.message; }
            });
        for (auto& t : threads) t.join();
        int covered = 0, sumAO = 0;
        for (int r = 0; r < N; r++) {
            if (!ok[r]) std::printf("thread %d: %s\n", r, errors[r].c_str());
            CHECK(ok[r] == 1 && rangeLo[r] == covered && rangeLo[r] % 64 == 0);
            covered = rangeHi[r];
            sumAO += aoRays[r];
        }
        CHECK(covered == W * H && sumAO == totalAO);
        CHECK(std::memcmp(assembled.getPtr(), whole.getPrimaryRays().getResultBuffer().getPtr(), (size_t)W * H * 16) == 0);
    }

    // layout mismatch is fatal (CudaBVHTracer.cpp:99-100)
    {
        HLBVHParams hp;
        HLBVHBuilder lb(&scene, Platform("GPU"), hp);
        CHECK(lb.getGPUTime() > 0.0f && lb.getNodeBuffer().getSize() == lb.getBuildResult().nodesBytes);
        struct WrongLayout : CudaBVH { WrongLayout() : CudaBVH(BVHLayout_AOS_AOS) {} } wrong;
        CudaBVHTracer t;
        t.setKernel("fermi_speculative_while_while");
        t.setBVH(&wrong);
        RayBuffer one(1);
        bool threw = false;
        try { t.traceBatch(one); } catch (const FatalError& e) { threw = e.message.find("Incorrect BVH layout") != std::string::npos; }
        CHECK(threw);
    }
}

int main(int argc, char** argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    try {
        if (gpu) gpuTests(); else cpuTests();
    } catch (const FatalError& e) {
        std::printf("unexpected FW::fail: %s\n", e.message.c_str());
        return 2;
    }
    std::printf("host_test %s: %s\n", gpu ? "gpu" : "cpu", g_failed ? "FAILED" : "ok");
    return g_failed ? 1 : 0;
}
