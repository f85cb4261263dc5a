// ASAN / UBSAN driver: host SAH build (threaded paths) + host CPU trace + serialize round trip on a random soup
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ntrace_amd.h"
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 300000;
    std::vector<int> tri(3 * (size_t)n);
    std::vector<float> pos(9 * (size_t)n);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (int t = 0; t < n; t++) {
        const float cx = rnd() * 100, cy = rnd() * 100, cz = rnd() * 100;
        for (int v = 0; v < 3; v++) {
            tri[3 * t + v] = 3 * t + v;
            pos[9 * t + 3 * v + 0] = cx + rnd(); pos[9 * t + 3 * v + 1] = cy + rnd(); pos[9 * t + 3 * v + 2] = cz + rnd();
        }
    }
    NtrHostBvh* h = nullptr;
    if (ntr_sah_build(n, tri.data(), 3 * n, pos.data(), 1, 1, &h) != NTR_OK) { printf("build failed: %s\n", ntr_last_error()); return 1; }
    NtrHostBvhInfo info;
    ntr_host_bvh_info(h, &info);
    printf("nodes %lld bytes, inner %d leaves %d depth %d\n", (long long)info.nodesBytes, info.numInnerNodes, info.numLeafNodes, info.maxDepth);
    const int R = 20000;
    std::vector<NtrRay> rays(R);
    std::vector<NtrRayResult> res(R);
    for (int i = 0; i < R; i++) {
        rays[i].ox = rnd() * 100; rays[i].oy = rnd() * 100; rays[i].oz = -5.0f; rays[i].tmin = 0.0f;
        rays[i].dx = rnd() - 0.5f; rays[i].dy = rnd() - 0.5f; rays[i].dz = 1.0f; rays[i].tmax = 1e30f;
    }
    NtrTraceStats st;
    if (ntr_host_bvh_trace(h, R, 0, rays.data(), res.data(), nullptr, 0, &st) != NTR_OK) { printf("trace failed: %s\n", ntr_last_error()); return 1; }
    int hits = 0;
    for (int i = 0; i < R; i++) hits += res[i].id >= 0;
    printf("hits %d of %d, inner visits %lld\n", hits, R, (long long)st.numInnerVisits);
    ntr_host_bvh_free(h);
    return 0;
}
