#include "MeshWavefrontIO.hpp"

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

namespace FW {

int WavefrontMesh::numTriangles(void) const
{
    size_t n = 0;
    for (const std::vector<Vec3i>& s : submeshes) n += s.size();
    return (int)n;
}

void WavefrontMesh::flatten(std::vector<Vec3i>& tris) const
{
    tris.clear();
    for (const std::vector<Vec3i>& s : submeshes) tris.insert(tris.end(), s.begin(), s.end());
}

Scene* WavefrontMesh::createScene(void) const
{
    std::vector<Vec3i> tris;
    flatten(tris);
    return new Scene((S32)tris.size(), tris.data(), (S32)vertices.size(), vertices.data());
}

static void skipSpace(const char*& p) { while (*p == ' ' || *p == '\t') p++; }
static bool literal(const char*& p, const char* lit)
{
    const char* q = p;
    while (*lit) if (*q++ != *lit++) return false;
    p = q;
    return true;
}
static bool parseInt(const char*& p, S32& v)
{
    char* e;
    long x = std::strtol(p, &e, 10);
    if (e == p) return false;
    v = (S32)x;
    p = e;
    return true;
}
static bool parseFloats(const char*& p, F32* out, int n)
{
    for (int i = 0; i < n; i++) {
        skipSpace(p);
        char* e;
        float x = std::strtof(p, &e);
        if (e == p) return false;
        out[i] = x;
        p = e;
    }
    return true;
}

bool parseWavefrontMesh(WavefrontMesh& mesh, const String& objText, const std::vector<String>& knownMaterials)
{
    std::vector<Vec3f> positions;
    int numTex = 0, numNrm = 0;
    std::map<std::tuple<S32, S32, S32>, S32> vertexHash;
    std::map<String, int> materialSubmesh;
    for (const String& m : knownMaterials) materialSubmesh[m] = -1;
    std::vector<S32> vertexTmp;
    std::vector<Vec3i> indexTmp;
    int submesh = -1, defaultSubmesh = -1;

    auto addSubmesh = [&](const String& mat) {
        mesh.submeshes.push_back(std::vector<Vec3i>());
        mesh.submeshMaterial.push_back(mat);
        return (int)mesh.submeshes.size() - 1;
    };

    std::istringstream in(objText);
    String line;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        const char* ptr = line.c_str();
        skipSpace(ptr);
        if (literal(ptr, "v ")) {
            Vec3f v;
            if (parseFloats(ptr, &v.x, 3)) positions.push_back(v);
        } else if (literal(ptr, "vt ")) {
            numTex++;
        } else if (literal(ptr, "vn ")) {
            numNrm++;
        } else if (literal(ptr, "f ")) {
            skipSpace(ptr);
            vertexTmp.clear();
            while (*ptr) {
                S32 ptn[3] = {0, 0, 0};
                if (!parseInt(ptr, ptn[0])) break;
                for (int i = 1; i < 4 && literal(ptr, "/"); i++) {
                    S32 tmp = 0;
                    parseInt(ptr, tmp);
                    if (i < 3) ptn[i] = tmp;
                }
                skipSpace(ptr);
                const S32 size[3] = {(S32)positions.size(), numTex, numNrm};
                for (int i = 0; i < 3; i++) {
                    if (ptn[i] < 0) ptn[i] += size[i]; else ptn[i]--;
                    if (ptn[i] < 0 || ptn[i] >= size[i]) ptn[i] = -1;
                }
                auto key = std::make_tuple(ptn[0], ptn[1], ptn[2]);
                auto it = vertexHash.find(key);
                if (it != vertexHash.end()) vertexTmp.push_back(it->second);
                else {
                    S32 idx = (S32)mesh.vertices.size();
                    vertexHash[key] = idx;
                    vertexTmp.push_back(idx);
                    mesh.vertices.push_back((ptn[0] == -1) ? Vec3f(0.0f) : positions[ptn[0]]);
                }
            }
            if (!*ptr) {
                if (submesh == -1) {
                    if (defaultSubmesh == -1) defaultSubmesh = addSubmesh("");
                    submesh = defaultSubmesh;
                }
                for (size_t i = 2; i < vertexTmp.size(); i++) indexTmp.push_back(Vec3i(vertexTmp[0], vertexTmp[i - 1], vertexTmp[i]));
            }
        } else if (literal(ptr, "usemtl ")) {
            skipSpace(ptr);
            if (submesh != -1) {
                std::vector<Vec3i>& dst = mesh.submeshes[submesh];
                dst.insert(dst.end(), indexTmp.begin(), indexTmp.end());
                indexTmp.clear();
                submesh = -1;
            }
            auto it = materialSubmesh.find(String(ptr));
            if (it != materialSubmesh.end()) {
                if (it->second == -1) it->second = addSubmesh(it->first);
                submesh = it->second;
                indexTmp.clear();
            }
        }
    }
    if (submesh != -1) {  // flush (MeshWavefrontIO.cpp end of import)
        std::vector<Vec3i>& dst = mesh.submeshes[submesh];
        dst.insert(dst.end(), indexTmp.begin(), indexTmp.end());
    }
    return true;
}

bool importWavefrontMesh(WavefrontMesh& mesh, const String& fileName)
{
    std::ifstream f(fileName.c_str(), std::ios::binary);
    if (!f) { setError("importWavefrontMesh: cannot open '%s'", fileName.c_str()); return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    const String text = ss.str();
    // materials defined by the mtllib files next to the OBJ (names only)
    std::vector<String> mats;
    String dir;
    size_t slash = fileName.find_last_of("/\\");
    if (slash != String::npos) dir = fileName.substr(0, slash);
    std::istringstream in(text);
    String line;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
        const char* p = line.c_str();
        skipSpace(p);
        if (literal(p, "mtllib ")) {
            skipSpace(p);
            std::ifstream mf(((dir.empty() ? String("") : dir + "/") + p).c_str());
            String ml;
            while (mf && std::getline(mf, ml)) {
                while (!ml.empty() && (ml.back() == '\r' || ml.back() == ' ')) ml.pop_back();
                const char* q = ml.c_str();
                skipSpace(q);
                if (literal(q, "newmtl ")) { skipSpace(q); mats.push_back(String(q)); }
            }
        }
    }
    return parseWavefrontMesh(mesh, text, mats);
}

}  // namespace FW
