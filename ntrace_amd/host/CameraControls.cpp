#include "CameraControls.hpp"

#include <cmath>

#include "Renderer.hpp"

namespace FW {

CameraControls::CameraControls(void)
    : m_position(0.0f, 0.0f, 1.5f), m_forward(0.0f, 0.0f, -1.0f), m_up(0.0f, 1.0f, 0.0f), m_speed(0.25f), m_fov(70.0f),
      m_near(0.001f), m_far(3.0f), m_keepAligned(false)
{
}

// ---- signature codec (CameraControls.cpp:342-399, 471-545) ---------------------------------------
static void encodeBits(String& dst, U32 v)
{
    int base = (v < 12) ? '/' : (v < 38) ? 'A' - 12 : 'a' - 38;
    dst += (char)(v + base);
}
static U32 decodeBits(const char*& src)
{
    if (*src >= '/' && *src <= ':') return *src++ - '/';
    if (*src >= 'A' && *src <= 'Z') return *src++ - 'A' + 12;
    if (*src >= 'a' && *src <= 'z') return *src++ - 'a' + 38;
    setError("CameraControls: Invalid signature!");
    return 0;
}
static void encodeFloat(String& dst, F32 v)
{
    U32 bits = floatToBits(v);
    for (int i = 0; i < 32; i += 6) encodeBits(dst, (bits >> i) & 0x3F);
}
static F32 decodeFloat(const char*& src)
{
    U32 bits = 0;
    for (int i = 0; i < 32; i += 6) bits |= decodeBits(src) << i;
    return bitsToFloat(bits);
}
static void encodeDirection(String& dst, const Vec3f& v)
{
    Vec3f a(std::fabs(v.x), std::fabs(v.y), std::fabs(v.z));
    int axis = (a.x >= FW::max(a.y, a.z)) ? 0 : (a.y >= a.z) ? 1 : 2;
    Vec3f tuv;
    switch (axis) {
    case 0: tuv = v; break;
    case 1: tuv = Vec3f(v.y, v.z, v.x); break;
    default: tuv = Vec3f(v.z, v.x, v.y); break;
    }
    int face = axis | ((tuv.x >= 0.0f) ? 0 : 4);
    if (tuv.y == 0.0f && tuv.z == 0.0f) {
        encodeBits(dst, face | 8);
        return;
    }
    encodeBits(dst, face);
    encodeFloat(dst, tuv.y / std::fabs(tuv.x));
    encodeFloat(dst, tuv.z / std::fabs(tuv.x));
}
static Vec3f normalized(const Vec3f& v)
{
    F32 len = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return v * (1.0f * (1.0f / len));  // VectorBase::normalized (Math.hpp:141)
}
static Vec3f decodeDirection(const char*& src)
{
    int face = decodeBits(src);
    Vec3f tuv;
    tuv.x = ((face & 4) == 0) ? 1.0f : -1.0f;
    tuv.y = ((face & 8) == 0) ? decodeFloat(src) : 0.0f;
    tuv.z = ((face & 8) == 0) ? decodeFloat(src) : 0.0f;
    tuv = normalized(tuv);
    switch (face & 3) {
    case 0: return tuv;
    case 1: return Vec3f(tuv.z, tuv.x, tuv.y);
    default: return Vec3f(tuv.y, tuv.z, tuv.x);
    }
}

String CameraControls::encodeSignature(void) const
{
    String sig;
    sig += '"';
    encodeFloat(sig, m_position.x);
    encodeFloat(sig, m_position.y);
    encodeFloat(sig, m_position.z);
    encodeDirection(sig, m_forward);
    encodeDirection(sig, m_up);
    encodeFloat(sig, m_speed);
    encodeFloat(sig, m_fov);
    encodeFloat(sig, m_near);
    encodeFloat(sig, m_far);
    encodeBits(sig, m_keepAligned ? 1 : 0);
    sig += "\",";
    return sig;
}

void CameraControls::decodeSignature(const String& sig)
{
    const char* src = sig.c_str();
    while (*src == ' ' || *src == '\t' || *src == '\n') src++;
    if (*src == '"') src++;
    F32 px = decodeFloat(src), py = decodeFloat(src), pz = decodeFloat(src);
    Vec3f forward = decodeDirection(src);
    Vec3f up = decodeDirection(src);
    F32 speed = decodeFloat(src), fov = decodeFloat(src), znear = decodeFloat(src), zfar = decodeFloat(src);
    bool keepAligned = (decodeBits(src) != 0);
    if (*src == '"') src++;
    if (*src == ',') src++;
    while (*src == ' ' || *src == '\t' || *src == '\n') src++;
    if (*src) setError("CameraControls: Invalid signature!");
    if (hasError()) return;
    m_position = Vec3f(px, py, pz);
    m_forward = forward;
    m_up = up;
    m_speed = speed;
    m_fov = fov;
    m_near = znear;
    m_far = zfar;
    m_keepAligned = keepAligned;
}

// ---- matrices ----------------------------------------------------------------------------------------
Mat4f mat4Mul(const Mat4f& a, const Mat4f& b)
{
    Mat4f r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            F32 s = 0.0f;
            for (int k = 0; k < 4; k++) s += a.m[4 * i + k] * b.m[4 * k + j];
            r.m[4 * i + j] = s;
        }
    return r;
}

static F32 det3(const F32 v[3][3])
{
    return v[0][0] * v[1][1] * v[2][2] - v[0][0] * v[1][2] * v[2][1] + v[1][0] * v[2][1] * v[0][2] - v[1][0] * v[2][2] * v[0][1] +
           v[2][0] * v[0][1] * v[1][2] - v[2][0] * v[0][2] * v[1][1];
}

Mat4f mat4Inverted(const Mat4f& a)
{
    Mat4f r;
    F32 d = 0.0f, si = 1.0f;
    for (int i = 0; i < 4; i++) {
        F32 sj = si;
        for (int j = 0; j < 4; j++) {
            F32 sub[3][3];
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 3; l++) sub[k][l] = a.m[4 * ((k < j) ? k : k + 1) + ((l < i) ? l : l + 1)];
            F32 dd = det3(sub) * sj;
            r.m[4 * i + j] = dd;
            d += dd * a.m[4 * j + i];
            sj = -sj;
        }
        si = -si;
    }
    F32 rd = 1.0f / d;
    for (int i = 0; i < 16; i++) r.m[i] = r.m[i] * rd * 4.0f;
    return r;
}

// CameraControls::getOrientation / getWorldToCamera (CameraControls.cpp:251-284)
Mat4f CameraControls::getWorldToCamera(void) const
{
    Vec3f c2 = normalized(m_forward) * -1.0f;
    Vec3f c0 = normalized(cross(m_up, c2));
    Vec3f c1 = normalized(cross(c2, c0));
    // pos = orient^T * position
    Vec3f pos(c0.x * m_position.x + c0.y * m_position.y + c0.z * m_position.z,
              c1.x * m_position.x + c1.y * m_position.y + c1.z * m_position.z,
              c2.x * m_position.x + c2.y * m_position.y + c2.z * m_position.z);
    Mat4f r;
    const F32 m[16] = {c0.x, c0.y, c0.z, -pos.x, c1.x, c1.y, c1.z, -pos.y, c2.x, c2.y, c2.z, -pos.z, 0, 0, 0, 1};
    memcpy(r.m, m, sizeof(m));
    return r;
}

// Mat4f::perspective (base/Math.cpp:79-92)
Mat4f CameraControls::getCameraToClip(void) const
{
    F32 f = 1.0f / std::tan(m_fov * 3.14159265358979323846f / 360.0f);
    F32 d = 1.0f / (m_near - m_far);
    Mat4f r;
    const F32 m[16] = {f, 0, 0, 0, 0, f, 0, 0, 0, 0, (m_near + m_far) * d, 2.0f * m_near * m_far * d, 0, 0, -1.0f, 0};
    memcpy(r.m, m, sizeof(m));
    return r;
}

Mat4f CameraControls::getWorldToClip(void) const { return mat4Mul(getCameraToClip(), getWorldToCamera()); }

// Mat4f::fitToView(pos = -1, size = 2, viewSize) (base/Math.cpp:66-75):
// scale(2 / viewSize) * scale(min(viewSize / size)) * translate(-pos - size / 2); the translate is 0 here.
Mat4f CameraControls::getNScreenToWorld(S32 viewW, S32 viewH) const
{
    const F32 vw = (F32)viewW, vh = (F32)viewH;
    const F32 s = FW::min(vw / 2.0f, vh / 2.0f);
    Mat4f fit;
    const F32 m[16] = {(2.0f / vw) * s, 0, 0, 0, 0, (2.0f / vh) * s, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(fit.m, m, sizeof(m));
    return mat4Inverted(mat4Mul(fit, getWorldToClip()));
}

CameraView CameraControls::getView(S32 viewW, S32 viewH) const
{
    CameraView v;
    v.position = m_position;
    v.nscreenToWorld = getNScreenToWorld(viewW, viewH);
    v.cameraFar = m_far;
    v.width = viewW;
    v.height = viewH;
    return v;
}

}  // namespace FW
