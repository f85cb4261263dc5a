#include "CameraControls.hpp"

#include <cmath>

#include "Renderer.hpp"

namespace FW {

CameraControls::CameraControls(void)
    : m_position(0.0f, 0.0f, 1.5f), m_forward(0.0f, 0.0f, -1.0f), m_up(0.0f, 1.0f, 0.0f), m_speed(0.25f), m_fov(70.0f),
      m_near(0.001f), m_far(3.0f), m_keepAligned(false)
{
}

// ---- camera signature codec ------------------------------------------------------------------------
// Wire format (what the reference's CameraControls::encodeSignature / decodeSignature exchange, framework/3d/CameraControls.cpp:342-399,
// 471-545): a quoted string of base-64 digits, ALPHABET below (digit values 0..63 = '/', '0'..'9', ':', 'A'..'Z',
// 'a'..'z').  A float is its 32 bits as six digits, least significant digit first (the last digit carries two bits).  A direction is one
// digit -- bits 0-1: dominant axis d, bit 2: that component is negative, bit 3: the other two components are exactly zero -- followed,
// unless bit 3 is set, by two floats: components d+1 and d+2 (cyclically) divided by |component d|.  Fields: position x y z, forward,
// up, speed, fov, near, far, one digit keepAligned; the writer closes with `",`.
namespace {

const char ALPHABET[65] = "/0123456789:ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz";

struct DigitTable {   // character -> digit value, -1 for characters outside the alphabet
    signed char value[256];
    DigitTable()
    {
        for (int c = 0; c < 256; c++) value[c] = -1;
        for (int d = 0; d < 64; d++) value[(unsigned char)ALPHABET[d]] = (signed char)d;
    }
};

class SignatureWriter {
public:
    void digit(U32 d) { m_text += ALPHABET[d & 63u]; }
    void real(F32 v)
    {
        U32 rest = floatToBits(v);
        for (int k = 0; k < 6; k++, rest >>= 6) digit(rest);
    }
    void direction(const Vec3f& v)
    {
        const F32 comp[3] = {v.x, v.y, v.z};
        int d = 0;   // dominant axis; ties go to the earlier axis
        for (int k = 1; k < 3; k++)
            if (std::fabs(comp[k]) > std::fabs(comp[d])) d = k;
        const F32 major = comp[d], u = comp[(d + 1) % 3], w = comp[(d + 2) % 3];
        const U32 face = (U32)d | ((major >= 0.0f) ? 0u : 4u);
        if (u == 0.0f && w == 0.0f) {   // axis-aligned: the face digit alone
            digit(face | 8u);
            return;
        }
        digit(face);
        real(u / std::fabs(major));
        real(w / std::fabs(major));
    }
    void raw(const char* t) { m_text += t; }
    const String& text() const { return m_text; }

private:
    String m_text;
};

class SignatureReader {
public:
    explicit SignatureReader(const char* p) : m_at(p), m_bad(false) {}
    void skipBlanks()
    {
        while (*m_at == ' ' || *m_at == '\t' || *m_at == '\n') m_at++;
    }
    void skipIf(char c)
    {
        if (*m_at == c) m_at++;
    }
    U32 digit()
    {
        static const DigitTable table;
        const int d = table.value[(unsigned char)*m_at];
        if (d < 0) {   // (a terminating NUL is not consumed)
            m_bad = true;
            return 0;
        }
        m_at++;
        return (U32)d;
    }
    F32 real()
    {
        U32 bits = 0;
        for (int k = 0; k < 6; k++) bits |= digit() << (6 * k);
        return bitsToFloat(bits);
    }
    Vec3f direction()
    {
        const U32 face = digit();
        const bool axisAligned = (face & 8u) != 0;
        F32 t[3];   // (major, next, next-next) in cyclic order from the dominant axis
        t[0] = (face & 4u) ? -1.0f : 1.0f;
        t[1] = axisAligned ? 0.0f : real();
        t[2] = axisAligned ? 0.0f : real();
        const F32 len = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        const F32 scale = 1.0f / len;   // the reference's normalized(): v * (1 * rcp(length)) (Math.hpp:141)
        const int d = (face & 3u) < 2u ? (int)(face & 3u) : 2;   // (axis code 3 reads as the z face, as in the reference)
        F32 out[3];
        for (int k = 0; k < 3; k++) out[(d + k) % 3] = t[k] * scale;
        return Vec3f(out[0], out[1], out[2]);
    }
    bool atEnd() const { return *m_at == 0; }
    bool bad() const { return m_bad; }

private:
    const char* m_at;
    bool m_bad;
};

}  // namespace

String CameraControls::encodeSignature(void) const
{
    SignatureWriter w;
    w.raw("\"");
    w.real(m_position.x);
    w.real(m_position.y);
    w.real(m_position.z);
    w.direction(m_forward);
    w.direction(m_up);
    w.real(m_speed);
    w.real(m_fov);
    w.real(m_near);
    w.real(m_far);
    w.digit(m_keepAligned ? 1u : 0u);
    w.raw("\",");
    return w.text();
}

void CameraControls::decodeSignature(const String& sig)
{
    SignatureReader r(sig.c_str());
    r.skipBlanks();
    r.skipIf('"');
    CameraControls parsed(*this);   // nothing of *this changes unless the whole signature parses
    parsed.m_position.x = r.real();
    parsed.m_position.y = r.real();
    parsed.m_position.z = r.real();
    parsed.m_forward = r.direction();
    parsed.m_up = r.direction();
    parsed.m_speed = r.real();
    parsed.m_fov = r.real();
    parsed.m_near = r.real();
    parsed.m_far = r.real();
    parsed.m_keepAligned = r.digit() != 0;
    r.skipIf('"');
    r.skipIf(',');
    r.skipBlanks();
    if (r.bad() || !r.atEnd()) {
        setError("CameraControls: Invalid signature!");
        return;
    }
    if (hasError()) return;
    *this = parsed;
}

static Vec3f normalized(const Vec3f& v)
{
    const F32 len = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return v * (1.0f * (1.0f / len));  // VectorBase::normalized (Math.hpp:141)
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
