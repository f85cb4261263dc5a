// Defs.hpp -- the few base definitions of the reference's framework that the
// tracer-backend API surface needs (types, error model, small vector math).
//
// Mirrors, by name and behaviour (not by code):
//   src/framework/base/Defs.hpp:100-151  (S32/U32/F32..., setError/hasError/clearError, fail)
//   src/framework/base/Math.hpp           (Vec2i/Vec3i/Vec3f/Vec4f/Mat4f subset)
//   src/rt/Util.hpp:35-87                 (AABB, Ray, RayResult)
#pragma once
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <string>

namespace FW {

typedef uint8_t  U8;
typedef int32_t  S32;
typedef uint32_t U32;
typedef int64_t  S64;
typedef uint64_t U64;
typedef float    F32;
typedef double   F64;
typedef std::string String;  // the reference's FW::String is replaced by std::string

#define FW_F32_MAX (3.402823466e+38f)

// Error model of src/framework/base/Defs.hpp:142-151: a sticky error string plus
// a fatal fail().  fail() throws FW::FatalError instead of exit(1) so that a host
// application (and the tests) can observe it.
struct FatalError {
    std::string message;
};
void          fail(const char* fmt, ...) __attribute__((format(printf, 1, 2), noreturn));
void          setError(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
bool          hasError(void);
const String& getError(void);
void          clearError(void);
void          failIfError(void);

// generic FW::min/FW::max are selects (Defs.hpp:212-213)
template <class T> inline T min(T a, T b) { return (a < b) ? a : b; }
template <class T> inline T max(T a, T b) { return (a > b) ? a : b; }

inline U32 floatToBits(F32 a) { U32 u; memcpy(&u, &a, 4); return u; }
inline F32 bitsToFloat(U32 a) { F32 f; memcpy(&f, &a, 4); return f; }

struct Vec2i { S32 x, y; Vec2i(S32 a = 0, S32 b = 0) : x(a), y(b) {} };
struct Vec3i {
    S32 x, y, z;
    Vec3i(S32 a = 0, S32 b = 0, S32 c = 0) : x(a), y(b), z(c) {}
    S32 operator[](int i) const { return (&x)[i]; }
};

struct Vec3f {
    F32 x, y, z;
    Vec3f(F32 a = 0.0f) : x(a), y(a), z(a) {}
    Vec3f(F32 a, F32 b, F32 c) : x(a), y(b), z(c) {}
    F32  operator[](int i) const { return (&x)[i]; }
    F32& operator[](int i) { return (&x)[i]; }
    Vec3f operator+(const Vec3f& v) const { return Vec3f(x + v.x, y + v.y, z + v.z); }
    Vec3f operator-(const Vec3f& v) const { return Vec3f(x - v.x, y - v.y, z - v.z); }
    Vec3f operator*(F32 s) const { return Vec3f(x * s, y * s, z * s); }
    Vec3f min(const Vec3f& v) const { return Vec3f(FW::min(x, v.x), FW::min(y, v.y), FW::min(z, v.z)); }
    Vec3f max(const Vec3f& v) const { return Vec3f(FW::max(x, v.x), FW::max(y, v.y), FW::max(z, v.z)); }
    F32 min(void) const { return FW::min(FW::min(x, y), z); }
    F32 max(void) const { return FW::max(FW::max(x, y), z); }
    F32 sum(void) const { return x + y + z; }
};
inline Vec3f cross(const Vec3f& a, const Vec3f& b)
{
    return Vec3f(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

struct Vec4f {
    F32 x, y, z, w;
    Vec4f(F32 a = 0.0f) : x(a), y(a), z(a), w(a) {}
    Vec4f(F32 a, F32 b, F32 c, F32 d) : x(a), y(b), z(c), w(d) {}
    Vec4f(const Vec3f& v, F32 d) : x(v.x), y(v.y), z(v.z), w(d) {}
};

// src/rt/Util.hpp:35-59
class AABB {
public:
    AABB(void) : m_mn(FW_F32_MAX, FW_F32_MAX, FW_F32_MAX), m_mx(-FW_F32_MAX, -FW_F32_MAX, -FW_F32_MAX) {}
    AABB(const Vec3f& mn, const Vec3f& mx) : m_mn(mn), m_mx(mx) {}
    void  grow(const Vec3f& pt) { m_mn = m_mn.min(pt); m_mx = m_mx.max(pt); }
    void  grow(const AABB& aabb) { grow(aabb.m_mn); grow(aabb.m_mx); }
    bool  valid(void) const { return m_mn.x <= m_mx.x && m_mn.y <= m_mx.y && m_mn.z <= m_mx.z; }
    F32   area(void) const
    {
        if (!valid()) return 0.0f;
        Vec3f d = m_mx - m_mn;
        return (d.x * d.y + d.y * d.z + d.z * d.x) * 2.0f;
    }
    const Vec3f& min(void) const { return m_mn; }
    const Vec3f& max(void) const { return m_mx; }
    Vec3f&       min(void) { return m_mn; }
    Vec3f&       max(void) { return m_mx; }

private:
    Vec3f m_mn, m_mx;
};

// src/rt/Util.hpp:62-71
struct Ray {
    Ray(void) : origin(0.0f), tmin(0.0f), direction(0.0f), tmax(0.0f) {}
    void degenerate(void) { tmax = tmin - 1.0f; }
    Vec3f origin;
    F32   tmin;
    Vec3f direction;
    F32   tmax;
};

// src/rt/Util.hpp:75-87
#define RAY_NO_HIT (-1)
struct RayResult {
    RayResult(S32 ii = RAY_NO_HIT, F32 ti = 0.f) : id(ii), t(ti), padA(0), padB(0) {}
    bool hit(void) const { return id != RAY_NO_HIT; }
    void clear(void) { id = RAY_NO_HIT; }
    S32 id;
    F32 t;
    S32 padA;
    S32 padB;
};

static_assert(sizeof(Ray) == 32, "Ray must be 32 bytes (Util.hpp:62-71)");
static_assert(sizeof(RayResult) == 16, "RayResult must be 16 bytes (Util.hpp:77-87)");

}  // namespace FW
