// Hash.hpp -- the hash helpers the reference keys its BVH cache files with (src/framework/base/Hash.hpp:169-189,
// Hash.cpp:34-112): Bob Jenkins' 1996 "lookup2" mix (public algorithm, burtleburtle.net/bob/hash/doobs.html) over three
// 32-bit lanes seeded with the golden-ratio constant.  hashBits / hashBuffer give the reference's values for the same input,
// so Platform::computeHash and BVH::BuildParams::computeHash are the reference's numbers.  Parity unpinned: the reference's Hash.hpp does
// not compile here (it pulls cuda.h through DLLImports.hpp) and its tail/length handling differs from Jenkins' published hash(), so no
// published vector applies; the values only name cache files.
#pragma once
#include "Defs.hpp"

namespace FW {

namespace detail {
const U32 kGolden = 0x9e3779b9u;
inline void lookup2(U32& a, U32& b, U32& c)
{
    a -= b; a -= c; a ^= (c >> 13);
    b -= c; b -= a; b ^= (a << 8);
    c -= a; c -= b; c ^= (b >> 13);
    a -= b; a -= c; a ^= (c >> 12);
    b -= c; b -= a; b ^= (a << 16);
    c -= a; c -= b; c ^= (b >> 5);
    a -= b; a -= c; a ^= (c >> 3);
    b -= c; b -= a; b ^= (a << 10);
    c -= a; c -= b; c ^= (b >> 15);
}
}  // namespace detail

inline U32 hashBits(U32 a, U32 b = detail::kGolden, U32 c = 0)
{
    c += detail::kGolden;
    detail::lookup2(a, b, c);
    return c;
}
inline U32 hashBits(U32 a, U32 b, U32 c, U32 d, U32 e = 0, U32 f = 0)
{
    c += detail::kGolden;
    detail::lookup2(a, b, c);
    a += d; b += e; c += f;
    detail::lookup2(a, b, c);
    return c;
}

// bytes are consumed as little-endian 32-bit words, 12 at a time; the tail and the length go into the last round
inline U32 hashBuffer(const void* ptr, S64 size)
{
    const U8* src = (const U8*)ptr;
    U32 lane[3] = {detail::kGolden, detail::kGolden, detail::kGolden};
    while (size >= 12) {
        for (int k = 0; k < 3; k++)
            lane[k] += (U32)src[4 * k] | ((U32)src[4 * k + 1] << 8) | ((U32)src[4 * k + 2] << 16) | ((U32)src[4 * k + 3] << 24);
        detail::lookup2(lane[0], lane[1], lane[2]);
        src += 12;
        size -= 12;
    }
    for (S64 i = 0; i < size; i++) lane[i >> 2] += (U32)src[i] << (8 * (i & 3));
    lane[2] += (U32)size;
    detail::lookup2(lane[0], lane[1], lane[2]);
    return lane[2];
}
inline U32 hashString(const String& s) { return hashBuffer(s.c_str(), (S64)s.size()); }

}  // namespace FW
