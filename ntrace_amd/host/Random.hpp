// Random.hpp -- RANROT-A generator of the reference framework (src/framework/base/Random.cpp:47-83),
// needed because ray-generation seeds are `Random(seed).getU32()` (src/rt/ray/RayGen.cpp:66, 220).
#pragma once
#include "Defs.hpp"

namespace FW {

class Random {
public:
    explicit Random(U32 seed = 0) { reset(seed); }
    void reset(U32 seed);
    U32  getU32(void);

private:
    S32 m_p1, m_p2;
    U32 m_buffer[11];
};

}  // namespace FW
