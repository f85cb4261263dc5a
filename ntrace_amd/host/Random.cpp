#include "Random.hpp"

namespace FW {

// RanrotA::reset (Random.cpp:59-75): xorshift-seeded lag buffer, then 11 warm-up draws.
void Random::reset(U32 seed)
{
    if (seed == 0) seed--;
    for (int i = 0; i < 11; i++) {
        seed ^= seed << 13;
        seed ^= seed >> 17;
        seed ^= seed << 5;
        m_buffer[i] = seed;
    }
    m_p1 = 0;
    m_p2 = 7;
    for (int i = 0; i < 11; i++) getU32();
}

// RanrotA::get (Random.cpp:77-88)
U32 Random::getU32(void)
{
    U32 x = m_buffer[m_p1] + m_buffer[m_p2];
    x = (x << 13) | (x >> 19);
    m_buffer[m_p1] = x;
    m_p1--;
    m_p1 += (m_p1 >> 31) & 11;
    m_p2--;
    m_p2 += (m_p2 >> 31) & 11;
    return x;
}

}  // namespace FW
