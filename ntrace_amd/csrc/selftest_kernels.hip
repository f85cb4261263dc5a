// selftest_kernels.hip -- device self tests of the FAST slab path's exact division (trace_arith.h): FAST == the hardware's correctly
// rounded `/`, bit for bit.  ntr_selftest_division / ntr_selftest_division_hard in the C-ABI (tests/test_trace_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trace_kernels.h"
#include "trace_arith.h"

namespace ntr {

// ---------------------------------------------------------------------------------
// Self test: FAST division == GENERIC division, bit for bit, on device.
// mismatches += number of differing quotients among x[i] / d[j] for all i, j.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void selftest_division_kernel(const float* __restrict__ x, const float* __restrict__ d,
                                                                int nx, int nd, unsigned int* __restrict__ mismatches)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx) return;
    const float xv = x[i];
    unsigned int bad = 0;
    for (int j = 0; j < nd; j++) {
        const float dv = d[j];
        const float q0 = xv / dv;
        const float q1 = fast_div(xv, dv, exact_rcp(dv));
        // the sign of a zero quotient is not observable by the tracer's comparisons
        bad += (__float_as_uint(q0) != __float_as_uint(q1)) && !(q0 == 0.0f && q1 == 0.0f);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// The hardest quotients for the one-correction divide, enumerated on the device: for every significand D in [2^23, 2^24) the X whose
// quotient X / D lies closest to a rounding boundary -- 2^b X - D Mo = N for a midpoint Mo (odd, 25 bits), b = 24 (X >= D) or 25 (X < D),
// every N with |N| <= 8 the equation admits -- what scripts/studies/div_one_correction_check.py checks in exact integer arithmetic.
// Here the hardware runs them: FAST divide against `/`, with x and d scaled by the powers of two [xe0, xe0 + 3] x [de0, de0 + 3]
// (the result may not depend on them inside the FASTDIV range).  One thread per D; counts[0] += pairs tested, counts[1] += mismatches.
__global__ __launch_bounds__(256) void selftest_division_hard_kernel(int xe0, int de0, unsigned long long* __restrict__ counts)
{
    const unsigned int D = (1u << 23) + blockIdx.x * 256u + threadIdx.x;   // grid: 2^23 / 256 workgroups
    const int k = __builtin_ctz(D);
    unsigned long long tested = 0, bad = 0;
    if (k <= 3) {   // (a D divisible by 16 admits no |N| <= 8)
        const unsigned int Dp = D >> k;
        unsigned int inv = Dp;   // Dp^-1 mod 2^32 (Newton: every step doubles the valid bits, 3 to start with)
        for (int it = 0; it < 5; it++) inv *= 2u - Dp * inv;
        for (int b = 24; b <= 25; b++) {
            const unsigned long long mod = 1ull << (b - k);
            for (int Np = -8; Np <= 8; Np++) {
                const long long N = (long long)Np * (1ll << k);
                if (Np == 0 || N > 8 || N < -8) continue;
                const unsigned long long base = ((unsigned long long)(unsigned int)(-Np) * inv) & (mod - 1ull);   // Dp Mo = -N' (mod 2^(b-k))
                for (unsigned int j = 0; j < (1u << k); j++) {
                    unsigned long long Mo = base + j * mod;
                    while (Mo < (1ull << 24)) Mo += 1ull << b;
                    if (Mo >= (1ull << 25) || !(Mo & 1ull)) continue;
                    const long long num = (long long)((unsigned long long)D * Mo) + N;
                    if (num & ((1ll << b) - 1ll)) continue;
                    const long long X = num >> b;
                    if (b == 24 ? (X < (long long)D || X >= (1ll << 24)) : (X < (1ll << 23) || X >= (long long)D)) continue;
                    for (int e = 0; e < 16; e++) {
                        const float xv = ldexpf((float)X, xe0 + (e & 3) - 23), dv = ldexpf((float)D, de0 + (e >> 2) - 23);
                        for (int sgn = 0; sgn < 2; sgn++) {
                            const float dd = sgn ? -dv : dv;
                            const float q0 = xv / dd;
                            const float q1 = fast_div(xv, dd, exact_rcp(dd));
                            tested++;
                            bad += __float_as_uint(q0) != __float_as_uint(q1);
                        }
                    }
                }
            }
        }
    }
    if (tested) atomicAdd(&counts[0], tested);
    if (bad) atomicAdd(&counts[1], bad);
}

}  // namespace ntr

extern "C" hipError_t ntr_launch_selftest_division(const float* d_x, const float* d_d, int nx, int nd,
                                                   unsigned int* d_mismatches, hipStream_t stream)
{
    hipLaunchKernelGGL(ntr::selftest_division_kernel, dim3((nx + 255) / 256), dim3(256), 0, stream, d_x, d_d, nx, nd, d_mismatches);
    return hipGetLastError();
}

extern "C" hipError_t ntr_launch_selftest_division_hard(int xe0, int de0, unsigned long long* d_counts, hipStream_t stream)
{
    hipLaunchKernelGGL(ntr::selftest_division_hard_kernel, dim3((1u << 23) / 256u), dim3(256), 0, stream, xe0, de0, d_counts);
    return hipGetLastError();
}
