// trace_arith.h -- the FAST slab path's exact division (shared by trace_kernels.hip and the device self tests, selftest_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ntr {

// The CORRECTLY ROUNDED reciprocal of a direction component (the IEEE divide 1 / d: once per ray and axis).  With it ONE residual
// correction makes a quotient correctly rounded:  q0 = x r;  e = fma(-d, q0, x);  q = fma(e, r, q0)  ==  RN(x / d).
// Why: q0 + e r = x/d (1 + theta) exactly, |theta| <~ 4 u^2 (u = 2^-24), so q can differ from RN(x/d) only when x/d lies within that
// distance of a midpoint of two neighbouring floats -- and those pairs are enumerable: for significands X, D and a midpoint Mo / 2^24 the
// distance is |2^24 X - D Mo| / (2^24 D), a non-zero integer over 2^24 D.  scripts/studies/div_one_correction_check.py checks every such
// pair (all D, both quotient binades, |numerator| <= 8: 46.5 M pairs) in exact integer arithmetic: none differs; with a reciprocal one ulp
// off 14 % of them do (which is why the hardware divide's own chain -- v_rcp refined once, NOT always correctly rounded -- needs the two
// corrections this path used until round 4).  ntr_selftest_division() checks FAST == GENERIC on the device, those pairs included.
__device__ __forceinline__ float exact_rcp(float d) { return 1.0f / d; }
__device__ __forceinline__ float fast_div(float x, float d, float r)
{
    const float q0 = x * r;
    const float e1 = __builtin_fmaf(-d, q0, x);
    const float q1 = __builtin_fmaf(e1, r, q0);
    return q1;
}

}  // namespace ntr
