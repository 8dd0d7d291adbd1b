// ppt_act.h -- activation helpers shared by the GEMM epilogues (gemm.hip, rowgemm.hip).
#pragma once
#include "ppt_common.h"

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): 1 v_rcp + 1 v_exp + 7 FMA instead of libm's branchy
// erff; used when the result is rounded to bf16 anyway (8 significand bits), never in the fp32 parity mode.
// exact-erf GELU for results that are rounded to bf16: erf(x / sqrt 2) as x P(x^2) on |x| < 4 (a near-minimax odd polynomial, 8
// coefficients: |error| <= 4.3e-5), +-1 beyond (1 - erf there: 6.3e-5).  12 full-rate VALU operations and NO transcendental
// (erf_fast below: v_rcp + v_exp at quarter rate, ~26 issue cycles) -- the GELU phase is a quarter of the fused MLP kernel.
// |gelu error| <= 2e-4 absolute, i.e. <= 1 % of a bf16 ulp for outputs of O(1).  Never used in the fp32 parity mode.
__device__ __forceinline__ float gelu_poly(float x)
{
    const float u = x * x;
    float p = fmaf(-3.161567230e-09f, u, 2.434219084e-07f);
    p = fmaf(p, u, -8.201724995e-06f);
    p = fmaf(p, u, 1.613346976e-04f);
    p = fmaf(p, u, -2.096408280e-03f);
    p = fmaf(p, u, 1.932974532e-02f);
    p = fmaf(p, u, -1.323507577e-01f);
    p = fmaf(p, u, 7.976950407e-01f);
    float e = p * x;
    e = fabsf(x) >= 4.0f ? copysignf(1.0f, x) : e;
    const float h = 0.5f * x;
    return fmaf(h, e, h);
}

__device__ __forceinline__ float erf_fast(float x)
{
    const float a = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = 1.0f - poly * t * __expf(-a * a);
    return copysignf(e, x);
}
