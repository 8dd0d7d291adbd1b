// ppt_act.h -- activation helpers shared by the GEMM epilogues (gemm.hip, rowgemm.hip).
#pragma once
#include "ppt_common.h"

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): 1 v_rcp + 1 v_exp + 7 FMA instead of libm's branchy
// erff; used when the result is rounded to bf16 anyway (8 significand bits), never in the fp32 parity mode.
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
