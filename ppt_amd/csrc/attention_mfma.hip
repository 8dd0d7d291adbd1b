// attention_mfma.hip -- bf16 flash-attention forward on the gfx950 matrix cores (placeholder
// dispatch: until the MFMA kernel lands every shape goes to the quad kernel of attention.hip).
#include "ppt_common.h"

extern "C" int ppt_attention_fwd_quad_bf16(const void *qkv, void *out, float *lse, int Bt, int T, int H, float scale,
                                           int causal, hipStream_t s);

extern "C" int ppt_attention_fwd_mfma_bf16(const void *qkv, void *out, float *lse, int Bt, int T, int H, float scale,
                                           int causal, hipStream_t s)
{
    return ppt_attention_fwd_quad_bf16(qkv, out, lse, Bt, T, H, scale, causal, s);
}
