// attn_rowmap.h -- row addressing shared by the attention kernels (attention.hip, attention_mfma.hip).
//
// Plain layout (P == 0): sequence v of Bt, position pos -> row v * T + pos of the packed [Bt * T, 3, H, 64] qkv tensor;
// softmax statistics (LSE, delta) at [(v * H + head) * T + pos].
//
// Prefix-shared layout (P > 0; the CLIP text tower under PromptLearner, ULIP_models.py:104-151,203-230): the first P
// positions of all C prompts are IDENTICAL (start token + the leading learnable context tokens) and the attention is causal,
// so their activations are the same in every prompt and at every layer.  They are stored once:
//     rows [0, P)                          the shared prefix, positions 0 .. P-1
//     rows [P + c (T - P), P + (c+1)(T - P))   prompt c, positions P .. T-1
// The kernels run C + 1 virtual sequences: v < C is prompt v (keys: the shared rows, then its own; QUERIES it owns:
// positions >= P), v == C is the prefix itself (length P).  Statistics are indexed by physical row: [row * H + head].
#pragma once
#include <stdint.h>

__device__ __forceinline__ int64_t am_row(int T, int P, int v, int pos)
{
    return P == 0 ? (int64_t)v * T + pos : (pos < P ? (int64_t)pos : (int64_t)P + (int64_t)v * (T - P) + (pos - P));
}
__device__ __forceinline__ int am_len(int T, int P, int C, int v) { return (P > 0 && v == C) ? P : T; }    // keys / queries of v
__device__ __forceinline__ int am_qlo(int P, int C, int v) { return (P > 0 && v < C) ? P : 0; }             // first query v owns
__device__ __forceinline__ int64_t am_stat(int T, int P, int H, int v, int head, int pos)
{
    return P == 0 ? ((int64_t)v * H + head) * T + pos : am_row(T, P, v, pos) * H + head;
}
