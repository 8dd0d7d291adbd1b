/*
 * ppt_hip.h -- C ABI of libppt_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the PPT
 * point-cloud encoder hot path.  The reference has NO native code on this
 * path (it is a sequence of PyTorch ops); each entry point below therefore names the reference
 * Python it replaces (file:line relative to the upstream repo) instead of an FFI symbol.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer to a contiguous row-major array; the library never
 *     allocates, frees or retains device memory -- outputs and workspaces are caller-owned;
 *   - kernels are enqueued on the caller's stream (`stream` is a hipStream_t passed as void*;
 *     NULL = the default stream); no device-wide synchronisation inside; graph-capturable;
 *   - return 0 on success, a negative PPT_E* code otherwise (never exit(), never throws);
 *     ppt_strerror() names the code;
 *   - re-entrant and thread-safe (no global mutable state).
 *   - `dtype`: PPT_BF16 = bf16 operands / fp32 accumulate on the MFMA pipes (performance mode),
 *     PPT_F32 = fp32 operands on the fp32 MFMA / VALU (parity mode).  Index outputs are int64,
 *     as in the reference.
 */
#ifndef PPT_HIP_H
#define PPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPT_OK 0
#define PPT_EINVAL (-1)   /* bad argument (shape, alignment, NULL pointer) */
#define PPT_ELAUNCH (-2)  /* hipGetLastError() after launch reported a failure */
#define PPT_EUNSUPPORTED (-3)

#define PPT_F32 0
#define PPT_BF16 1
#define PPT_F16 2   /* IEEE half operands / fp32 accumulate: the MFMA rate of bf16 with 11 significand bits instead of 8 -- the
                     * operand format of the performance mode's text tower, PointBERT tokenizer + blocks and part-seg decoder
                     * (residual stream and statistics in fp32; every backward stage that carries gradients in half multiplies
                     * what it receives by a power-of-two scale and what it hands out by its inverse -- ppt_convert_scaled, the
                     * alpha of ppt_rows_matmul_f32, ppt_gemm's row_scale, ppt_prompt_rows_bwd's scale; host side:
                     * ppt_amd/gradscale.py); bf16 stays the format of PointNet++ / PointMLP */

const char *ppt_strerror(int code);
/* ABI version of this header (currently 7); bumped on any signature change or added entry point.
 * 7: ppt_gemm_params.split_overflow (new trailing field: split16 saturates finite values beyond IEEE half's range and counts
 *    the workgroups that did), ppt_vit_mlp3_bf16 / ppt_vit_mlp3_retile (new: csrc/mlp_fused3.hip), ppt_text_mlp_pair /
 *    ppt_text_mlp_retile (new: csrc/text_mlp.hip), ppt_lnlin / ppt_lnlin_retile (new: csrc/lnlin.hip),
 *    ppt_text_mlp_retile_split + the split16 fields of ppt_text_mlp_params (csrc/text_mlp_split.hip), ppt_text_lin_split /
 *    ppt_text_lin_retile_split (csrc/text_lin_split.hip).
 * 6: ppt_gemm_params.split16 / split_a_pow2 / split_b_pow2 (new trailing fields: fp32 operands as hi + lo half pairs),
 *    ppt_attention_fwd_split16 / ppt_attention_bwd_split16 (new), ppt_pointmlp_cloud_rstd / ppt_pointmlp_pq (new).
 * 5: ppt_labels_check (new), ppt_gemm256 (new: the 256-row macro-tile GEMM core), ppt_set_gemm256 / ppt_get_gemm256 (new),
 *    ppt_layernorm_fwd_sum / ppt_layernorm_bwd_sum (new: split-K consumers), ppt_adamw_* (skipped == NULL: no guard).
 * 3: PPT_F16 (dtype arguments / struct fields), ppt_cross_entropy_rows (ignored labels, loss[2]), ppt_adamw_step (grad_scale),
 *    ppt_bn_rows_bwd_apply (half_dtype), ppt_*_half entry points.
 * 4: gradient scaling local to the 16-bit backward stages -- ppt_convert_scaled (new), ppt_rows_matmul_f32 (alpha),
 *    ppt_prompt_rows_bwd (scale); ppt_adamw_step (skipped counter), ppt_adamw_multi (new); ppt_cross_entropy_rows (ignore_index);
 *    ppt_mini_pointnet_conv34_half + ppt_mpn34_retile + ppt_scale_rows_convert + ppt_health_check + ppt_weights_prep (new), ppt_mini_pointnet_conv3_half (y may be
 *    NULL: statistics only). */
int ppt_abi_version(void);

/* ---- H1: farthest point sampling ---------------------------------------------------------
 * Replaces models/pointbert/misc.py:44-69 farthest_point_sample (+ the fps/index_points gather,
 * misc.py:12-42) and its twins models/pointbert/pointnet2_utils.py:95-116,
 * models/pointnet2/pointnet2_utils.py:63-84.  Bit-exact indices: distance ((dx*dx+dy*dy)+dz*dz)
 * without FMA contraction, first-maximum tie rule (torch.max), start index injected.
 *   xyz [B,N,3] f32, start [B] i64 -> out_idx [B,M] i64, out_xyz [B,M,3] f32 (may be NULL).
 * One workgroup per cloud; points and running distances live in registers, the cloud's
 * coordinates are never re-read from HBM.  N <= 16384. */
int ppt_fps_f32(const float *xyz, int B, int N, int M, const int64_t *start, int64_t *out_idx,
                float *out_xyz, void *stream);

/* ---- H2: kNN grouping -----------------------------------------------------------------------
 * Replaces models/pointbert/dvae.py:116-149 (knn_point + square_distance) and the gather /
 * centre-subtract of Group.forward, dvae.py:171-180.  Expanded-form distances with the exact
 * rounding sequence of the reference; the k nearest under the total order
 * (distance, index), emitted in that order; the [B,G,N] distance matrix is never materialised.
 *   xyz [B,N,3], center [B,G,3] -> nbr_idx [B,G,k] i64 (may be NULL),
 *   neighborhood [B,G,k,3] f32 = xyz[nbr] - center (may be NULL).   k <= 64, k <= N <= 8192. */
int ppt_knn_group_f32(const float *xyz, const float *center, int B, int N, int G, int k,
                      int64_t *nbr_idx, float *neighborhood, float *nbr_dist /* [B,G,k] expanded-form d, or NULL */,
                      void *stream);

/* square_distance (models/pointbert/dvae.py:130-149; twins pointbert/pointnet2_utils.py:51-72, pointnet2/pointnet2_utils.py:
 * 19-40): out[b, s, n] = ((-2 * <src_s, dst_n>) + |src_s|^2) + |dst_n|^2 with the reference CPU path's rounding sequence
 * (dot = fma(z, z', fma(y, y', x * x')), |a|^2 = (x^2 + y^2) + z^2): bit-identical to torch on the
 * CPU for config-sized shapes.  The product path never materialises this matrix (ppt_knn_group_f32 fuses it); the entry
 * point serves direct callers of the surface function.  src [B,S,3], dst [B,N,3], out [B,S,N] f32. */
int ppt_square_distance_f32(const float *src, const float *dst, int B, int S, int N, float *out, void *stream);

/* ---- H7: ball query ---------------------------------------------------------------------------
 * Replaces models/pointnet2/pointnet2_utils.py:87-107 query_ball_point: first K indices in
 * ascending order with d <= r^2 (expanded-form distance), padded with the first hit. */
int ppt_ball_query_f32(const float *xyz, const float *center, int B, int N, int S, float radius_sq,
                       int K, int64_t *idx, float *grouped_xyz /* [B,S,K,3] = xyz[idx] - center, or NULL */,
                       void *stream);

/* Up to three ball queries around the SAME centres in one pass (PointNetSetAbstractionMsg, pointnet2_utils.py:228-266: one
 * (radius, nsample) pair per scale): cloud staged once, every distance evaluated once.  Results identical to n calls of
 * ppt_ball_query_f32.  idx[j] [B,S,K[j]] i64, gxyz[j] [B,S,K[j],3] f32 or NULL. */
typedef struct ppt_ball_multi {
    int n;                           /* 1..3 queries */
    float r2[3];                     /* squared radii */
    int K[3];
    int64_t *idx[3];
    float *gxyz[3];
} ppt_ball_multi;
int ppt_ball_query_multi_f32(const float *xyz, const float *center, int B, int N, int S, const ppt_ball_multi *q, void *stream);

/* ---- GEMM with fused prologue / epilogue ----------------------------------------------------
 * C[M,N] = epilogue( prologue(A)[M,K] . B[N,K]^T ).  A and B are K-contiguous (torch Linear /
 * k=1 Conv1d weight layout [out,in]).  Replaces every nn.Linear / Conv1d(k=1) on the path:
 * models/pointbert/dvae.py:188-199, point_encoder.py:18-20,41-43,133,138-142,
 * models/ULIP_models.py:38-46 (in_proj/out_proj/c_fc/c_proj), :222, :257, :281.
 * bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate; f32: v_mfma_f32_32x32x2_f32.
 */
typedef struct ppt_gemm_params {
    /* operands */
    const void *A; int64_t lda;      /* [M,K] dtype (ignored when a_mode == PPT_A_CONV1) */
    const void *B; int64_t ldb;      /* [N,K] dtype */
    void *C; int64_t ldc;            /* [M,N] c_dtype (may be NULL when only pooled output wanted) */
    int M, N, K;
    int dtype;                       /* operand dtype of A and B */
    int c_dtype;                     /* dtype of C */
    /* A prologue */
    int a_mode;                      /* PPT_A_PLAIN / PPT_A_AFFINE_RELU / PPT_A_CONV1 */
    const float *a_scale;            /* [K]  AFFINE_RELU: a' = relu(a*scale[k]+shift[k]) (BatchNorm+ReLU) */
    const float *a_shift;            /* [K] */
    const float *pts;                /* CONV1: points [M,3] f32; a'[m][c] = relu(scale[c]*(w1[c].p_m + b1[c]) + shift[c]) */
    const float *w1;                 /* CONV1: [K,3] f32 */
    const float *b1;                 /* CONV1: [K] f32 */
    /* epilogue, applied in this order */
    const float *bias;               /* [N] f32 or NULL */
    const float *group_add;          /* [M/group_rows, N] f32 or NULL: + group_add[m/group_rows][n] */
    int group_rows;
    int act;                         /* PPT_ACT_*: v = act(v); or, when dact_pre != NULL, v *= act'(dact_pre) instead */
    const void *dact_pre;            /* [M,N] operand dtype or NULL: saved pre-activation (backward through act) */
    int64_t ld_dact;
    const float *row_scale;          /* [M/row_scale_rows] f32 or NULL (DropPath factor per sample) */
    int row_scale_rows;
    const float *residual;           /* [M,N] f32 or NULL: + residual (ld = ld_res) */
    int64_t ld_res;
    const float *residual2;          /* [M,N] f32 or NULL: + residual2 (positional embedding re-add) */
    int64_t ld_res2;
    /* side outputs */
    void *C2; int64_t ldc2; int c2_dtype;  /* optional second output in another dtype: the final value, or ... */
    int c2_pre;                      /* ... when != 0 the PRE-activation (after bias/group_add): saved for the backward */
    float *col_sum;                  /* [ceil(M/32), N] per 32-row chunk: column sums of the pre-activation... */
    float *col_sqsum;                /* ... and M2 = sum (v - chunk mean)^2 (BatchNorm batch statistics), or NULL */
    void *pool_max;                  /* [M/pool_rows, N] pool_dtype: max over each group of pool_rows consecutive rows */
    int pool_dtype;
    int pool_rows;                   /* 16, 32 or 64 (0 = 32): one kNN / ball-query group per pool_rows rows */
    void *pool_min;                  /* optional [M/pool_rows, N] minimum (BatchNorm with a negative scale flips the max) */
    /* batching (blockIdx.z): pointer offsets in ELEMENTS per batch */
    int batch; int64_t strideA, strideB, strideC;
    int wave_prio;                   /* != 0: the kernel raises its waves' issue priority (0: what ppt_set_wave_priority set) */
    /* split16 (ABI 6; dtype must be PPT_F32): != 0 multiplies the fp32 operands as hi + lo IEEE-half pairs on the 16-bit matrix
     * pipe -- A.B = A_hi.B_hi + A_hi.B_lo + A_lo.B_hi, fp32 accumulation -- instead of the fp32 MFMA: ~22 significand bits per
     * operand (an absolute floor of 2^-25 after scaling, inf above 65 504) at ~3-5x the fp32 MFMA's rate.  Operand values are
     * multiplied by 2^split_a_pow2 / 2^split_b_pow2 before the split (|.| <= 24) and the product by the inverse before the
     * epilogue, so the caller places each operand's magnitudes in half's range; everything else (A prologues, epilogues,
     * outputs) is the fp32 path's.  Large plain problems run on 256 x 128 tiles (csrc/gemm256.hip: gemm256s_kernel), the rest on
     * the register-staged 128 x 128 / 64 x 64 loops -- the same bits either way; split16 == 2 keeps a launch on the loops (A/B). */
    int split16, split_a_pow2, split_b_pow2;
    /* ABI 7: a FINITE operand value whose scaled magnitude exceeds 65 504 is saturated to +-65 504 before the split (the product
     * stays finite where the fp32 MFMA's would be; inf / NaN inputs propagate unchanged) and every wave that saturated anything
     * adds 1 to *split_overflow (device memory, may be NULL: saturation without a report) -- ppt_amd/health.py polls it. */
    unsigned int *split_overflow;
} ppt_gemm_params;

#define PPT_A_PLAIN 0
#define PPT_A_AFFINE_RELU 1
#define PPT_A_CONV1 2

#define PPT_ACT_NONE 0
#define PPT_ACT_RELU 1
#define PPT_ACT_GELU 2       /* exact erf GELU, point_encoder.py:17 */
#define PPT_ACT_QUICKGELU 3  /* x*sigmoid(1.702x), ULIP_models.py:30-32 */

int ppt_gemm(const ppt_gemm_params *p, void *stream);
/* The 256-row macro-tile core of ppt_gemm, called explicitly (csrc/gemm256.hip; ABI 5): 16-bit operands, a_mode PLAIN, K % 32 == 0,
 * K >= 64, no pooling; 256 x 256 output tiles per 512-thread workgroup (256 x 128 for narrow N), both operands through a ring of
 * LDS-DMA stages, the same epilogues.  ppt_gemm routes the large plain problems (every nn.Linear of the frozen ViT blocks over the
 * 16 k-32 k token rows of a batch: point_encoder.py:14-30,46-58) here by itself; PPT_GEMM256=0 in the environment turns that off.
 * PPT_EUNSUPPORTED when the problem is outside these limits (nothing launched). */
int ppt_gemm256(const ppt_gemm_params *p, void *stream);
/* ppt_gemm's automatic use of that core for this host thread's following launches: 0 off, 1 on, -1 (default) what the
 * environment says (PPT_GEMM256, default on).  For same-process A/B timing. */
void ppt_set_gemm256(int mode);
int ppt_get_gemm256(void);

/* Wave (issue) priority of the launches this host thread makes from now on: != 0 raises it (s_setprio 3) in the kernels of the
 * prompt chain -- ppt_gemm's 64x64 tile loop, LayerNorm forward / backward, the causal / short attention kernels, the head,
 * AdamW and prompt-row kernels.  For callers whose critical path is the text side (only the prompt trains); default 0. */
void ppt_set_wave_priority(int prio);
/* Percent (10..100, default 100) of the CUs the persistent point-tower kernels of this host thread's following launches take
 * (ppt_mini_pointnet_conv12 / conv3 / conv4_bf16: one long-lived workgroup per CU, HBM-bound -- alone they are as fast on 60 %
 * of the CUs' wave slots as on all).  A workgroup that lives for the whole kernel frees its CU only when the kernel ends, so
 * while such a kernel covers the chip, the small kernels of the prompt chain on the other stream wait: when that chain is the
 * step's critical path (only the prompt trains), leaving it CUs shortens the step (C2: 3.67 -> 3.50 ms at 60 %). */
void ppt_set_persistent_occupancy(int percent);
int ppt_get_persistent_occupancy(void);
int ppt_get_wave_priority(void);

/* ---- short-K linears with the weight stationary in registers (csrc/rowgemm.hip) --------------------------------
 * Replaces, for the frozen transformer blocks in the bf16 mode, nn.LayerNorm + nn.Linear pairs and the residual-form
 * linears: PointBERT norm1 -> attn.qkv, norm2 -> mlp.fc1 + GELU, attn.proj + DropPath + residual
 * (point_encoder.py:46-58,65-79); CLIP ln_1 -> in_proj, ln_2 -> c_fc + QuickGELU, out_proj + residual
 * (ULIP_models.py:35-56).   C[M,N] = epilogue( prologue(A)[M,K] . W[N,K]^T ), everything contiguous row-major:
 *   a_ln == 0: A is bf16 [M,K];  a_ln != 0: A is the f32 residual stream [M,K] and the operand is
 *              bf16( (A - mean) * rstd * ln_w + ln_b ) per row (two-pass mean / biased variance, eps = ln_eps);
 *   residual_form == 0: C (bf16) = act(acc + bias); C2 (bf16, optional) = acc + bias (the saved pre-activation);
 *   residual_form != 0: C (f32)  = residual + row_scale[m / row_scale_rows] * (acc + bias) + residual2; C may alias
 *              residual; row_scale / residual2 / bias may be NULL.
 * K must be 384 or 512, N % 8 == 0 (N % 4 == 0 in the residual form), pointers 16-byte aligned, a_ln and residual_form not both set: anything else
 * PPT_EUNSUPPORTED / PPT_EINVAL.
 * walkers: workgroups per column group walking the 32-row tiles (0 = as many as fill the chip). */
typedef struct ppt_rowgemm_params {
    const void *A;
    const void *W;                   /* [N,K] bf16 */
    void *C;
    void *C2;
    int M, N, K;
    int a_ln;
    const float *ln_w;               /* [K] */
    const float *ln_b;               /* [K] */
    float ln_eps;
    float *ln_mean;                  /* [M] f32 or NULL: the rows' mean ... */
    float *ln_rstd;                  /* ... and 1 / sqrt(var + eps), saved for ppt_layernorm_bwd (both or neither) */
    const float *bias;               /* [N] or NULL */
    int act;                         /* PPT_ACT_NONE / PPT_ACT_GELU / PPT_ACT_QUICKGELU (residual_form == 0 only) */
    int residual_form;
    const float *residual;           /* [M,N] f32 */
    const float *residual2;          /* [M,N] f32 or NULL */
    const float *row_scale;          /* [ceil(M / row_scale_rows)] f32 or NULL */
    int row_scale_rows;
    int walkers;
    int groups;                      /* set by the library (column groups of the launch); callers leave it 0 */
    int dtype;                       /* PPT_BF16 (also 0) or PPT_F16: the 16-bit format of A (or of the LayerNorm output), W, C, C2 */
} ppt_rowgemm_params;

int ppt_rowgemm_bf16(const ppt_rowgemm_params *p, void *stream);

/* ---- the MLP half of a frozen PointBERT block as one kernel (csrc/mlp_fused.hip) ---------------------------------------
 * Replaces nn.LayerNorm + Mlp (fc1, GELU, fc2) + DropPath + the residual add of Block.forward (point_encoder.py:14-30, 69,
 * 78-79) -- and the next block's "+ pos" (:103) -- for the bf16 mode:
 *   out[m, :] = x[m, :] + row_scale[m / row_scale_rows] * ( GELU( LN(x[m, :]) W1^T + b1 ) W2^T + b2 ) + residual2[m, :]
 * x, out [M, 384] f32 (out may be x); W1 / W2: the bf16 weights ([1536, 384] / [384, 1536] row-major) RE-TILED once by
 * ppt_vit_mlp_retile into the order the kernel's waves consume them ([slab 12][wave 8][fragment 12][lane 64][8 bf16]: every
 * load of a wave is 1 KB of consecutive bytes; same sizes as the originals); ln_w / ln_b [384], b1 [1536], b2 [384] f32
 * (biases may be NULL), row_scale / residual2 may be NULL.  The [M, 1536] hidden tensor and the LayerNorm output never
 * exist in memory.  D must be 384 and hidden 1536 (PPT_EUNSUPPORTED otherwise).  workgroups: 0 = one per CU.
 * n_chunks / rows_per_chunk are set by the library. */
typedef struct ppt_vit_mlp_params {
    const float *x;
    float *out;
    const void *W1;
    const void *W2;
    const float *ln_w;
    const float *ln_b;
    float ln_eps;
    const float *b1;
    const float *b2;
    const float *row_scale;
    int row_scale_rows;
    const float *residual2;
    int M, D, hidden;
    int workgroups;
    int n_chunks, rows_per_chunk;
    int dtype;                       /* PPT_BF16 (also 0) or PPT_F16: the 16-bit format of W1 / W2 and of the in-kernel operands */
    /* optional (ABI 3): the attention branch's tail in front -- x_mid = x + proj_row_scale[m / rows] * (proj_a Wp^T + proj_b)
     * (point_encoder.py:57-58,77) is formed first, written to `out`, and is what the MLP branch then reads.  proj_a [M, D] in
     * `dtype`; proj_W = attn.proj.weight [D, D] through ppt_vit_proj_retile.  proj_a == NULL: plain form. */
    const void *proj_a;
    const void *proj_W;
    const float *proj_b;
    const float *proj_row_scale;
    int proj_row_scale_rows;
} ppt_vit_mlp_params;

int ppt_vit_mlp_retile(const void *W1, const void *W2, void *W1_tiled, void *W2_tiled, void *stream);
int ppt_vit_proj_retile(const void *Wp, void *Wp_tiled, void *stream);
int ppt_vit_mlp_bf16(const ppt_vit_mlp_params *p, void *stream);

/* ABI 7 -- the same contract on a different schedule (csrc/mlp_fused3.hip): 256-unit hidden slabs (every LDS fragment of fc1
 * feeds two MFMAs), the GELU of slab j + 1 issued between the MFMAs of fc2 on slab j, the fc2 accumulators STARTING at the
 * residual (x, or x_mid of the proj prologue: no x_mid round trip through memory, no residual read in the epilogue; DropPath's
 * factor rides on the GELU output), weights through two register rings.  W1 / W2 must come from ppt_vit_mlp3_retile
 * ([slab 6][wave 8][k-step 12][half 2][lane 64][8] / [slab 6][wave 8][k-step 8][column block 3][lane 64][8]); proj_W from
 * ppt_vit_proj_retile as before.  With proj_a set, `out` receives only the FINAL rows (x_mid is never written). */
int ppt_vit_mlp3_retile(const void *W1, const void *W2, void *W1_tiled, void *W2_tiled, void *stream);
int ppt_vit_mlp3_bf16(const ppt_vit_mlp_params *p, void *stream);

/* ---- LayerNorm --------------------------------------------------------------------------------
 * Replaces nn.LayerNorm at point_encoder.py:65,69,152 and ULIP_models.py:21-27,39,46,176.
 * fwd: xs = x (+ add); y = LN(xs)*w + b.  x, add, xs f32 [M,D] (xs may alias x, may be NULL);
 *      y in `y_dtype`; mean/rstd [M] f32 saved for the backward (may be NULL).
 *      add_rows > 0: `add` has add_rows rows and row m uses add[m % add_rows] (positional table).
 * bwd: dx (+)= LN'(dy) (f32; accumulate_dx adds into the residual-stream gradient); dx_copy (may be
 *      NULL) receives the final dx in `dx_copy_dtype` (the next GEMM's operand); optional dw/db
 *      partials [partial_rows, D] reduced by the caller (ppt_reduce_rows). */
int ppt_layernorm_fwd(const float *x, const float *add, int add_rows, float *xs, const float *w,
                      const float *b, void *y, int y_dtype, float *mean, float *rstd, int M, int D,
                      float eps, void *stream);
int ppt_layernorm_bwd(const float *dy, const float *xs, const float *w, const float *mean,
                      const float *rstd, float *dx, int accumulate_dx, void *dx_copy, int dx_copy_dtype,
                      float *dw_partial, float *db_partial, int partial_rows, int M, int D, void *stream);
/* The consumer side of a SPLIT-K linear (ABI 5; the prompt chain's K = 1536 / 2048 linears over 817 rows, ULIP_models.py:35-67:
 * 104 workgroups walking K serially at one CU's L2 -> LDS rate): the S partial products are written by ppt_gemm as plain fp32
 * slices parts[S][M][D] (batch = S, no epilogue) and the LayerNorm that reads the result adds them up in slice order -- no reduction
 * launch, no atomics, fixed summation order.
 * fwd_sum: row = x + bias + parts[0] + ... + parts[S-1] (bias [D] or NULL); xs receives the row, y = LN(row); D % 8 == 0, D <= 512.
 * bwd_sum: ppt_layernorm_bwd's input-gradient form (no dw / db) with dy = parts[0] + ... + parts[S-1]. */
int ppt_layernorm_fwd_sum(const float *x, const float *bias, const float *parts, int S, float *xs, const float *w, const float *b,
                          void *y, int y_dtype, float *mean, float *rstd, int M, int D, float eps, void *stream);
int ppt_layernorm_bwd_sum(const float *dy_parts, int S, const float *xs, const float *w, const float *mean, const float *rstd,
                          float *dx, int accumulate_dx, void *dx_copy, int dx_copy_dtype, int M, int D, void *stream);

/* ---- LayerNorm + a K = 384 linear with the rows stationary and the weight streamed (ABI 7; csrc/lnlin.hip) -------------------
 * C[M, N] (dtype) = LayerNorm(x[M, 384]; ln_w, ln_b, ln_eps) W[N, 384]^T (+ bias): norm1 + qkv of a frozen PointBERT block
 * (point_encoder.py:46-55, 76).  x f32 (the residual stream, "+ pos" already in it), N a multiple of 384, W the 16-bit weight RE-TILED
 * once by ppt_lnlin_retile ([slice N/384][wave 8][k-step 12][column block 3][lane 64][8]).  A workgroup = 64 rows x one 384-column
 * slice, two per CU.  `slices` is set by the library. */
typedef struct ppt_lnlin_params {
    const float *x; const void *W; void *C;
    const float *ln_w; const float *ln_b; float ln_eps;
    const float *bias;                       /* [N] or NULL */
    int M, N, K;
    int dtype;                               /* PPT_BF16 | PPT_F16: W, C and the in-kernel LayerNorm output */
    int slices;
} ppt_lnlin_params;
int ppt_lnlin_retile(const void *W, void *W_tiled, int N, void *stream);
int ppt_lnlin(const ppt_lnlin_params *p, void *stream);

/* ---- the MLP half of a CLIP text-tower layer as one launch per direction (ABI 7; csrc/text_mlp.hip) -------------------------
 * Replaces the two Linears + QuickGELU of ResidualAttentionBlock.mlp (ULIP_models.py:41-42, 49-51) on the prompt chain:
 *   mode 0 (forward):  parts[s] = QuickGELU( A W1[slice s]^T + b1[slice s] ) W2[:, slice s]^T,  pre (optional) = A W1^T + b1
 *   mode 1 (backward): parts[s] = ( (A W1[slice s]^T) * QuickGELU'(pre[:, slice s]) ) W2[:, slice s]^T
 * for the eight 256-unit slices of the 2048-wide hidden dimension: A [M, 512] (lda elements per row; fp32 with the LayerNorm prologue
 * below) and pre [M, 2048] in `dtype`
 * (PPT_BF16 | PPT_F16), parts [8, M, 512] f32 -- the caller's LayerNorm adds the slices up in order (ppt_layernorm_fwd_sum /
 * ppt_layernorm_bwd_sum).  W1 [2048, 512] / W2 [512, 2048] row-major 16-bit (forward: c_fc.weight / c_proj.weight; backward:
 * c_proj.weight^T / c_fc.weight^T) RE-TILED once by ppt_text_mlp_retile.  The hidden activation never exists in memory.
 * D must be 512 and hidden 2048 (PPT_EUNSUPPORTED otherwise). */
typedef struct ppt_text_mlp_params {
    const void *A; int64_t lda;
    const void *W1; const void *W2;          /* fragment-ordered copies (ppt_text_mlp_retile) */
    const float *b1;                         /* [2048] or NULL (mode 0 only) */
    void *pre;                               /* [M, 2048] dtype: written in mode 0 (may be NULL), read in mode 1 */
    float *parts;                            /* [8, M, 512] */
    int M, D, hidden;
    int mode;                                /* 0 forward, 1 backward */
    int dtype;
    int wave_prio;                           /* != 0: raised issue priority (0: what ppt_set_wave_priority set) */
    /* optional LayerNorm prologue (mode 0 only): ln_w != NULL -> A is the FP32 residual stream [M, 512] (lda floats per row) and
     * LayerNorm(A; ln_w, ln_b, ln_eps) -- ln_2 of the layer, computed in fp32 as ULIP_models.py:21-27 does -- is applied while the rows
     * are staged; ln_mean / ln_rstd [M] (optional) receive the row statistics for the LayerNorm backward. */
    const float *ln_w; const float *ln_b; float ln_eps; float *ln_mean; float *ln_rstd;
    /* dtype == PPT_F32: the split16 form (csrc/text_mlp_split.hip) -- A and `pre` are fp32, W1 / W2 the hi + lo half copies of
     * ppt_text_mlp_retile_split (made with the same split_b_pow2), every product three MFMAs on hi + lo IEEE-half pairs
     * (ppt_gemm_params.split16); A and the hidden activation are multiplied by 2^split_a_pow2 before they are split; a wave that
     * saturated a finite value beyond half's range adds 1 to *split_overflow (may be NULL).  The LayerNorm prologue as above. */
    int split_a_pow2, split_b_pow2; unsigned int *split_overflow;
} ppt_text_mlp_params;
int ppt_text_mlp_retile(const void *W1, const void *W2, void *W1_tiled, void *W2_tiled, void *stream);
/* fp32 W1 [2048, 512] / W2 [512, 2048] -> the fragment-ordered hi + lo half copies (4 MB each) of the split16 form */
int ppt_text_mlp_retile_split(const float *W1, const float *W2, void *W1_tiled, void *W2_tiled, int b_pow2, void *stream);
int ppt_text_mlp_pair(const ppt_text_mlp_params *p, void *stream);

/* ---- one linear of the text tower's attention half on split16 products, rows stationary (ABI 7; csrc/text_lin_split.hip) -------
 * C[M, N] = A[M, K] W[N, K]^T (+ bias) (+ residual), every product three MFMAs on hi + lo IEEE-half pairs of the fp32 operands
 * (ppt_gemm_params.split16): nn.MultiheadAttention's in_proj / out_proj of a CLIP ResidualAttentionBlock (ULIP_models.py:38, 45-51)
 * and their input-gradient products.  K a multiple of 512, N of 256.  K > 512: the K chunks' partial products leave separately,
 * C = parts[K / 512][M][N] (ldc == N, no bias / residual) for a consumer that adds them up (ppt_layernorm_bwd_sum).
 * W: the fragment-ordered hi + lo half copy of ppt_text_lin_retile_split (N * K * 4 bytes, made with the same split_b_pow2). */
typedef struct ppt_text_lin_params {
    const void *A; int64_t lda;              /* fp32 [M, K] */
    const void *W;
    const float *bias;                       /* [N] or NULL */
    const float *residual; int64_t ld_res;   /* fp32 [M, N] or NULL */
    float *C; int64_t ldc;
    int M, N, K;
    int split_a_pow2, split_b_pow2; unsigned int *split_overflow;
    int wave_prio;
} ppt_text_lin_params;
int ppt_text_lin_retile_split(const float *W, void *W_tiled, int N, int K, int b_pow2, void *stream);
int ppt_text_lin_split(const ppt_text_lin_params *p, void *stream);

/* ---- Attention ---------------------------------------------------------------------------------
 * softmax(scale * q k^T [+ causal mask]) v per (batch, head).  Replaces
 * point_encoder.py:46-55 (scale after the product, non-causal, T=513) and nn.MultiheadAttention at
 * ULIP_models.py:38,49-51 with the causal mask of :224-230 (L=77).
 * qkv [Bt, T, 3, H, hd] packed as produced by the qkv / in_proj GEMM (row stride 3*H*hd);
 * out [Bt, T, H*hd]; lse [Bt, H, T] f32 (log-sum-exp of the scaled scores, for the backward).
 * hd == 64.  16-bit (dtype PPT_BF16 | PPT_F16): flash-style MFMA kernels -- for the ViT shape (non-causal, T = 64 n + 1 <= 513,
 * Bt * H >= 128) the one that keeps K / V of a (batch, head) resident in LDS (csrc/attention_mfma.hip: attn_fwd_resident);
 * f32: VALU kernel (parity mode). */
int ppt_attention_fwd(const void *qkv, void *out, float *lse, int Bt, int T, int H, int hd,
                      float scale, int causal, int dtype, void *stream);
/* dqkv [Bt,T,3,H,hd] (dtype) from dout [Bt,T,H*hd] (dtype); `delta` [Bt,H,T] f32 workspace. */
int ppt_attention_bwd(const void *qkv, const void *out, const void *dout, const float *lse,
                      float *delta, void *dqkv, int Bt, int T, int H, int hd, float scale,
                      int causal, int dtype, void *stream);

/* Prefix-shared causal attention for the CLIP text tower under PromptLearner (ULIP_models.py:104-151, 203-230).  With the class
 * name in the "middle" / "end" position the first P positions of all C prompts are the same tokens (start token + leading
 * learnable context), and the mask is causal, so their activations are identical in every prompt at every layer: they are
 * stored -- and every LayerNorm / linear of the tower evaluated -- ONCE.  Row layout of qkv / out / dout / dqkv:
 *   rows [0, P) = positions 0..P-1 (shared); rows [P + c (T - P), P + (c + 1)(T - P)) = positions P..T-1 of prompt c.
 * A prompt's queries (positions >= P) see the shared rows as their first P keys; the gradient of the shared keys / values is
 * the sum over the prompts (+ the prefix's own causal attention): each sequence writes an fp32 partial into `workspace`
 * [C + 1, P, 2, H * hd] and a last kernel folds them in sequence order (deterministic, no atomics).
 * lse / delta: [P + C (T - P), H] f32, indexed by ROW.  0 < P < T; hd == 64; same kernels as ppt_attention_fwd / _bwd. */
size_t ppt_attention_prefix_workspace_bytes(int C, int P, int H, int hd);
/* split16 forward (ABI 6; csrc/attention_split.hip): fp32 qkv / out in the layouts of ppt_attention_fwd (P == 0) or
 * ppt_attention_prefix_fwd (P > 0, causal, C = Bt prompts), both products on the 16-bit matrix pipe from hi + lo IEEE-half pairs
 * of the fp32 operands (K.Q^T and V^T.P each as three MFMAs, fp32 accumulation; softmax statistics, rescale and output fp32):
 * the fp32 kernel's results to ~1e-6 at 4x its rate.  hd must be 64; qkv and out 16-byte aligned. */
int ppt_attention_fwd_split16(const void *qkv, void *out, float *lse, int Bt, int T, int P, int H, int hd, float scale, int causal,
                              void *stream);
/* ... and the backward: the arguments and results of ppt_attention_bwd (P == 0, workspace may be NULL) or of
 * ppt_attention_prefix_bwd (P > 0, causal, Bt = C prompts, workspace as there) with dtype PPT_F32 (delta is scratch the call
 * fills), dS = P (dP - delta) scale and the three gradient products from hi + lo half pairs.  Rows of dO and the
 * P / dS tiles are scaled by powers of two taken from their own largest element before they are split (block floating point,
 * divided out exactly): the fp32 kernels' accuracy at any gradient magnitude.  The un-frozen ViT block at T = 513, B = 64: 2.7 +
 * 1.4 ms on the fp32 VALU kernels. */
int ppt_attention_bwd_split16(const void *qkv, const void *out, const void *dout, const float *lse, float *delta, void *dqkv,
                              float *workspace, int Bt, int T, int P, int H, int hd, float scale, int causal, void *stream);
int ppt_attention_prefix_fwd(const void *qkv, void *out, float *lse, int C, int T, int P, int H, int hd, float scale, int dtype,
                             void *stream);
int ppt_attention_prefix_bwd(const void *qkv, const void *out, const void *dout, const float *lse, float *delta, void *dqkv,
                             float *workspace, int C, int T, int P, int H, int hd, float scale, int dtype, void *stream);

/* ---- mini-PointNet helpers -----------------------------------------------------------------------
 * BatchNorm1d in train mode inside dvae.py:188-199.
 * conv1_stats: per-channel (sum, M2 about the chunk mean) of y = w1.p + b1 per chunk of
 *   ppt_conv1_stats_rows_per_partial() points (K=3 layer, VALU).  The GEMM's col_sum/col_sqsum use
 *   the same (sum, M2) form with 32-row chunks.
 * bn_finalize: chunk partials (merged with the parallel-variance formula in fp64) -> scale = g/sqrt(var+eps), shift = b - mean*scale; running-stat update
 * (momentum 0.1, unbiased variance) when running_mean != NULL.  train == 0: scale/shift from the
 * running statistics (eval mode), partials ignored. */
int ppt_conv1_stats(const float *pts, int64_t M, const float *w1, const float *b1, int C,
                    float *part_sum, float *part_sqsum, int *n_partials, void *stream);
int ppt_bn_finalize(const float *part_sum, const float *part_sqsum, int n_partials,
                    int rows_per_partial, int64_t count, int C, const float *gamma, const float *beta, float eps, int train,
                    float momentum, float *running_mean, float *running_var,
                    int64_t *num_batches_tracked, float *scale, float *shift, void *stream);
/* the same with a caller-provided scratch buffer (ppt_bn_finalize_workspace_bytes(n_partials, C) bytes, 8-byte aligned;
 * 0 bytes = not needed): thousands of partial rows (conv3: 16 384 x 512) are folded by 32 x C/32 workgroups into
 * additive fp64 sums first, a second tiny kernel finishes.  Falls back to ppt_bn_finalize without a workspace. */
size_t ppt_bn_finalize_workspace_bytes(int n_partials, int C);
int ppt_bn_finalize_ws(const float *part_sum, const float *part_sqsum, int n_partials,
                       int rows_per_partial, int64_t count, int C, const float *gamma, const float *beta, float eps, int train,
                       float momentum, float *running_mean, float *running_var,
                       int64_t *num_batches_tracked, float *scale, float *shift,
                       float *mean_out /* [C] or NULL */, float *rstd_out /* [C] or NULL: what a backward needs */,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ---- BatchNorm1d (+ ReLU) over the rows of a row-major fp32 [M, C] tensor, forward statistics and backward: the
 * part-segmentation decoder's `F.relu(bn(conv(x)))` (models/pointbert/pointnet2_utils.py:362-366) with trainable
 * gamma / beta.  rows_stats: (sum, M2) partials per ppt_rows_stats_rows_per_partial() rows, ceil(M / that) x C each, for
 * ppt_bn_finalize(_ws); the normalisation itself is ppt_bn_act_rows.  Backward, with g = dy * [x*scale+shift > 0] when
 * relu: bwd_reduce writes the per-chunk partials of sum(g) and sum(g * xhat) (reduce them with ppt_reduce_rows: they are
 * d beta and d gamma); bwd_apply: dx = scale * (g - sum_g/M - xhat * sum_gx/M) (batch_stats = 0, eval mode: dx = scale*g). */
int ppt_rows_stats_rows_per_partial(void);
int ppt_rows_stats_f32(const float *x, int64_t M, int C, float *part_sum, float *part_m2, void *stream);
int ppt_bn_rows_bwd_reduce(const float *dy, const float *x, const float *scale, const float *shift, const float *mean,
                           const float *rstd, int relu, int64_t M, int C, float *part_g, float *part_gx, void *stream);
int ppt_bn_rows_bwd_apply(const float *dy, const float *x, const float *scale, const float *shift, const float *mean,
                          const float *rstd, const float *sum_g, const float *sum_gx, int relu, int batch_stats, int64_t M,
                          int C, float *dx, void *dx_half, int half_dtype, void *stream);
                          /* dx f32 and / or a 16-bit copy in half_dtype = PPT_BF16 | PPT_F16 (either pointer may be NULL) */
int ppt_conv1_stats_max_partials(int64_t M);
int ppt_conv1_stats_rows_per_partial(void);

/* ---- GroupNorm(G, C) + LeakyReLU + max over k of DGCNN_Propagation (models/pointbert/pointnet2_utils.py:371-467), on the
 * channels-last rows the 1x1 conv writes: y [B][Q][K][C] f32.  Forward: ppt_gn_stats leaves (sum, sumsq) per (cloud, chunk of
 * rows, group) in part [B][ppt_gn_stats_chunks(Q*K)][G][2] (f64; the caller folds them to mean / rstd [B][G]); ppt_gn_lrelu_max
 * writes out [B][Q][C] and the maximising neighbour arg [B][Q][C].  Backward: ppt_gn_bwd_sums leaves the two group sums of
 * the GroupNorm backward in psum [B][ppt_gn_bwd_chunks(Q)][G][2] (f64) and (dgamma, dbeta) partials in pgb [B][chunks][C][2];
 * ppt_gn_bwd_apply writes dy [B][Q][K][C] given s12n [B][G][2] = the folded sums divided by n = Q*K*C/G. */
int ppt_gn_stats_chunks(int R);
int ppt_gn_bwd_chunks(int Q);
/* (cloud, group) statistics out of ppt_gn_stats' / ppt_gn_bwd_sums' fp64 chunk partials [B, nch, G, 2], n = elements per group:
 * mode 0 -> out0 = mean [B,G], out1 = rstd [B,G] (biased variance + eps, nn.GroupNorm); mode 1 -> out0 [B,G,2] = sums / n. */
int ppt_gn_finish(const double *part, int B, int nch, int G, double n, double eps, int mode, float *out0, float *out1, void *stream);
int ppt_gn_stats(const float *y, int B, int R, int C, int G, double *part, void *stream);
int ppt_gn_lrelu_max(const float *y, const float *mean, const float *rstd, const float *gamma, const float *beta, int B, int Q, int K,
                     int C, int G, float slope, float *out, int32_t *arg, void *stream);
int ppt_gn_bwd_sums(const float *y, const float *dout, const float *out, const int32_t *arg, const float *mean, const float *rstd,
                    const float *gamma, int B, int Q, int K, int C, int G, float slope, double *psum, float *pgb, void *stream);
int ppt_gn_bwd_apply(const float *y, const float *dout, const float *out, const int32_t *arg, const float *mean, const float *rstd,
                     const float *gamma, const float *s12n, int B, int Q, int K, int C, int G, float slope, float *dy, void *stream);

/* ---- first half of the PointBERT mini-PointNet in one kernel (Encoder.first_conv + the group max, dvae.py:188-193,206-210):
 * y2[m, :] = W2 . relu(a_scale * (w1 . pts[m] + b1) + a_shift) + bias2 (bf16, [M,N]) and gmax[g, :] = max over the 32 rows of
 * group g (bf16, [M/32, N]).  Same arithmetic as ppt_gemm with PPT_A_CONV1 + bias + pool_max over 32 rows (bit-identical
 * results), without its tile staging.  C1 = 128, N = 256, M % 32 == 0, W2 [N, C1] bf16; anything else: PPT_EUNSUPPORTED. */
int ppt_mini_pointnet_conv12_bf16(const float *pts, int64_t M, const float *w1, const float *b1, const float *a_scale,
                                  const float *a_shift, int C1, const void *W2, const float *bias2, int N, void *y2, void *gmax,
                                  void *stream);

/* The middle conv (Encoder.second_conv[0] on cat(global, local), dvae.py:194-195, 211-212) in split form:
 * y[m, :] = W . A[m, :] + gterm[m / 32, :], A [M,256] bf16 contiguous, W [512,256] bf16 (the local half of the weight), gterm
 * [M/32, 512] f32 (the global half applied to the group maxima, bias included), y [M,512] bf16; part_sum / part_m2 [M/32, 512]
 * (both or neither) receive the BatchNorm partials per 32-row chunk.  Same results as ppt_gemm with group_add + col_sum.
 * K = 256, N = 512, M % 32 == 0; anything else: PPT_EUNSUPPORTED. */
int ppt_mini_pointnet_conv3_bf16(const void *A, int64_t M, int K, const void *W, const float *gterm, int N, void *y, float *part_sum,
                                 float *part_m2, void *stream);

/* The second half's last conv with its group max (Encoder.second_conv[1:] + max, dvae.py:194-199, 213-214):
 * tok[g, :] = max over the 32 rows of group g of W . relu(a_scale * A + a_shift) + bias, A [M,512] bf16 contiguous, W [256,512]
 * bf16, tok [M/32, 256] bf16.  Same results as ppt_gemm with PPT_A_AFFINE_RELU + bias + pool_max over 32 rows, with W held in
 * registers and the group's rows read from HBM once.  K = 512, N = 256, M % 32 == 0; anything else: PPT_EUNSUPPORTED. */
int ppt_mini_pointnet_conv4_bf16(const void *A, int64_t M, int K, const float *a_scale, const float *a_shift, const void *W,
                                 const float *bias, int N, void *tok, void *stream);

/* The three kernels above for either 16-bit operand format: dtype = PPT_BF16 or PPT_F16 (W, the bf16-typed activations and
 * outputs are then all in that format). */
int ppt_mini_pointnet_conv12_half(const float *pts, int64_t M, const float *w1, const float *b1, const float *a_scale,
                                  const float *a_shift, int C1, const void *W2, const float *bias2, int N, void *y2, void *gmax,
                                  int dtype, void *stream);
/* ---- conv3 + BatchNorm + ReLU + conv4 + max as ONE kernel (csrc/mpn34.hip; SURVEY 8(f) N1, dvae.py:194-199, 211-214) ---------
 * tok[g, :] = max over the 32 rows m of group g of  W4 . relu(W3s . y2[m, :] + gs[g, :]) + bias4.   y2 [M,256] 16-bit; W3s [512,256]
 * 16-bit = the local half of the conv3 weight with the folded BatchNorm scale multiplied in (scale[n] * W3[n, 256:512]); gs
 * [M/32, 512] f32 = scale * (global half of conv3 per group, bias included) + shift; W4_tiled = ppt_mpn34_retile(W4 [256,512]);
 * tok [M/32, 256] 16-bit.  M % 32 == 0.  The [M,512] intermediate never reaches HBM.
 * ppt_mini_pointnet_conv3_half with y == NULL (part_sum / part_m2 given) is the statistics pass of the training step in front of it. */
int ppt_mpn34_retile(const void *W4, void *W4_tiled, void *stream);
int ppt_mini_pointnet_conv34_half(const void *y2, int64_t M, const void *W3s, const float *gs, const void *W4_tiled,
                                  const float *bias4, void *tok, int dtype, void *stream);
int ppt_mini_pointnet_conv3_half(const void *A, int64_t M, int K, const void *W, const float *gterm, int N, void *y, float *part_sum,
                                 float *part_m2, int dtype, void *stream);
int ppt_mini_pointnet_conv4_half(const void *A, int64_t M, int K, const float *a_scale, const float *a_shift, const void *W,
                                 const float *bias, int N, void *tok, int dtype, void *stream);

/* The same product with BatchNorm partials of its output instead of the group max -- the first two convs of a PointNet2
 * set-abstraction branch that sees raw coordinates (models/pointnet2/pointnet2_utils.py:168-199, 217-262): y2 [M,N] bf16,
 * part_sum / part_m2 [M/32, N] f32 = per 32-row chunk (sum, sum (v - chunk mean)^2), what ppt_bn_finalize_ws takes with
 * rows_per_partial = 32.  (C1, N) in {(32,32), (64,64), (64,96), (64,128), (128,128)}, M % 32 == 0; bias2 may be NULL. */
int ppt_conv12_stats_bf16(const float *pts, int64_t M, const float *w1, const float *b1, const float *a_scale, const float *a_shift,
                          int C1, const void *W2, const float *bias2, int N, void *y2, float *part_sum, float *part_m2, void *stream);

/* ---- last conv of a PointNet2 set-abstraction branch (models/pointnet2/pointnet2_utils.py:186-206, 236-266) without tile
 * staging: v = relu(a_scale * A + a_shift) @ W^T + bias over A [M,K] bf16 (row stride lda), W [N,K] bf16; v is not written:
 * part_sum / part_m2 [M/32, N] = BatchNorm partials per 32-row chunk, pmax / pmin [M/pool_rows, N] f32 = max / min of v
 * over every pool_rows consecutive rows (ppt_pool_finish folds and finishes).  Same results as ppt_gemm with
 * PPT_A_AFFINE_RELU + pool_max/pool_min + col_sum.  (K, N, pool_rows) in {(32,64,16), (64,128,32), (96,128,64), (128,256,64)};
 * anything else: PPT_EUNSUPPORTED. */
int ppt_affine_conv_pool_bf16(const void *A, int64_t lda, int64_t M, int K, const float *a_scale, const float *a_shift, const void *W,
                              const float *bias, int N, int pool_rows, float *pmax, float *pmin, float *part_sum, float *part_m2,
                              void *stream);

/* ---- PointMLP (models/pointmlp/pointMLP.py) pieces outside the GEMM / BatchNorm / gather kernels above.
 * ppt_group_anchor_stats: LocalGrouper normalize="anchor" (:170-175): out[(b*S+s)*2 + {0,1}] = sum, sum of squares over
 *   j < K, c < D of x[b*Nsrc + idx[b,s,j], c] - x[b*Nsrc + anchor[b,s], c]; x [B*Nsrc, D] f32 or bf16.
 * ppt_bn_res_act_rows: the tail of ConvBNReLURes1D (:220-221) y = relu(scale*x + shift + res) over rows [M, C]; pool > 1
 *   writes the max over every `pool` consecutive rows instead ([M/pool, C]: adaptive_max_pool1d of :251 and :332).
 *   res_scale/res_shift (both or neither): res is a raw conv output, relu(res_scale*res + res_shift) is the block input. */
int ppt_group_anchor_stats(const void *x, int x_dtype, const int64_t *idx, const int64_t *anchor, int B, int Nsrc, int S, int K,
                           int D, float *out, void *stream);
/* PointMLP LocalGrouper, the rest of the "anchor" normalisation in two launches (ABI 6; pointMLP.py:170-175 and the transfer conv
 * taken by linearity, engine.pointmlp_forward):
 *   ppt_pointmlp_cloud_rstd: stats [B, S, 2] f32 (ppt_group_anchor_stats) -> r [B] = 1 / (unbiased std over n values + 1e-5), fp64 inside;
 *   ppt_pointmlp_pq: PQ [B*N, 2C] f32, r [B], cidx [B, S] i64 (FPS indices), c0 [C] ->
 *       P [B*N, C] = PQ[:, :C] * r[b];  Q [B*S, C] = (c0 + PQ[b*N + cidx, C:]) - P[b*N + cidx].  C % 4 == 0. */
int ppt_pointmlp_cloud_rstd(const float *stats, int B, int S, double n, float *r, void *stream);
int ppt_pointmlp_pq(const float *PQ, const float *r, const int64_t *cidx, const float *c0, float *P, float *Q, int B, int N, int S,
                    int C, void *stream);
int ppt_bn_res_act_rows(const void *x, int x_dtype, const void *res, int res_dtype, int64_t M, int C, int pool, const float *scale,
                        const float *shift, const float *res_scale, const float *res_shift, void *y, int y_dtype, void *stream);

/* ---- weight-gradient GEMM on the operands as stored: part[z][N1][N2] (fp32) = A[z-th M-slice, N1]^T @ B[z-th M-slice, N2],
 * A [M,N1], B [M,N2] bf16 row-major (dW = dY^T X of nn.Linear / Conv1d(k=1): torch.autograd does this product with its
 * own transposes).  n_slices cuts M so that the few output tiles fill the chip; fold the slices with ppt_reduce_rows.
 * Needs M % (32 * n_slices) == 0 and N1, N2, lda, ldb multiples of 8. */
int ppt_gemm_tn_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t M, int N1, int N2, int n_slices, float *part,
                     void *stream);
/* the same kernel for operands of either 16-bit format: dtype = PPT_BF16 or PPT_F16 */
int ppt_gemm_tn_half(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t M, int N1, int N2, int n_slices, float *part,
                     int dtype, void *stream);

/* nn.CrossEntropyLoss(label_smoothing, reduction='mean') over R rows of C <= 96 classes and its gradient (main_partseg.py:213;
 * main_cls.py:52,196).  A row whose label == ignore_index (outside [0, C); torch's default -100) is an ignored row, as in ATen:
 * no loss, zero gradient, not counted in the mean.  Any OTHER label outside [0, C) is a corrupt label (ATen: device assert): the
 * loss comes back NaN so that the caller's non-finite check fires (its gradient row is zero).
 * loss: TWO floats -- loss[0] = the mean over the counted rows, loss[1] = R / counted rows (exactly 1 when
 * none is ignored); dlogits [R,C] = d loss / d logits scaled by 1 / R: multiply by loss[1] for the exact gradient.
 * partial: scratch of 2 * ceil(R / 128) floats.  Fixed summation order. */
int ppt_cross_entropy_rows(const float *logits, const int64_t *labels, float smoothing, int64_t R, int C, int64_t ignore_index,
                           float *loss, float *dlogits, float *partial, void *stream);

/* out[M,N] = alpha * A[M,K] . W[K,N], fp32, few rows (K <= 1536, K % 32 == 0): the EOT projection `x @ self.text_projection`
 * (ULIP_models.py:222) and its backward.  W row-major as stored ([K,N]).  alpha > 0: 1, or the power-of-two gradient scale with
 * which the text tower's backward enters its 16-bit stages (the product by a power of two is exact). */
int ppt_rows_matmul_f32(const float *A, const float *W, int M, int K, int N, float alpha, float *out, void *stream);

/* ---- the step between the towers when only the prompt trains (head_type 0) --------------------------
 * head_logits: spc[B,E] = exp(logit_scale) * feat[B,F] @ w[F,E] (w = pc_projection as stored, ULIP_models.py:257), logits[B,C] =
 *   spc @ (text / |text|)^T with text[C,E] the un-normalised text features (:279-281).
 * head_ce_bwd: loss = CrossEntropyLoss(label_smoothing) with mean reduction (main_cls.py:52,196) of logits / labels[B] (int64)
 *   and d_text[C,E] = d loss / d text (through the L2 normalisation).  fp32, fixed summation order. */
int ppt_head_logits(const float *feat, const float *w, const float *text, const float *logit_scale, int B, int F, int E, int C,
                    float *spc, float *logits, void *stream);
int ppt_head_ce_bwd(const float *logits, const int64_t *labels, const float *spc, const float *text, float smoothing, int B,
                    int E, int C, float *loss, float *d_text, void *stream);

/* ---- small fused ops ---------------------------------------------------------------------------
 * pos_embed first layer + GELU (point_encoder.py:138-140): y[m][c] = gelu(w[c].p_m + b[c]), K=3. */
int ppt_linear3_gelu(const float *pts, int64_t M, const float *w, const float *b, int C, void *y,
                     int y_dtype, void *stream);
/* point_encoder.py:251: out[b] = cat(x[b,0,:], max_t x[b,1:,:]) -> [B, 2D]; argmax [B,D] i32 for bwd. */
int ppt_cls_max_pool(const void *x, int x_dtype, int B, int T, int D, float *out, int32_t *argmax,
                     void *stream);
/* ---- part-segmentation decoder glue (models/pointbert/pointnet2_utils.py:297-467; csrc/interp.hip) -------------------
 * three_nn_interp_fwd: PointNetFeaturePropagation.forward :333-358 after the 3-NN search (ppt_knn_group_f32 with k = 3 and
 *   distances): weight_j = (1 / (dist_j + 1e-8)) / sum_j (...), out[b*N + n] = [points1[b,n,:D1] | sum_j weight_j *
 *   points2[b, idx[b,n,j], :D2] | 0 ...] with row stride ld_out (even, >= D1 + D2) in out_dtype -- directly the A operand of
 *   the first conv (torch.cat + K padding + dtype conversion fused); weight_out [B,N,3] f32 (may be NULL) keeps the
 *   normalised weights for the backward.  points1 may be NULL when D1 == 0.
 * scatter_rows_bwd: gradient of a row gather, d_src[b, s, :C] = sum over e in [0, E) with idx[b, e] == s of
 *   weight[b, e] * d_rows[b * (E / rows_div) + e / rows_div, col_off : col_off + C] (weight NULL = 1), entries taken in
 *   ascending e: deterministic, no atomics.  Interpolation: E = 3 N, rows_div = 3, weight = weight_out; neighbour gather
 *   of DGCNN_Propagation (:404-440): E = Nq * k, rows_div = 1, no weight.  S <= 65535, C <= 512.
 * sum_groups: out[g, :] = sum over j < k of x[g, j, :], x [G, k, C] f32, C % 4 == 0. */
int ppt_three_nn_interp_fwd(const float *points1, int D1, const float *points2, int D2, const int64_t *idx, const float *dist,
                            int B, int N, int S, void *out, int out_dtype, int ld_out, float *weight_out, void *stream);
int ppt_scatter_rows_bwd(const int64_t *idx, const float *weight, const float *d_rows, int64_t ld, int col_off, int B, int E,
                         int rows_div, int S, int C, float *d_src, void *stream);
int ppt_sum_groups(const float *x, int64_t G, int k, int C, float *out, void *stream);

/* ---- PointNet2 set-abstraction helpers (models/pointnet2/pointnet2_utils.py:228-266) ------------------------
 * A 1x1 convolution over gathered neighbours is linear, so the first layer of a grouped MLP is evaluated per
 * SOURCE point (P = W.[feat|xyz], a small GEMM) and per centre (Q = b - W_xyz.centre) and then gathered:
 *   y[b,s,k,:] = P[b, idx[b,s,k], :] + Q[b,s,:]      (ppt_gather_add; also emits the 32-row (sum, M2) BN partials)
 * instead of gathering 323-wide inputs and multiplying K times more rows.
 * pool_finish: out[g, c] = relu(scale[c] * (scale[c] >= 0 ? max : min)[g, c] + shift[c]) folded over `fold`
 *   consecutive pooled rows, written with leading dimension ld_out (concatenation of the MSG branches).
 * bn_act_rows: y = relu(x*scale + shift) * mask (mask may be NULL): the BatchNorm1d+ReLU+Dropout of the FC head. */
int ppt_gather_add(const void *P, int p_dtype, const float *Q, const int64_t *idx, int B, int Nsrc, int S, int K,
                   int C, void *y, int y_dtype, float *part_sum, float *part_m2, void *stream);
int ppt_pool_finish(const void *pmax, const void *pmin, int p_dtype, int G, int fold, int C, const float *scale,
                    const float *shift, void *out, int out_dtype, int64_t ld_out, void *stream);
int ppt_bn_act_rows(const float *x, int M, int C, const float *scale, const float *shift, const float *mask,
                    void *y, int y_dtype, void *stream);
/* ---- the prompt side's serial steps as single kernels (csrc/optim.hip) ------------------------------------------------
 * adamw_step: one torch.optim.AdamW update (main_cls.py:58-60, 198; amsgrad off) of a tensor of n f32 values:
 *   p *= 1 - lr * weight_decay; exp_avg += (g - exp_avg)(1 - beta1); exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps).   step >= 1 (this update's number).
 * prompt_rows: PromptLearner.forward (ULIP_models.py:104-151) + the positional add of encode_text (:210) in the text tower's
 *   row layout: out[i] = slot[i] >= 0 ? tokens[slot[i]] + pos_rows[i] : base[i]; base [rows, W] = frozen embedding +
 *   positional embedding per row (a constant), slot [rows] i32, pos_rows [rows, W].  W % 4 == 0.
 * prompt_rows_bwd: d_tokens[t] = scale * sum of g[row] over rows_of[t * max_rows + k] (ascending, -1 terminated); scale > 0:
 *   1, or the inverse of the gradient scale the text tower's backward ran with.
 * adamw_multi: adamw_step for `count` tensors (each with its own step number) in one launch per PPT_ADAMW_MAX_TENSORS tensors;
 *   `tensors` is a HOST array (it travels as a kernel argument). */
int ppt_adamw_step(float *p, float *g, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, uint64_t *skipped, void *stream);
                   /* grad_scale: g is multiplied by it first (1 / loss scale of a caller that scaled its loss; 1 = none) and,
                    * when != 1, the product is written back to g, so that g ends as the gradient of the un-scaled loss.
                    * skipped != NULL (device memory, 64-bit, atomically incremented): an element whose gradient is not
                    * finite is skipped (p, moments unchanged, g = 0) and counted there -- a 16-bit backward stage can
                    * overflow where the fp32 reference cannot, and one such step must not poison the moments for good.
                    * skipped == NULL: torch.optim.AdamW's own behaviour (a NaN gradient propagates; main_cls.py:205-207
                    * then stops the run) -- what the fp32 parity mode passes (ABI 5). */
#define PPT_ADAMW_MAX_TENSORS 64
typedef struct ppt_adamw_tensor {
    float *p, *g, *exp_avg, *exp_avg_sq;     /* device, f32, contiguous, n elements each */
    int64_t n;
    int step;                                /* >= 1: this update's number for this tensor (bias corrections) */
} ppt_adamw_tensor;
int ppt_adamw_multi(const ppt_adamw_tensor *tensors, int count, float lr, float beta1, float beta2, float eps, float weight_decay,
                    float grad_scale, uint64_t *skipped, void *stream);
int ppt_prompt_rows(const float *base, const int *slot, const float *tokens, const float *pos_rows, int rows, int W, float *out,
                    void *stream);
int ppt_prompt_rows_bwd(const float *g, const int *rows_of, int max_rows, int n_tok, int W, float scale, float *d_tokens, void *stream);

/* dtype conversion / transposition helpers (weights are converted once, activations never). */
int ppt_convert(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, void *stream);
/* dst = convert(src * scale), scale > 0: the operand copy of an fp32 activation gradient at the entry of a 16-bit backward
 * stage, multiplied by the stage's power-of-two gradient scale on the way (exact; ppt_amd/gradscale.py). */
int ppt_convert_scaled(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, float scale, void *stream);
/* The 16-bit operand copies of MANY trained weights in one launch (the part-seg decoder re-makes 15 of them, each with its
 * transpose, every step: pointbert/pointnet2_utils.py:297-467 trains all of its convs).  Item i: A = w[:, col0 : col0 + K] (f32, row
 * stride ldw elements), minus w[:, sub_col0 : sub_col0 + K] when sub_col0 >= 0 (DGCNN_Propagation's Wb - Wa, see
 * ppt_amd/autograd.py); out [N, Kp] = convert(A) with columns K .. Kp - 1 zero (may be NULL); out_t [Kp, N] = its transpose (may be
 * NULL).  `items` is a HOST array (it travels as a kernel argument, PPT_WPREP_MAX items per launch); dtype = PPT_BF16 | PPT_F16 | PPT_F32
 * (fp32 copies for the fp32 / split16 modes: ABI 6). */
#define PPT_WPREP_MAX 32
typedef struct ppt_wprep_item {
    const float *w; int64_t ldw;
    int N, col0, K, sub_col0, Kp;
    void *out, *out_t;
} ppt_wprep_item;
int ppt_weights_prep(const ppt_wprep_item *items, int count, int dtype, void *stream);
/* *flags |= bit when any of the n values of x (any dtype) is not finite; *maxabs (optional; a non-negative float the caller
 * initialised) = max(*maxabs, max |x| over the finite values).  One small launch: the overflow flag of a 16-bit stage's output
 * (ppt_amd/health.py) and the range probe of tools/fp16_stress.py. */
int ppt_health_check(const void *x, int x_dtype, int64_t n, uint32_t *flags, uint32_t bit, float *maxabs, void *stream);
/* *flags |= bit when any of the n labels is outside [0, C) and != ignore_index: a corrupt label, where ATen's
 * nn.CrossEntropyLoss raises a device assert (main_cls.py:196, main_partseg.py:213).  ppt_cross_entropy_rows / ppt_head_ce_bwd
 * make the loss NaN for such a row; this flag lets the caller tell a DATA bug from a numeric overflow (ppt_amd/health.py:
 * BIT_LABEL raises, it never demotes a stage). */
int ppt_labels_check(const int64_t *labels, int64_t n, int64_t C, int64_t ignore_index, uint32_t *flags, uint32_t bit, void *stream);
/* out[n][k] = convert(scale[n] * W[n][k]) for an [N, K] window of a row-major f32 matrix (row stride ldw elements; K % 4 == 0, ldw %
 * 4 == 0, 16-byte aligned), out [N, K] contiguous in out_dtype; bs (optional, [N] f32) = scale * b + shift (b / shift may be NULL).
 * A folded BatchNorm multiplied into the weight rows of the conv in front of it (csrc/mpn34.hip's W3s / gs). */
int ppt_scale_rows_convert(const float *W, int64_t ldw, int N, int K, const float *scale, void *out, int out_dtype, const float *b,
                           const float *shift, float *bs, void *stream);
/* src [rows, cols] contiguous -> dst [cols, rows] with row stride ld_dst >= rows (padding untouched) */
int ppt_transpose(const void *src, int src_dtype, void *dst, int dst_dtype, int rows, int cols,
                  int64_t ld_dst, void *stream);
/* column sums of x [M,D] (any dtype) -> partial [ceil(M/256), D] f32 (bias gradients); reduce with ppt_reduce_rows */
int ppt_col_sums(const void *x, int x_dtype, int M, int D, int64_t ldx, float *partial, void *stream);
/* sum the [P, D] partial buffers produced by col_sum / dw_partial style outputs -> [D] */
int ppt_reduce_rows(const float *partial, int P, int D, float *out, int accumulate, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PPT_HIP_H */
