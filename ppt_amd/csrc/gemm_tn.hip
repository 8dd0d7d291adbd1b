// gemm_tn.hip -- weight-gradient GEMM  C[N1,N2] = A[M,N1]^T @ B[M,N2]  (dW = dY^T X) for bf16 operands as they are
// stored: both row-major over the reduction dimension M.  The NT kernel of gemm.hip needs K-contiguous operands, i.e.
// a transposed copy of both activations per gradient (0.35 ms of a C3 step, 0.63 ms of a C5 step); here the 32-row
// slabs go to LDS in their natural layout and BOTH MFMA operands are fetched with the transposing LDS read
// (ds_read_b64_tr_b16, the V^T / Q^T path of attention_mfma.hip: same image, same bank-half swizzle).
// One workgroup = 128 x 128 (or 64 x 64) outputs of one M-slice (blockIdx.z); the S slice products land in part[S][N1][N2] (fp32)
// and are folded by ppt_reduce_rows in a fixed order.  Requires M % (32 * S) == 0, N1 % 8 == 0, N2 % 8 == 0.
#include "ppt_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s4_t;

constexpr int TN_KS = 32;                // rows of M per slab
constexpr int TN_IMG = TN_KS * 128;      // one LDS image: 32 rows x 64 bf16 columns

__device__ __forceinline__ int tn_off(int row, int dbyte) { return row * 128 + (dbyte ^ (((row >> 1) & 1) << 6)); }

__device__ __forceinline__ uint4 tn_frag(const unsigned char *img, int row0, int dbyte)
{
    struct { s4_t a, b; } f;
    f.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(img + tn_off(row0, dbyte)));
    f.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(img + tn_off(row0 + 8, dbyte)));
    return __builtin_bit_cast(uint4, f);
}

// 2 x 2 waves, each TI x TJ MFMA tiles of 32 x 32: the workgroup owns 64 TI x 64 TJ outputs.  Operand columns are kept in
// TI (TJ) images of 64 columns so that every transposed read sees the 128-byte rows the swizzle was made for.
template <typename F, int TI, int TJ>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t *__restrict__ A, int64_t lda, const bf16_t *__restrict__ B,
                                                         int64_t ldb, int N1, int N2, int rows_per_slice,
                                                         float *__restrict__ part)
{
    constexpr int STAGE = (TI + TJ) * TN_IMG;
    __shared__ __align__(16) unsigned char smem[2 * STAGE];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = w >> 1, wj = w & 1;
    const int n1_0 = blockIdx.y * (64 * TI), n2_0 = blockIdx.x * (64 * TJ);
    const int64_t m0 = (int64_t)blockIdx.z * rows_per_slice;
    const int nslab = rows_per_slice / TN_KS;

    // staging: 16-byte chunks, 8 TI (8 TJ) per row; thread t moves chunks t, t + 256, ...
    uint4 ra[TI], rb[TJ];
    int a_row[TI], a_col[TI], a_lds[TI], b_row[TJ], b_col[TJ], b_lds[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int idx = threadIdx.x + 256 * i, ch = idx % (8 * TI);
        a_row[i] = idx / (8 * TI);
        a_col[i] = min(n1_0 + ch * 8, N1 - 8);                         // clamped: those outputs are never stored
        a_lds[i] = (ch >> 3) * TN_IMG + tn_off(a_row[i], (ch & 7) * 16);
    }
#pragma unroll
    for (int i = 0; i < TJ; ++i) {
        const int idx = threadIdx.x + 256 * i, ch = idx % (8 * TJ);
        b_row[i] = idx / (8 * TJ);
        b_col[i] = min(n2_0 + ch * 8, N2 - 8);
        b_lds[i] = (TI + (ch >> 3)) * TN_IMG + tn_off(b_row[i], (ch & 7) * 16);
    }
#define TN_LOAD(slab)                                                                                                   \
    {                                                                                                                  \
        const int64_t m_ = m0 + (int64_t)(slab) * TN_KS;                                                               \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) ra[i] = *reinterpret_cast<const uint4 *>(A + (m_ + a_row[i]) * lda + a_col[i]); \
        _Pragma("unroll") for (int i = 0; i < TJ; ++i) rb[i] = *reinterpret_cast<const uint4 *>(B + (m_ + b_row[i]) * ldb + b_col[i]); \
    }
#define TN_STORE(buf)                                                                                                  \
    {                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) *reinterpret_cast<uint4 *>(smem + (buf) * STAGE + a_lds[i]) = ra[i];         \
        _Pragma("unroll") for (int i = 0; i < TJ; ++i) *reinterpret_cast<uint4 *>(smem + (buf) * STAGE + b_lds[i]) = rb[i];         \
    }

    // per-lane constant part of the transposed reads (as attention_mfma.hip): lane = 16g + 4q + p
    const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_row = 4 * (g >> 1) + tq;
    const int tr_dbyte = (16 * (g & 1) + 4 * tp) * 2;
    // this wave's first column, as (image, byte in the image row)
    const int a_img = (wi * 64 * TI) >> 7, a_db = (wi * 64 * TI) & 127;
    const int b_img = TI + ((wj * 64 * TJ) >> 7), b_db = (wj * 64 * TJ) & 127;

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    TN_LOAD(0);
    TN_STORE(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
        TN_LOAD(min(s + 1, nslab - 1));          // unconditional: a branch here parks the staging registers in scratch
        const unsigned char *At = smem + cur * STAGE + a_img * TN_IMG, *Bt = smem + cur * STAGE + b_img * TN_IMG;
#pragma unroll
        for (int k = 0; k < TN_KS / 16; ++k) {
            uint4 af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = tn_frag(At, 16 * k + tr_row, tr_dbyte + a_db + 64 * i);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = tn_frag(Bt, 16 * k + tr_row, tr_dbyte + b_db + 64 * j);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = h16<F>::mfma32(af[i], bf[j], acc[i][j]);
        }
        TN_STORE(cur ^ 1);
        __syncthreads();
    }
    // C layout: column (lane & 31), rows (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    float *out = part + (int64_t)blockIdx.z * N1 * N2;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int n2 = n2_0 + 32 * (wj * TJ + j) + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n1 = n1_0 + 32 * (wi * TI + i) + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (n1 < N1 && n2 < N2) out[(int64_t)n1 * N2 + n2] = acc[i][j][e];
            }
        }
}

}  // namespace

static int gemm_tn_any(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t M, int N1, int N2, int n_slices,
                       float *part, int dtype, void *stream)
{
    if (!A || !B || !part || M <= 0 || N1 <= 0 || N2 <= 0 || n_slices <= 0) return PPT_EINVAL;
    if ((N1 % 8) || (N2 % 8) || (lda % 8) || (ldb % 8) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return PPT_EINVAL;
    if (M % (TN_KS * (int64_t)n_slices)) return PPT_EUNSUPPORTED;
    const int rows = (int)(M / n_slices);
    // 128 x 128 tiles when they (times the slices) fill the chip, 64 x 64 otherwise
    const int64_t big = (int64_t)((N1 + 127) / 128) * ((N2 + 127) / 128) * n_slices;
    if (n_slices > 65535 || (N1 + 63) / 64 > 65535) return PPT_EUNSUPPORTED;
#define PPT_TN(FF)                                                                                                                     \
    do {                                                                                                                               \
        if (big >= 256)                                                                                                                \
            hipLaunchKernelGGL((gemm_tn_kernel<FF, 2, 2>), dim3((N2 + 127) / 128, (N1 + 127) / 128, n_slices), dim3(256), 0,            \
                               ppt_stream(stream), (const bf16_t *)A, lda, (const bf16_t *)B, ldb, N1, N2, rows, part);                  \
        else                                                                                                                           \
            hipLaunchKernelGGL((gemm_tn_kernel<FF, 1, 1>), dim3((N2 + 63) / 64, (N1 + 63) / 64, n_slices), dim3(256), 0,                \
                               ppt_stream(stream), (const bf16_t *)A, lda, (const bf16_t *)B, ldb, N1, N2, rows, part);                  \
    } while (0)
    if (dtype == PPT_F16) PPT_TN(f16_t); else PPT_TN(bf16_t);
#undef PPT_TN
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_gemm_tn_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t M, int N1, int N2, int n_slices,
                                float *part, void *stream)
{
    return gemm_tn_any(A, lda, B, ldb, M, N1, N2, n_slices, part, PPT_BF16, stream);
}

/* the same for operands of either 16-bit format (dtype = PPT_BF16 or PPT_F16) */
extern "C" int ppt_gemm_tn_half(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t M, int N1, int N2, int n_slices,
                                float *part, int dtype, void *stream)
{
    if (dtype != PPT_BF16 && dtype != PPT_F16) return PPT_EINVAL;
    return gemm_tn_any(A, lda, B, ldb, M, N1, N2, n_slices, part, dtype, stream);
}
