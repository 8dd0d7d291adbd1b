// misc.hip -- small memory-bound helpers: cls/max pooling head (point_encoder.py:251), dtype
// conversion, transposition (weight preparation and the dW operands), partial-row reduction.
#include "ppt_common.h"

namespace {

// block = (sample b, 64-column slab): 16 waves stripe the tokens, lane == column (coalesced 256-B rows), four loads in
// flight per wave (with 4 waves and one load at a time the 513-token walk was latency: 38 us on the tail of the tower);
// the partial (max, first index) pairs fold through LDS with torch.max's first-maximum rule.
constexpr int CMP_W = 16;
template <typename TX>
__global__ __launch_bounds__(CMP_W * 64) void cls_max_pool_kernel(const TX *__restrict__ x, int T, int D, float *__restrict__ out,
                                                                  int32_t *__restrict__ argmax)
{
    __shared__ float bv[CMP_W][64];
    __shared__ int bi[CMP_W][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + lane;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    if (d < D) {
        const TX *xb = x + (size_t)b * T * D + d;
        int t = 1 + w;
        for (; t + 3 * CMP_W < T; t += 4 * CMP_W) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = dt<TX>::load(xb + (size_t)(t + k * CMP_W) * D);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[k] > best) { best = v[k]; besti = t + k * CMP_W; }
        }
        for (; t < T; t += CMP_W) {
            const float v = dt<TX>::load(xb + (size_t)t * D);
            if (v > best) { best = v; besti = t; }
        }
    }
    bv[w][lane] = best; bi[w][lane] = besti;
    __syncthreads();
    if (w == 0 && d < D) {
        for (int k = 1; k < CMP_W; ++k) {
            const float v = bv[k][lane];
            const int i = bi[k][lane];
            if (v > best || (v == best && i < besti)) { best = v; besti = i; }
        }
        out[(size_t)b * 2 * D + d] = dt<TX>::load(x + (size_t)b * T * D + d);
        out[(size_t)b * 2 * D + D + d] = best;
        if (argmax) argmax[(size_t)b * D + d] = besti;
    }
}

template <typename TS, typename TD>
__global__ void convert_kernel(const TS *__restrict__ s, TD *__restrict__ d, int64_t n, float scale)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dt<TD>::store(d + i, dt<TS>::load(s + i) * scale);
}

// fp32 -> 16-bit, eight values per thread (two 16-byte loads, one 16-byte store): the operand copy of an activation gradient at
// the entry of a 16-bit backward stage -- [B x N, 128 ... 512] rows in part segmentation -- where the element-per-thread walk above
// issued 16x the memory instructions.  scale: the stage's power-of-two gradient scale (ppt_convert_scaled), exact.
template <typename TD>
__global__ __launch_bounds__(256) void convert8_kernel(const float *__restrict__ s, TD *__restrict__ d, int64_t n8, float scale)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4 *>(s)[2 * i], b = reinterpret_cast<const float4 *>(s)[2 * i + 1];
        uint4 o;
        o.x = h16<TD>::pack2(a.x * scale, a.y * scale);
        o.y = h16<TD>::pack2(a.z * scale, a.w * scale);
        o.z = h16<TD>::pack2(b.x * scale, b.y * scale);
        o.w = h16<TD>::pack2(b.z * scale, b.w * scale);
        reinterpret_cast<uint4 *>(d)[i] = o;
    }
}

// out[n][k] = T(scale[n] * W[n][k]) over an [N, K] window of a row-major f32 matrix (row stride ldw); workgroup 0 also folds a bias:
// bs[n] = scale[n] * b[n] + shift[n].  The folded BatchNorm of a conv's OUTPUT multiplied into that conv's weight rows
// (engine.mini_pointnet -> csrc/mpn34.hip); one thread per 4 consecutive k.
template <typename TD>
__global__ __launch_bounds__(256) void scale_rows_convert_kernel(const float *__restrict__ W, int64_t ldw, int N, int K,
                                                                 const float *__restrict__ scale, TD *__restrict__ out,
                                                                 const float *__restrict__ b, const float *__restrict__ shift,
                                                                 float *__restrict__ bs)
{
    const int k4 = K >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (int64_t)N * k4) {
        const int n = (int)(i / k4), k = 4 * (int)(i - (int64_t)n * k4);
        const float sc = scale[n];
        const float4 v = *reinterpret_cast<const float4 *>(W + (int64_t)n * ldw + k);
        if constexpr (sizeof(TD) == 2) {
            *reinterpret_cast<uint2 *>(out + (int64_t)n * K + k) = make_uint2(h16<TD>::pack2(v.x * sc, v.y * sc), h16<TD>::pack2(v.z * sc, v.w * sc));
        } else {
            *reinterpret_cast<float4 *>(out + (int64_t)n * K + k) = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
        }
    }
    if (bs && blockIdx.x == 0)
        for (int n = threadIdx.x; n < N; n += 256) bs[n] = scale[n] * (b ? b[n] : 0.f) + (shift ? shift[n] : 0.f);
}

// flags[0] |= bit when any of the n values is not finite; maxabs (optional, one float, must start >= 0) = max(maxabs, max |x|) over the
// finite values (unsigned-integer atomic max on the bits: exact for non-negative floats).  The per-stage overflow flag of the
// 16-bit mode (ppt_amd/health.py) and the probe of tools/fp16_stress.py.
template <typename TS>
__global__ __launch_bounds__(256) void health_check_kernel(const TS *__restrict__ x, int64_t n, uint32_t *__restrict__ flags, uint32_t bit,
                                                           float *__restrict__ maxabs)
{
    bool bad = false;
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = fabsf(dt<TS>::load(x + i));
        if (!(v <= 3.0e38f)) bad = true; else mx = fmaxf(mx, v);
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flags, bit);
    if (maxabs) {
        mx = wave_reduce_max(mx);
        if ((threadIdx.x & 63) == 63) atomicMax(reinterpret_cast<unsigned int *>(maxabs), __float_as_uint(mx));
    }
}

// flags[0] |= bit when any label is outside [0, C) and is not ignore_index: a corrupt label (ATen: device assert)
__global__ __launch_bounds__(256) void labels_check_kernel(const int64_t *__restrict__ labels, int64_t n, int64_t C, int64_t ignore_index,
                                                           uint32_t *__restrict__ flags, uint32_t bit)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t y = labels[i];
        if ((y < 0 || y >= C) && y != ignore_index) bad = true;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flags, bit);
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_kernel(const TS *__restrict__ s, TD *__restrict__ d, int rows, int cols,
                                                        int64_t ldd)
{
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? dt<TS>::load(s + (size_t)r * cols + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;            // output row = c, output col = r
        if (c < cols && r < rows) dt<TD>::store(d + (size_t)c * ldd + r, tile[tx][i]);
    }
}

template <typename TX>
__global__ __launch_bounds__(256) void col_sums_kernel(const TX *__restrict__ x, int M, int D, int64_t ldx,
                                                       float *__restrict__ part)
{
    const int r0 = blockIdx.y * 256, r1 = min(M, r0 + 256);
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += dt<TX>::load(x + (int64_t)r * ldx + d);
    part[(size_t)blockIdx.y * D + d] = s;
}

// bf16, D % 8 == 0: a workgroup owns 256 rows x 128 columns as 16 row groups x 16 column octets (16-byte loads, 256-byte row
// pieces), fp32 sums, the 16 group sums folded through LDS in a fixed order.  (The one-thread-per-column walk: 64 us for a
// 32768 x 256 bias gradient, 0.26 TB/s; 7 per part-seg step.)
template <typename F>
__global__ __launch_bounds__(256) void col_sums_vec8_bf16_kernel(const bf16_t *__restrict__ x, int M, int D, int64_t ldx,
                                                                 float *__restrict__ part)
{
    __shared__ float red[16][16][9];
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int d = blockIdx.x * 128 + q * 8;
    const int r0 = blockIdx.y * 256 + g * 16;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (d < D) {
#pragma unroll 8
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + i;
            if (r < M) {
                const uint4 v = *reinterpret_cast<const uint4 *>(x + (int64_t)r * ldx + d);
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[2 * j] += h16<F>::lo(wv[j]);
                    s[2 * j + 1] += h16<F>::hi(wv[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[g][q][j] = s[j];
    __syncthreads();
    if (g < 8 && d < D) {                                 // thread (g, q) finishes column d + g
        float t = red[0][q][g];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][q][g];
        if (d + g < D) part[(size_t)blockIdx.y * D + d + g] = t;
    }
}

// fp32, D % 4 == 0: a workgroup owns 256 rows x 64 columns as 16 row groups x 16 column quads (float4 loads, 256-byte row
// pieces), the 16 group sums folded through LDS in a fixed order.  The one-thread-per-column walk above took 74 us for a
// 32768 x 256 bias gradient (0.45 TB/s; 7 per part-seg step).
__global__ __launch_bounds__(256) void col_sums_vec_kernel(const float *__restrict__ x, int M, int D, int64_t ldx,
                                                           float *__restrict__ part)
{
    __shared__ float4 red[16][17];
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int d = blockIdx.x * 64 + q * 4;
    const int r0 = blockIdx.y * 256 + g * 16;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d < D) {
#pragma unroll 8
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + i;
            if (r < M) {
                const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t)r * ldx + d);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
    }
    red[g][q] = s;
    __syncthreads();
    if (g == 0 && d < D) {
        float4 t = red[0][q];
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float4 v = red[k][q];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        *reinterpret_cast<float4 *>(part + (size_t)blockIdx.y * D + d) = t;
    }
}

// 32 columns x 32 row stripes per workgroup (128-byte row pieces per half wave), fp64 partial sums folded through LDS in
// a fixed order: one serial walk per column took 49 us for a 512 x 1536 table (part-seg decoder, 21 calls per step)
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float *__restrict__ part, int P, int D, float *__restrict__ out,
                                                           int accumulate)
{
    __shared__ double red[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int d = blockIdx.x * 32 + tx;
    double s = 0.0;
    if (d < D)
        for (int p = ty; p < P; p += 32) s += (double)part[(size_t)p * D + d];
    red[ty][tx] = s;
    __syncthreads();
    for (int off = 16; off > 0; off >>= 1) {
        if (ty < off) red[ty][tx] += red[ty + off][tx];
        __syncthreads();
    }
    if (ty == 0 && d < D) out[d] = accumulate ? out[d] + (float)red[0][tx] : (float)red[0][tx];
}

// few partial rows, many columns (the S slices of a split weight gradient: S x (N1*N2)): one thread per column
__global__ __launch_bounds__(256) void reduce_rows_small_kernel(const float *__restrict__ part, int P, int64_t D,
                                                                float *__restrict__ out, int accumulate)
{
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    double s = 0.0;
    for (int p = 0; p < P; ++p) s += (double)part[(int64_t)p * D + d];
    out[d] = accumulate ? out[d] + (float)s : (float)s;
}

// ---- operand copies of many weights in ONE launch (ppt_weights_prep) --------------------------------------------------------------
// item i: A = W_i[:, col0 : col0 + K] (f32, row stride ldw), minus W_i[:, sub_col0 : sub_col0 + K] when sub_col0 >= 0 (the DGCNN layer's
// Wb - Wa); out [N, Kp] = A in the 16-bit format, columns K .. Kp - 1 zero; out_t [Kp, N] = its transpose (the B operand of dX = dY @ W).
// A workgroup = one 32 x 32 tile of one item (found by a scan of the block prefix), transposed through LDS.
struct wprep_table {
    const float *w[PPT_WPREP_MAX];
    void *out[PPT_WPREP_MAX], *out_t[PPT_WPREP_MAX];
    int ldw[PPT_WPREP_MAX], N[PPT_WPREP_MAX], col0[PPT_WPREP_MAX], K[PPT_WPREP_MAX], sub_col0[PPT_WPREP_MAX], Kp[PPT_WPREP_MAX];
    int first_block[PPT_WPREP_MAX + 1];
    int count;
};

template <typename TD>
__global__ __launch_bounds__(256) void weights_prep_kernel(const wprep_table tb)
{
    __shared__ float tile[32][33];
    int t = 0;
    while (t + 1 < tb.count && (int)blockIdx.x >= tb.first_block[t + 1]) ++t;
    const int N = tb.N[t], K = tb.K[t], Kp = tb.Kp[t], ldw = tb.ldw[t];
    const int tiles_k = (Kp + 31) / 32;
    const int lb = (int)blockIdx.x - tb.first_block[t];
    const int n0 = (lb / tiles_k) * 32, k0 = (lb % tiles_k) * 32;
    const float *w = tb.w[t] + tb.col0[t];
    const float *ws = tb.sub_col0[t] >= 0 ? tb.w[t] + tb.sub_col0[t] : nullptr;
    TD *out = reinterpret_cast<TD *>(tb.out[t]), *out_t = reinterpret_cast<TD *>(tb.out_t[t]);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, k = k0 + tx;
        float v = 0.f;
        if (n < N && k < K) {
            v = w[(int64_t)n * ldw + k];
            if (ws) v -= ws[(int64_t)n * ldw + k];
        }
        tile[i][tx] = v;
        if (n < N && k < Kp && out) dt<TD>::store(out + (int64_t)n * Kp + k, v);
    }
    __syncthreads();
    if (out_t)
        for (int i = ty; i < 32; i += 8) {
            const int k = k0 + i, n = n0 + tx;
            if (k < Kp && n < N) dt<TD>::store(out_t + (int64_t)k * N + n, tile[tx][i]);
        }
}

template <typename TS>
int convert_from(const void *src, void *dst, int dd, int64_t n, float scale, hipStream_t s)
{
    if constexpr (sizeof(TS) == 4) {
        if ((dd == PPT_BF16 || dd == PPT_F16) && (n & 7) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
            const int64_t n8 = n >> 3;
            const unsigned grid8 = (unsigned)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
            if (dd == PPT_BF16) hipLaunchKernelGGL(convert8_kernel<bf16_t>, dim3(grid8), dim3(256), 0, s, (const float *)src, (bf16_t *)dst, n8, scale);
            else hipLaunchKernelGGL(convert8_kernel<f16_t>, dim3(grid8), dim3(256), 0, s, (const float *)src, (f16_t *)dst, n8, scale);
            PPT_CHECK_LAUNCH();
            return PPT_OK;
        }
    }
    const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (dd == PPT_BF16) hipLaunchKernelGGL((convert_kernel<TS, bf16_t>), dim3(grid), dim3(256), 0, s, (const TS *)src, (bf16_t *)dst, n, scale);
    else if (dd == PPT_F16) hipLaunchKernelGGL((convert_kernel<TS, f16_t>), dim3(grid), dim3(256), 0, s, (const TS *)src, (f16_t *)dst, n, scale);
    else if (dd == PPT_F32) hipLaunchKernelGGL((convert_kernel<TS, float>), dim3(grid), dim3(256), 0, s, (const TS *)src, (float *)dst, n, scale);
    else return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

template <typename TS>
int transpose_from(const void *src, void *dst, int dd, int rows, int cols, int64_t ldd, hipStream_t s)
{
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    if (dd == PPT_BF16) hipLaunchKernelGGL((transpose_kernel<TS, bf16_t>), grid, dim3(256), 0, s, (const TS *)src, (bf16_t *)dst, rows, cols, ldd);
    else if (dd == PPT_F16) hipLaunchKernelGGL((transpose_kernel<TS, f16_t>), grid, dim3(256), 0, s, (const TS *)src, (f16_t *)dst, rows, cols, ldd);
    else if (dd == PPT_F32) hipLaunchKernelGGL((transpose_kernel<TS, float>), grid, dim3(256), 0, s, (const TS *)src, (float *)dst, rows, cols, ldd);
    else return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

}  // namespace

extern "C" int ppt_cls_max_pool(const void *x, int x_dtype, int B, int T, int D, float *out, int32_t *argmax, void *stream)
{
    if (!x || !out || B <= 0 || T < 2 || D <= 0) return PPT_EINVAL;
    if (x_dtype == PPT_F32)
        hipLaunchKernelGGL(cls_max_pool_kernel<float>, dim3((D + 63) / 64, B), dim3(CMP_W * 64), 0, ppt_stream(stream), (const float *)x, T, D, out, argmax);
    else if (x_dtype == PPT_BF16)
        hipLaunchKernelGGL(cls_max_pool_kernel<bf16_t>, dim3((D + 63) / 64, B), dim3(CMP_W * 64), 0, ppt_stream(stream), (const bf16_t *)x, T, D, out, argmax);
    else if (x_dtype == PPT_F16)
        hipLaunchKernelGGL(cls_max_pool_kernel<f16_t>, dim3((D + 63) / 64, B), dim3(CMP_W * 64), 0, ppt_stream(stream), (const f16_t *)x, T, D, out, argmax);
    else
        return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_weights_prep(const ppt_wprep_item *items, int count, int dtype, void *stream)
{
    if (!items || count <= 0 || (dtype != PPT_BF16 && dtype != PPT_F16 && dtype != PPT_F32)) return PPT_EINVAL;
    for (int c0 = 0; c0 < count; c0 += PPT_WPREP_MAX) {
        wprep_table tb;
        tb.count = count - c0 < PPT_WPREP_MAX ? count - c0 : PPT_WPREP_MAX;
        int64_t blocks = 0;
        for (int i = 0; i < tb.count; ++i) {
            const ppt_wprep_item &it = items[c0 + i];
            if (!it.w || (!it.out && !it.out_t) || it.N <= 0 || it.K <= 0 || it.Kp < it.K || it.col0 < 0 || it.ldw < it.col0 + it.K ||
                (it.sub_col0 >= 0 && it.ldw < it.sub_col0 + it.K))
                return PPT_EINVAL;
            tb.w[i] = it.w; tb.out[i] = it.out; tb.out_t[i] = it.out_t; tb.ldw[i] = (int)it.ldw; tb.N[i] = it.N; tb.col0[i] = it.col0;
            tb.K[i] = it.K; tb.sub_col0[i] = it.sub_col0; tb.Kp[i] = it.Kp;
            tb.first_block[i] = (int)blocks;
            blocks += (int64_t)((it.N + 31) / 32) * ((it.Kp + 31) / 32);
            if (blocks > 0x7fffffff) return PPT_EUNSUPPORTED;
        }
        tb.first_block[tb.count] = (int)blocks;
        if (dtype == PPT_F16) hipLaunchKernelGGL(weights_prep_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, ppt_stream(stream), tb);
        else if (dtype == PPT_F32) hipLaunchKernelGGL(weights_prep_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, ppt_stream(stream), tb);   // (fp32 / split16 modes)
        else hipLaunchKernelGGL(weights_prep_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, ppt_stream(stream), tb);
        PPT_CHECK_LAUNCH();
    }
    return PPT_OK;
}

extern "C" int ppt_health_check(const void *x, int x_dtype, int64_t n, uint32_t *flags, uint32_t bit, float *maxabs, void *stream)
{
    if (!x || !flags || n <= 0) return PPT_EINVAL;
    const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    if (x_dtype == PPT_F32) hipLaunchKernelGGL(health_check_kernel<float>, dim3(grid), dim3(256), 0, ppt_stream(stream), (const float *)x, n, flags, bit, maxabs);
    else if (x_dtype == PPT_BF16) hipLaunchKernelGGL(health_check_kernel<bf16_t>, dim3(grid), dim3(256), 0, ppt_stream(stream), (const bf16_t *)x, n, flags, bit, maxabs);
    else if (x_dtype == PPT_F16) hipLaunchKernelGGL(health_check_kernel<f16_t>, dim3(grid), dim3(256), 0, ppt_stream(stream), (const f16_t *)x, n, flags, bit, maxabs);
    else return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_labels_check(const int64_t *labels, int64_t n, int64_t C, int64_t ignore_index, uint32_t *flags, uint32_t bit, void *stream)
{
    if (!labels || !flags || n <= 0 || C <= 0) return PPT_EINVAL;
    const unsigned grid = (unsigned)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
    hipLaunchKernelGGL(labels_check_kernel, dim3(grid), dim3(256), 0, ppt_stream(stream), labels, n, C, ignore_index, flags, bit);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_scale_rows_convert(const float *W, int64_t ldw, int N, int K, const float *scale, void *out, int out_dtype,
                                      const float *b, const float *shift, float *bs, void *stream)
{
    if (!W || !scale || !out || N <= 0 || K <= 0 || (K & 3) || ldw < K || (ldw & 3)) return PPT_EINVAL;
    if ((((uintptr_t)W) | ((uintptr_t)out)) & 15) return PPT_EINVAL;
    const unsigned grid = (unsigned)(((int64_t)N * (K / 4) + 255) / 256);
    if (out_dtype == PPT_BF16)
        hipLaunchKernelGGL(scale_rows_convert_kernel<bf16_t>, dim3(grid), dim3(256), 0, ppt_stream(stream), W, ldw, N, K, scale, (bf16_t *)out, b, shift, bs);
    else if (out_dtype == PPT_F16)
        hipLaunchKernelGGL(scale_rows_convert_kernel<f16_t>, dim3(grid), dim3(256), 0, ppt_stream(stream), W, ldw, N, K, scale, (f16_t *)out, b, shift, bs);
    else if (out_dtype == PPT_F32)
        hipLaunchKernelGGL(scale_rows_convert_kernel<float>, dim3(grid), dim3(256), 0, ppt_stream(stream), W, ldw, N, K, scale, (float *)out, b, shift, bs);
    else
        return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_convert_scaled(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, float scale, void *stream)
{
    if (!src || !dst || n <= 0 || !(scale > 0.f)) return PPT_EINVAL;
    if (src_dtype == PPT_F32) return convert_from<float>(src, dst, dst_dtype, n, scale, ppt_stream(stream));
    if (src_dtype == PPT_BF16) return convert_from<bf16_t>(src, dst, dst_dtype, n, scale, ppt_stream(stream));
    if (src_dtype == PPT_F16) return convert_from<f16_t>(src, dst, dst_dtype, n, scale, ppt_stream(stream));
    return PPT_EINVAL;
}

extern "C" int ppt_convert(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t n, void *stream)
{
    return ppt_convert_scaled(src, src_dtype, dst, dst_dtype, n, 1.0f, stream);
}

extern "C" int ppt_transpose(const void *src, int src_dtype, void *dst, int dst_dtype, int rows, int cols, int64_t ld_dst,
                             void *stream)
{
    if (!src || !dst || rows <= 0 || cols <= 0 || ld_dst < rows) return PPT_EINVAL;
    if (src_dtype == PPT_F32) return transpose_from<float>(src, dst, dst_dtype, rows, cols, ld_dst, ppt_stream(stream));
    if (src_dtype == PPT_BF16) return transpose_from<bf16_t>(src, dst, dst_dtype, rows, cols, ld_dst, ppt_stream(stream));
    if (src_dtype == PPT_F16) return transpose_from<f16_t>(src, dst, dst_dtype, rows, cols, ld_dst, ppt_stream(stream));
    return PPT_EINVAL;
}

extern "C" int ppt_col_sums(const void *x, int x_dtype, int M, int D, int64_t ldx, float *partial, void *stream)
{
    if (!x || !partial || M <= 0 || D <= 0) return PPT_EINVAL;
    dim3 grid((D + 255) / 256, (M + 255) / 256);
    if (x_dtype == PPT_F32 && D % 4 == 0 && ldx % 4 == 0 && !(((uintptr_t)x | (uintptr_t)partial) & 15))
        hipLaunchKernelGGL(col_sums_vec_kernel, dim3((D + 63) / 64, (M + 255) / 256), dim3(256), 0, ppt_stream(stream),
                           (const float *)x, M, D, ldx, partial);
    else if (x_dtype == PPT_F32)
        hipLaunchKernelGGL(col_sums_kernel<float>, grid, dim3(256), 0, ppt_stream(stream), (const float *)x, M, D, ldx, partial);
    else if (x_dtype == PPT_BF16 && D % 8 == 0 && ldx % 8 == 0 && !((uintptr_t)x & 15))
        hipLaunchKernelGGL(col_sums_vec8_bf16_kernel<bf16_t>, dim3((D + 127) / 128, (M + 255) / 256), dim3(256), 0, ppt_stream(stream),
                           (const bf16_t *)x, M, D, ldx, partial);
    else if (x_dtype == PPT_F16 && D % 8 == 0 && ldx % 8 == 0 && !((uintptr_t)x & 15))
        hipLaunchKernelGGL(col_sums_vec8_bf16_kernel<f16_t>, dim3((D + 127) / 128, (M + 255) / 256), dim3(256), 0, ppt_stream(stream),
                           (const bf16_t *)x, M, D, ldx, partial);
    else if (x_dtype == PPT_BF16)
        hipLaunchKernelGGL(col_sums_kernel<bf16_t>, grid, dim3(256), 0, ppt_stream(stream), (const bf16_t *)x, M, D, ldx, partial);
    else if (x_dtype == PPT_F16)
        hipLaunchKernelGGL(col_sums_kernel<f16_t>, grid, dim3(256), 0, ppt_stream(stream), (const f16_t *)x, M, D, ldx, partial);
    else
        return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_reduce_rows(const float *partial, int P, int D, float *out, int accumulate, void *stream)
{
    if (!partial || !out || P <= 0 || D <= 0) return PPT_EINVAL;
    if (P <= 32)
        hipLaunchKernelGGL(reduce_rows_small_kernel, dim3((D + 255) / 256), dim3(256), 0, ppt_stream(stream), partial, P,
                           (int64_t)D, out, accumulate);
    else
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((D + 31) / 32), dim3(1024), 0, ppt_stream(stream), partial, P, D, out,
                           accumulate);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
