// norm.hip -- LayerNorm forward / backward (point_encoder.py:65,69,152; ULIP_models.py:21-27).
// One wave per token row (D <= 1024): the row lives in registers, mean and variance are the
// two-pass forms the reference computes, reductions are wave-wide shuffles -- no LDS, no barrier.
// The forward optionally fuses the `x + pos` of TransformerEncoder.forward (point_encoder.py:103)
// and of encode_text (ULIP_models.py:210), writing the updated residual stream back.
#include "ppt_common.h"

namespace {

constexpr int MAX_EPL = 16;   // elements per lane -> D <= 1024

template <typename TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ add,
                                                     int add_rows, float *__restrict__ xs,
                                                     const float *__restrict__ w, const float *__restrict__ b,
                                                     TY *__restrict__ y, float *__restrict__ mean_out,
                                                     float *__restrict__ rstd_out, int M, int D, float eps)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *xr = x + (size_t)row * D;
    const float *ar = add ? add + (size_t)(add_rows > 0 ? row % add_rows : row) * D : nullptr;
    float v[MAX_EPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_EPL; ++j) {
        const int e = lane + 64 * j;
        float t = 0.f;
        if (e < D) { t = xr[e]; if (ar) t += ar[e]; }
        v[j] = t; s += t;
    }
    const float mean = wave_reduce_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_EPL; ++j) {
        const int e = lane + 64 * j;
        const float d = e < D ? v[j] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_reduce_sum(q) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < MAX_EPL; ++j) {
        const int e = lane + 64 * j;
        if (e < D) {
            if (xs) xs[(size_t)row * D + e] = v[j];
            dt<TY>::store(y + (size_t)row * D + e, (v[j] - mean) * rstd * w[e] + b[e]);
        }
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// D % 8 == 0 and D <= 512 (both towers: 384, 512): lane c owns columns 8c .. 8c+7, so a row is two 16-byte loads
// (+ two for the fused add) and ONE 16-byte bf16 store per lane instead of six 4-byte loads and six 2-byte stores;
// gamma / beta stay in registers while the wave walks its rows (rows strided by the number of waves in the grid).
// Same arithmetic as ln_fwd_kernel (two-pass mean / variance), summed in a different lane order.
// SUM (round 5): the row is x + bias + parts[0] + ... + parts[S-1] (added in that order) -- the consumer side of a split-K GEMM
// whose S partial products were written as plain fp32 slices [S, M, D]: the reduction costs no launch of its own, no atomics, and
// the summation order is fixed (the prompt chain's K = 2048 linears: engine.text_tower_forward).
template <typename TY, bool SUM = false>
__global__ __launch_bounds__(256) void ln_fwd_vec8_kernel(const float *__restrict__ x, const float *__restrict__ add,
                                                          int add_rows, float *__restrict__ xs,
                                                          const float *__restrict__ w, const float *__restrict__ b,
                                                          TY *__restrict__ y, float *__restrict__ mean_out,
                                                          float *__restrict__ rstd_out, int M, int D, float eps, int prio,
                                                          const float *__restrict__ sum_bias = nullptr,
                                                          const float *__restrict__ parts = nullptr, int S = 0)
{
    PPT_PRIO(prio);
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    const bool on = lane * 8 < D;
    const int c = on ? lane * 8 : 0;
    float g[8], be[8];
    {
        const float4 g0 = *reinterpret_cast<const float4 *>(w + c), g1 = *reinterpret_cast<const float4 *>(w + c + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(b + c), b1 = *reinterpret_cast<const float4 *>(b + c + 4);
        g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
        be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w;
    }
    const float inv_d = 1.0f / (float)D;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += nw) {
        const float *xr = x + (size_t)row * D + c;
        float4 v0 = *reinterpret_cast<const float4 *>(xr), v1 = *reinterpret_cast<const float4 *>(xr + 4);
        if (add) {
            const float *ar = add + (size_t)(add_rows > 0 ? row % add_rows : row) * D + c;
            const float4 a0 = *reinterpret_cast<const float4 *>(ar), a1 = *reinterpret_cast<const float4 *>(ar + 4);
            v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w; v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
        }
        if constexpr (SUM) {
            if (sum_bias) {
                const float4 a0 = *reinterpret_cast<const float4 *>(sum_bias + c), a1 = *reinterpret_cast<const float4 *>(sum_bias + c + 4);
                v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w; v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
            }
            for (int z = 0; z < S; ++z) {
                const float *pr = parts + ((size_t)z * M + row) * D + c;
                const float4 a0 = *reinterpret_cast<const float4 *>(pr), a1 = *reinterpret_cast<const float4 *>(pr + 4);
                v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w; v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
            }
        }
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
        const float mean = wave_reduce_sum(on ? s : 0.f) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.0f / sqrtf(wave_reduce_sum(on ? q : 0.f) * inv_d + eps);
        if (on) {
            if (xs) {
                *reinterpret_cast<float4 *>(xs + (size_t)row * D + c) = v0;
                *reinterpret_cast<float4 *>(xs + (size_t)row * D + c + 4) = v1;
            }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[j] - mean) * rstd * g[j] + be[j];
            if constexpr (sizeof(TY) == 2) {
                *reinterpret_cast<uint4 *>(y + (size_t)row * D + c) =
                    make_uint4(h16<TY>::pack2(o[0], o[1]), h16<TY>::pack2(o[2], o[3]), h16<TY>::pack2(o[4], o[5]), h16<TY>::pack2(o[6], o[7]));
            } else {
                *reinterpret_cast<float4 *>(y + (size_t)row * D + c) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4 *>(y + (size_t)row * D + c + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
    }
}

// wave g handles rows [g*rpw, (g+1)*rpw); dw/db partial row g.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ xs,
                                                     const float *__restrict__ w, const float *__restrict__ mean,
                                                     const float *__restrict__ rstd, float *__restrict__ dx,
                                                     int accumulate, void *__restrict__ dx_copy, int copy_dtype,
                                                     float *__restrict__ dw_part,
                                                     float *__restrict__ db_part, int rpw, int M, int D, int prio)
{
    PPT_PRIO(prio);
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = g * rpw, r1 = min(M, r0 + rpw);
    if (r0 >= M) return;
    float wv[MAX_EPL], dwa[MAX_EPL], dba[MAX_EPL];
#pragma unroll
    for (int j = 0; j < MAX_EPL; ++j) {
        const int e = lane + 64 * j;
        wv[j] = e < D ? w[e] : 0.f;
        dwa[j] = 0.f; dba[j] = 0.f;
    }
    for (int row = r0; row < r1; ++row) {
        const float mu = mean[row], rs = rstd[row];
        float gv[MAX_EPL], xh[MAX_EPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_EPL; ++j) {
            const int e = lane + 64 * j;
            float d = 0.f, xhat = 0.f;
            if (e < D) { d = dy[(size_t)row * D + e]; xhat = (xs[(size_t)row * D + e] - mu) * rs; }
            dwa[j] += d * xhat; dba[j] += d;
            gv[j] = d * wv[j]; xh[j] = xhat;
            s1 += gv[j]; s2 += gv[j] * xhat;
        }
        s1 = wave_reduce_sum(s1) / (float)D;
        s2 = wave_reduce_sum(s2) / (float)D;
#pragma unroll
        for (int j = 0; j < MAX_EPL; ++j) {
            const int e = lane + 64 * j;
            if (e < D) {
                const float r = rs * (gv[j] - s1 - xh[j] * s2);
                float *o = dx + (size_t)row * D + e;
                const float t = accumulate ? *o + r : r;
                *o = t;
                if (dx_copy) {
                    if (copy_dtype != PPT_F32) ((uint16_t *)dx_copy)[(size_t)row * D + e] = from_f32_dt(copy_dtype, t);
                    else ((float *)dx_copy)[(size_t)row * D + e] = t;
                }
            }
        }
    }
    if (dw_part) {
#pragma unroll
        for (int j = 0; j < MAX_EPL; ++j) {
            const int e = lane + 64 * j;
            if (e < D) { dw_part[(size_t)g * D + e] = dwa[j]; db_part[(size_t)g * D + e] = dba[j]; }
        }
    }
}

// The input-gradient-only backward (no dw / db: the text tower's LayerNorms are frozen) for D % 8 == 0, D <= 512: one wave per
// row, a lane owns EIGHT consecutive columns -- two 16-byte loads per tensor, and the row of dx that the result is added to is
// requested together with dy and xs instead of after the reduction (the scalar kernel's third dependent round trip); the bf16
// operand copy leaves as one 16-byte store per lane.  8.5 -> ~5 us for 817 x 512 (24 launches per step of the prompt chain).
// S > 0 (round 5): dy is the sum of S fp32 slices [S, M, D] (a split-K dX GEMM's partial products), added in slice order.
__global__ __launch_bounds__(256) void ln_bwd_dx8_kernel(const float *__restrict__ dy, const float *__restrict__ xs,
                                                         const float *__restrict__ w, const float *__restrict__ mean,
                                                         const float *__restrict__ rstd, float *__restrict__ dx, int accumulate,
                                                         void *__restrict__ dx_copy, int copy_dtype, int M, int D, int prio, int S = 0)
{
    PPT_PRIO(prio);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bool act = lane * 8 < D;
    const size_t off = (size_t)row * D + lane * 8;
    float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), d1 = d0, x0 = d0, x1 = d0, a0 = d0, a1 = d0, w0 = d0, w1 = d0;
    if (act) {
        d0 = *reinterpret_cast<const float4 *>(dy + off); d1 = *reinterpret_cast<const float4 *>(dy + off + 4);
        for (int z = 1; z < S; ++z) {
            const float *pr = dy + (size_t)z * M * D + off;
            const float4 e0 = *reinterpret_cast<const float4 *>(pr), e1 = *reinterpret_cast<const float4 *>(pr + 4);
            d0.x += e0.x; d0.y += e0.y; d0.z += e0.z; d0.w += e0.w; d1.x += e1.x; d1.y += e1.y; d1.z += e1.z; d1.w += e1.w;
        }
        x0 = *reinterpret_cast<const float4 *>(xs + off); x1 = *reinterpret_cast<const float4 *>(xs + off + 4);
        w0 = *reinterpret_cast<const float4 *>(w + lane * 8); w1 = *reinterpret_cast<const float4 *>(w + lane * 8 + 4);
        if (accumulate) { a0 = *reinterpret_cast<const float4 *>(dx + off); a1 = *reinterpret_cast<const float4 *>(dx + off + 4); }
    }
    const float mu = mean[row], rs = rstd[row];
    const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w}, xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    float gv[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xh[j] = act ? (xv[j] - mu) * rs : 0.f;
        gv[j] = dv[j] * wv[j];
        s1 += gv[j]; s2 += gv[j] * xh[j];
    }
    s1 = wave_reduce_sum(s1) / (float)D;
    s2 = wave_reduce_sum(s2) / (float)D;
    if (!act) return;
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = av[j] + rs * (gv[j] - s1 - xh[j] * s2);
    *reinterpret_cast<float4 *>(dx + off) = make_float4(t[0], t[1], t[2], t[3]);
    *reinterpret_cast<float4 *>(dx + off + 4) = make_float4(t[4], t[5], t[6], t[7]);
    if (dx_copy) {
        if (copy_dtype != PPT_F32) {
            *reinterpret_cast<uint4 *>((uint16_t *)dx_copy + off) = make_uint4(pack2_dt(copy_dtype, t[0], t[1]), pack2_dt(copy_dtype, t[2], t[3]),
                                                                               pack2_dt(copy_dtype, t[4], t[5]), pack2_dt(copy_dtype, t[6], t[7]));
        } else {
            *reinterpret_cast<float4 *>((float *)dx_copy + off) = make_float4(t[0], t[1], t[2], t[3]);
            *reinterpret_cast<float4 *>((float *)dx_copy + off + 4) = make_float4(t[4], t[5], t[6], t[7]);
        }
    }
}

// The backward WITH weight / bias gradients (trainable LayerNorms: the un-frozen last PointBERT block, point_encoder.py:70-79) in
// the 16-byte form: wave g walks rows [g * rpw, (g + 1) * rpw), a lane owns EIGHT consecutive columns (two float4 per tensor and
// row), the NEXT row's dy / xs / dx are requested before this row's two wave reductions (ln_bwd_kernel: one 4-byte element per
// lane and load, three dependent round trips per row -- 92 us for 32 832 x 384 at C3, 0.55 TB/s), and dw / db accumulate per
// column in the lane's registers in row order -- the partial rows ln_bwd_kernel writes, bit for bit; dx sums its two row
// statistics in ln_bwd_dx8_kernel's order.
__global__ __launch_bounds__(256) void ln_bwd_w8_kernel(const float *__restrict__ dy, const float *__restrict__ xs,
                                                        const float *__restrict__ w, const float *__restrict__ mean,
                                                        const float *__restrict__ rstd, float *__restrict__ dx, int accumulate,
                                                        void *__restrict__ dx_copy, int copy_dtype, float *__restrict__ dw_part,
                                                        float *__restrict__ db_part, int rpw, int M, int D, int prio)
{
    PPT_PRIO(prio);
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = g * rpw, r1 = min(M, r0 + rpw);
    if (r0 >= M) return;
    const bool act = lane * 8 < D;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w0 = z4, w1 = z4;
    if (act) { w0 = *reinterpret_cast<const float4 *>(w + lane * 8); w1 = *reinterpret_cast<const float4 *>(w + lane * 8 + 4); }
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float dwa[8], dba[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { dwa[j] = 0.f; dba[j] = 0.f; }
    float4 d0 = z4, d1 = z4, x0 = z4, x1 = z4, a0 = z4, a1 = z4;
    auto fetch = [&](int row, float4 &D0, float4 &D1, float4 &X0, float4 &X1, float4 &A0, float4 &A1) {
        if (act) {
            const size_t off = (size_t)row * D + lane * 8;
            D0 = *reinterpret_cast<const float4 *>(dy + off); D1 = *reinterpret_cast<const float4 *>(dy + off + 4);
            X0 = *reinterpret_cast<const float4 *>(xs + off); X1 = *reinterpret_cast<const float4 *>(xs + off + 4);
            if (accumulate) { A0 = *reinterpret_cast<const float4 *>(dx + off); A1 = *reinterpret_cast<const float4 *>(dx + off + 4); }
        }
    };
    fetch(r0, d0, d1, x0, x1, a0, a1);
    for (int row = r0; row < r1; ++row) {
        float4 nd0 = z4, nd1 = z4, nx0 = z4, nx1 = z4, na0 = z4, na1 = z4;
        if (row + 1 < r1) fetch(row + 1, nd0, nd1, nx0, nx1, na0, na1);       // in flight during this row's reductions
        const float mu = mean[row], rs = rstd[row];
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w}, xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        float gv[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xh[j] = act ? (xv[j] - mu) * rs : 0.f;
            dwa[j] += dv[j] * xh[j]; dba[j] += dv[j];
            gv[j] = dv[j] * wv[j];
            s1 += gv[j]; s2 += gv[j] * xh[j];
        }
        s1 = wave_reduce_sum(s1) / (float)D;
        s2 = wave_reduce_sum(s2) / (float)D;
        if (act) {
            const size_t off = (size_t)row * D + lane * 8;
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = av[j] + rs * (gv[j] - s1 - xh[j] * s2);
            *reinterpret_cast<float4 *>(dx + off) = make_float4(t[0], t[1], t[2], t[3]);
            *reinterpret_cast<float4 *>(dx + off + 4) = make_float4(t[4], t[5], t[6], t[7]);
            if (dx_copy) {
                if (copy_dtype != PPT_F32) {
                    *reinterpret_cast<uint4 *>((uint16_t *)dx_copy + off) = make_uint4(pack2_dt(copy_dtype, t[0], t[1]), pack2_dt(copy_dtype, t[2], t[3]),
                                                                                       pack2_dt(copy_dtype, t[4], t[5]), pack2_dt(copy_dtype, t[6], t[7]));
                } else {
                    *reinterpret_cast<float4 *>((float *)dx_copy + off) = make_float4(t[0], t[1], t[2], t[3]);
                    *reinterpret_cast<float4 *>((float *)dx_copy + off + 4) = make_float4(t[4], t[5], t[6], t[7]);
                }
            }
        }
        d0 = nd0; d1 = nd1; x0 = nx0; x1 = nx1; a0 = na0; a1 = na1;
    }
    if (act) {
        float *pw = dw_part + (size_t)g * D + lane * 8, *pb = db_part + (size_t)g * D + lane * 8;
        *reinterpret_cast<float4 *>(pw) = make_float4(dwa[0], dwa[1], dwa[2], dwa[3]);
        *reinterpret_cast<float4 *>(pw + 4) = make_float4(dwa[4], dwa[5], dwa[6], dwa[7]);
        *reinterpret_cast<float4 *>(pb) = make_float4(dba[0], dba[1], dba[2], dba[3]);
        *reinterpret_cast<float4 *>(pb + 4) = make_float4(dba[4], dba[5], dba[6], dba[7]);
    }
}

}  // namespace

extern "C" int ppt_layernorm_fwd(const float *x, const float *add, int add_rows, float *xs, const float *w,
                                 const float *b, void *y, int y_dtype, float *mean, float *rstd, int M, int D,
                                 float eps, void *stream)
{
    if (!x || !w || !b || !y || M <= 0 || D <= 0 || D > 64 * MAX_EPL) return PPT_EINVAL;
    if (y_dtype != PPT_BF16 && y_dtype != PPT_F32 && y_dtype != PPT_F16) return PPT_EINVAL;
    const bool al = (((uintptr_t)x | (uintptr_t)w | (uintptr_t)b | (uintptr_t)y | (uintptr_t)add | (uintptr_t)xs) & 15) == 0;
    if ((D % 8) == 0 && D <= 512 && al) {
        dim3 vgrid(min((M + 3) / 4, 256 * 8));         // 8 workgroups per CU; each wave walks rows with that stride
        if (y_dtype == PPT_BF16)
            hipLaunchKernelGGL(ln_fwd_vec8_kernel<bf16_t>, vgrid, dim3(256), 0, ppt_stream(stream), x, add, add_rows, xs, w, b,
                               (bf16_t *)y, mean, rstd, M, D, eps, ppt_get_wave_priority());
        else if (y_dtype == PPT_F16)
            hipLaunchKernelGGL(ln_fwd_vec8_kernel<f16_t>, vgrid, dim3(256), 0, ppt_stream(stream), x, add, add_rows, xs, w, b,
                               (f16_t *)y, mean, rstd, M, D, eps, ppt_get_wave_priority());
        else
            hipLaunchKernelGGL(ln_fwd_vec8_kernel<float>, vgrid, dim3(256), 0, ppt_stream(stream), x, add, add_rows, xs, w, b,
                               (float *)y, mean, rstd, M, D, eps, ppt_get_wave_priority());
        PPT_CHECK_LAUNCH();
        return PPT_OK;
    }
    dim3 grid((M + 3) / 4);
    if (y_dtype == PPT_BF16)
        hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, grid, dim3(256), 0, ppt_stream(stream), x, add, add_rows, xs, w, b,
                           (bf16_t *)y, mean, rstd, M, D, eps);
    else if (y_dtype == PPT_F16)
        hipLaunchKernelGGL(ln_fwd_kernel<f16_t>, grid, dim3(256), 0, ppt_stream(stream), x, add, add_rows, xs, w, b,
                           (f16_t *)y, mean, rstd, M, D, eps);
    else if (y_dtype == PPT_F32)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, dim3(256), 0, ppt_stream(stream), x, add, add_rows, xs, w, b,
                           (float *)y, mean, rstd, M, D, eps);
    else
        return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

extern "C" int ppt_layernorm_bwd(const float *dy, const float *xs, const float *w, const float *mean,
                                 const float *rstd, float *dx, int accumulate_dx, void *dx_copy, int dx_copy_dtype,
                                 float *dw_partial, float *db_partial, int partial_rows, int M, int D, void *stream)
{
    if (!dy || !xs || !w || !mean || !rstd || !dx || M <= 0 || D <= 0 || D > 64 * MAX_EPL) return PPT_EINVAL;
    if ((dw_partial == nullptr) != (db_partial == nullptr)) return PPT_EINVAL;
    if (dw_partial && partial_rows <= 0) return PPT_EINVAL;
    if (!dw_partial && D % 8 == 0 && D <= 512 && ((((uintptr_t)dy | (uintptr_t)xs | (uintptr_t)w | (uintptr_t)dx | (uintptr_t)dx_copy) & 15) == 0)) {
        hipLaunchKernelGGL(ln_bwd_dx8_kernel, dim3((M + 3) / 4), dim3(256), 0, ppt_stream(stream), dy, xs, w, mean, rstd, dx, accumulate_dx,
                           dx_copy, dx_copy_dtype, M, D, ppt_get_wave_priority());
        PPT_CHECK_LAUNCH();
        return PPT_OK;
    }
    const int waves = dw_partial ? partial_rows : M;
    const int rpw = (M + waves - 1) / waves;
    // every partial row must be written (rows past the data get zeros from an empty range): the
    // kernel returns early for empty ranges, so clear the tail partial rows here.
    const int used = (M + rpw - 1) / rpw;
    if (dw_partial && used < partial_rows) {
        (void)hipMemsetAsync(dw_partial + (size_t)used * D, 0, sizeof(float) * (size_t)(partial_rows - used) * D, ppt_stream(stream));
        (void)hipMemsetAsync(db_partial + (size_t)used * D, 0, sizeof(float) * (size_t)(partial_rows - used) * D, ppt_stream(stream));
    }
    static const bool w8 = getenv("PPT_LN_BWD_W8") == nullptr || atoi(getenv("PPT_LN_BWD_W8")) != 0;
    if (w8 && dw_partial && D % 8 == 0 && D <= 512 &&
        ((((uintptr_t)dy | (uintptr_t)xs | (uintptr_t)w | (uintptr_t)dx | (uintptr_t)dx_copy | (uintptr_t)dw_partial | (uintptr_t)db_partial) & 15) == 0)) {
        hipLaunchKernelGGL(ln_bwd_w8_kernel, dim3((used + 3) / 4), dim3(256), 0, ppt_stream(stream), dy, xs, w, mean, rstd, dx,
                           accumulate_dx, dx_copy, dx_copy_dtype, dw_partial, db_partial, rpw, M, D, ppt_get_wave_priority());
        PPT_CHECK_LAUNCH();
        return PPT_OK;
    }
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((used + 3) / 4), dim3(256), 0, ppt_stream(stream), dy, xs, w, mean, rstd, dx,
                       accumulate_dx, dx_copy, dx_copy_dtype, dw_partial, db_partial, rpw, M, D, ppt_get_wave_priority());
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// LayerNorm of x + bias + parts[0] + ... + parts[S-1] (split-K consumer, see ln_fwd_vec8_kernel<.., true>); xs (required) receives
// the summed row.  D % 8 == 0, D <= 512, 16-byte aligned operands.
extern "C" int ppt_layernorm_fwd_sum(const float *x, const float *bias, const float *parts, int S, float *xs, const float *w, const float *b,
                                     void *y, int y_dtype, float *mean, float *rstd, int M, int D, float eps, void *stream)
{
    if (!x || !parts || !xs || !w || !b || !y || S <= 0 || S > 8 || M <= 0 || D <= 0) return PPT_EINVAL;
    if ((D % 8) != 0 || D > 512) return PPT_EUNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)b | (uintptr_t)y | (uintptr_t)parts | (uintptr_t)xs | (uintptr_t)bias) & 15) != 0) return PPT_EINVAL;
    dim3 vgrid(min((M + 3) / 4, 256 * 8));
    if (y_dtype == PPT_BF16)
        hipLaunchKernelGGL((ln_fwd_vec8_kernel<bf16_t, true>), vgrid, dim3(256), 0, ppt_stream(stream), x, (const float *)nullptr, 0, xs, w, b,
                           (bf16_t *)y, mean, rstd, M, D, eps, ppt_get_wave_priority(), bias, parts, S);
    else if (y_dtype == PPT_F16)
        hipLaunchKernelGGL((ln_fwd_vec8_kernel<f16_t, true>), vgrid, dim3(256), 0, ppt_stream(stream), x, (const float *)nullptr, 0, xs, w, b,
                           (f16_t *)y, mean, rstd, M, D, eps, ppt_get_wave_priority(), bias, parts, S);
    else if (y_dtype == PPT_F32)
        hipLaunchKernelGGL((ln_fwd_vec8_kernel<float, true>), vgrid, dim3(256), 0, ppt_stream(stream), x, (const float *)nullptr, 0, xs, w, b,
                           (float *)y, mean, rstd, M, D, eps, ppt_get_wave_priority(), bias, parts, S);
    else
        return PPT_EINVAL;
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}

// ppt_layernorm_bwd's input-gradient form with dy = dy_parts[0] + ... + dy_parts[S-1] ([S, M, D] fp32, summed in that order).
extern "C" int ppt_layernorm_bwd_sum(const float *dy_parts, int S, const float *xs, const float *w, const float *mean, const float *rstd,
                                     float *dx, int accumulate_dx, void *dx_copy, int dx_copy_dtype, int M, int D, void *stream)
{
    if (!dy_parts || !xs || !w || !mean || !rstd || !dx || S <= 0 || S > 8 || M <= 0 || D <= 0) return PPT_EINVAL;
    if ((D % 8) != 0 || D > 512) return PPT_EUNSUPPORTED;
    if ((((uintptr_t)dy_parts | (uintptr_t)xs | (uintptr_t)w | (uintptr_t)dx | (uintptr_t)dx_copy) & 15) != 0) return PPT_EINVAL;
    hipLaunchKernelGGL(ln_bwd_dx8_kernel, dim3((M + 3) / 4), dim3(256), 0, ppt_stream(stream), dy_parts, xs, w, mean, rstd, dx, accumulate_dx,
                       dx_copy, dx_copy_dtype, M, D, ppt_get_wave_priority(), S);
    PPT_CHECK_LAUNCH();
    return PPT_OK;
}
