// LayerNorm over the last dimension, one wave per row, with the two affine forms the
// Transformer uses fused in:
//   - nn.LayerNorm(D) with gamma/beta          (encoder blocks, layers.py:327,499,559)
//   - AdaLNZero: scale(cond)*LN(x)+shift(cond) (decoder blocks, layers.py:130-153),
//     scale/shift being per-token (M,D) tensors produced by the cond GEMMs.
// eps = 1e-5, biased variance, fp32 throughout (torch.nn.LayerNorm semantics).
#include "qarig_common.h"

namespace qarig {

constexpr int LN_WAVES = 4;

// NV > 0: D == NV * 256 and every pointer 16-B aligned (the Transformer widths); NV == 0:
// any D, scalar three-pass form.
template <int NV>
__global__ __launch_bounds__(LN_WAVES * 64) void layernorm_fwd_kernel(
    const float* __restrict__ x, int M, int D, float eps, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ scale,
    const float* __restrict__ shift, const int* __restrict__ mod_idx, float* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (int64_t)row * D;
    // scale/shift row: the token's own, or (position-table form) the row its index names
    const int64_t mrow = (int64_t)(mod_idx ? mod_idx[row] : row) * D;
    if (NV > 0) {
        // the row lives in registers: one 16-B load per 256 columns per lane, no re-reads
        float4 r[NV > 0 ? NV : 1];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            r[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
            s += (r[i].x + r[i].y) + (r[i].z + r[i].w);
        }
        const float mean = wave_sum(s) / (float)D;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            r[i].x -= mean; r[i].y -= mean; r[i].z -= mean; r[i].w -= mean;
            v = fmaf(r[i].x, r[i].x, v); v = fmaf(r[i].y, r[i].y, v);
            v = fmaf(r[i].z, r[i].z, v); v = fmaf(r[i].w, r[i].w, v);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
        float* yr = y + (int64_t)row * D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            float4 h = make_float4(r[i].x * rstd, r[i].y * rstd, r[i].z * rstd, r[i].w * rstd);
            if (gamma) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + c);
                const float4 b = *reinterpret_cast<const float4*>(beta + c);
                h = make_float4(h.x * g.x + b.x, h.y * g.y + b.y, h.z * g.z + b.z, h.w * g.w + b.w);
            } else if (scale) {
                const float4 g = *reinterpret_cast<const float4*>(scale + mrow + c);
                const float4 b = *reinterpret_cast<const float4*>(shift + mrow + c);
                h = make_float4(g.x * h.x + b.x, g.y * h.y + b.y, g.z * h.z + b.z, g.w * h.w + b.w);
            }
            *reinterpret_cast<float4*>(yr + c) = h;
        }
        return;
    }
    float s = 0.0f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)D;
    float v = 0.0f;
    for (int c = lane; c < D; c += 64) {
        const float t = xr[c] - mean;
        v = fmaf(t, t, v);
    }
    const float var = wave_sum(v) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
    float* yr = y + (int64_t)row * D;
    for (int c = lane; c < D; c += 64) {
        float h = (xr[c] - mean) * rstd;
        if (gamma) h = h * gamma[c] + beta[c];
        else if (scale) h = scale[mrow + c] * h + shift[mrow + c];
        yr[c] = h;
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * (gamma | scale | 1)
// dy_xhat (optional) = dy * xhat: AdaLN's d(scale), or the column-summed d(gamma).
// NV as in the forward kernel: D == NV * 256, 16-B aligned: x and dy are read once into registers.
template <int NV>
__global__ __launch_bounds__(LN_WAVES * 64) void layernorm_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, const float* __restrict__ gamma,
    const float* __restrict__ scale, const int* __restrict__ mod_idx, int M, int D,
    float* __restrict__ dx, float* __restrict__ dy_xhat, const float* __restrict__ dx_add) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t off = (int64_t)row * D;
    const int64_t moff = (int64_t)(mod_idx ? mod_idx[row] : row) * D;
    const float mean = mean_in[row], rstd = rstd_in[row];
    if (NV > 0) {
        float4 h[NV > 0 ? NV : 1], d[NV > 0 ? NV : 1], g[NV > 0 ? NV : 1];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            const float4 xv = *reinterpret_cast<const float4*>(x + off + c);
            d[i] = *reinterpret_cast<const float4*>(dy + off + c);
            h[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd,
                               (xv.w - mean) * rstd);
            float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
            if (gamma) m = *reinterpret_cast<const float4*>(gamma + c);
            else if (scale) m = *reinterpret_cast<const float4*>(scale + moff + c);
            g[i] = (gamma || scale) ? make_float4(d[i].x * m.x, d[i].y * m.y, d[i].z * m.z, d[i].w * m.w)
                                    : d[i];
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 = fmaf(g[i].x, h[i].x, s2); s2 = fmaf(g[i].y, h[i].y, s2);
            s2 = fmaf(g[i].z, h[i].z, s2); s2 = fmaf(g[i].w, h[i].w, s2);
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            float4 r = make_float4(rstd * ((g[i].x - s1) - h[i].x * s2), rstd * ((g[i].y - s1) - h[i].y * s2),
                                   rstd * ((g[i].z - s1) - h[i].z * s2), rstd * ((g[i].w - s1) - h[i].w * s2));
            if (dx_add) {   // the skip connection's gradient of the same x: autograd's own add, fused
                const float4 a = *reinterpret_cast<const float4*>(dx_add + off + c);
                r = make_float4(a.x + r.x, a.y + r.y, a.z + r.z, a.w + r.w);
            }
            *reinterpret_cast<float4*>(dx + off + c) = r;
            if (dy_xhat)
                *reinterpret_cast<float4*>(dy_xhat + off + c) =
                    make_float4(d[i].x * h[i].x, d[i].y * h[i].y, d[i].z * h[i].z, d[i].w * h[i].w);
        }
        return;
    }
    float s1 = 0.0f, s2 = 0.0f;
    for (int c = lane; c < D; c += 64) {
        const float h = (x[off + c] - mean) * rstd;
        const float d = dy[off + c];
        const float g = gamma ? d * gamma[c] : (scale ? d * scale[moff + c] : d);
        s1 += g;
        s2 = fmaf(g, h, s2);
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    for (int c = lane; c < D; c += 64) {
        const float h = (x[off + c] - mean) * rstd;
        const float d = dy[off + c];
        const float g = gamma ? d * gamma[c] : (scale ? d * scale[moff + c] : d);
        const float r = rstd * ((g - s1) - h * s2);
        dx[off + c] = dx_add ? dx_add[off + c] + r : r;
        if (dy_xhat) dy_xhat[off + c] = d * h;
    }
}

}  // namespace qarig

using namespace qarig;

extern "C" int qarig_layernorm_fwd(const float* x, int M, int D, float eps, const float* gamma,
                                   const float* beta, const float* scale, const float* shift,
                                   const int* mod_idx, float* y, float* mean, float* rstd,
                                   void* stream) {
    QARIG_CHECK_ARG(!mod_idx || scale, "layernorm_fwd: mod_idx needs scale/shift tables");
    QARIG_CHECK_ARG(x && y && mean && rstd && M > 0 && D > 0, "layernorm_fwd: bad arguments");
    QARIG_CHECK_DIMS("layernorm_fwd", M, D);
    QARIG_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "layernorm_fwd: gamma/beta pair");
    QARIG_CHECK_ARG((scale == nullptr) == (shift == nullptr), "layernorm_fwd: scale/shift pair");
    QARIG_CHECK_ARG(!(gamma && scale), "layernorm_fwd: affine and AdaLN forms are exclusive");
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool vec = D % 256 == 0 && al16(x) && al16(y) && al16(gamma) && al16(beta) &&
                     al16(scale) && al16(shift);
    const dim3 grid((M + LN_WAVES - 1) / LN_WAVES), block(LN_WAVES * 64);
#define QARIG_LN_LAUNCH(NV)                                                                      \
    hipLaunchKernelGGL((layernorm_fwd_kernel<NV>), grid, block, 0, (hipStream_t)stream, x, M, D, \
                       eps, gamma, beta, scale, shift, mod_idx, y, mean, rstd)
    switch (vec ? D / 256 : 0) {
        case 1: QARIG_LN_LAUNCH(1); break;
        case 2: QARIG_LN_LAUNCH(2); break;
        case 4: QARIG_LN_LAUNCH(4); break;
        case 8: QARIG_LN_LAUNCH(8); break;
        default: QARIG_LN_LAUNCH(0); break;
    }
#undef QARIG_LN_LAUNCH
    QARIG_CHECK_LAUNCH("layernorm_fwd");
    return QARIG_OK;
}

extern "C" int qarig_layernorm_bwd(const float* dy, const float* x, const float* mean,
                                   const float* rstd, const float* gamma, const float* scale,
                                   const int* mod_idx, int M, int D, float* dx, float* dy_xhat,
                                   const float* dx_add, void* stream) {
    QARIG_CHECK_ARG(dy && x && mean && rstd && dx && M > 0 && D > 0, "layernorm_bwd: bad arguments");
    QARIG_CHECK_DIMS("layernorm_bwd", M, D);
    QARIG_CHECK_ARG(!(gamma && scale), "layernorm_bwd: affine and AdaLN forms are exclusive");
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool vec = D % 256 == 0 && al16(dy) && al16(x) && al16(gamma) && al16(scale) && al16(dx) &&
                     al16(dy_xhat) && al16(dx_add);
    const dim3 grid((M + LN_WAVES - 1) / LN_WAVES), block(LN_WAVES * 64);
#define QARIG_LNB_LAUNCH(NV)                                                                       \
    hipLaunchKernelGGL((layernorm_bwd_kernel<NV>), grid, block, 0, (hipStream_t)stream, dy, x, mean, \
                       rstd, gamma, scale, mod_idx, M, D, dx, dy_xhat, dx_add)
    switch (vec ? D / 256 : 0) {
        case 1: QARIG_LNB_LAUNCH(1); break;
        case 2: QARIG_LNB_LAUNCH(2); break;
        case 4: QARIG_LNB_LAUNCH(4); break;
        default: QARIG_LNB_LAUNCH(0); break;
    }
#undef QARIG_LNB_LAUNCH
    QARIG_CHECK_LAUNCH("layernorm_bwd");
    return QARIG_OK;
}
