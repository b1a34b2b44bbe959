// Attention for head dims 65 ... 128 (the reference takes any head count that divides the model width,
// models/layers.py:433: a 512-wide model with 4 heads has 128-wide heads).  The MFMA kernels of attention.hip
// keep a query's q and o rows in registers next to their LDS fragments and stop at head dim 64; this family
// serves the rare wide-head model with the plainest exact-fp32 form instead of widening those tiles:
//   one lane = one query (forward, dQ) or one key (dV, dK); its own row(s) live in registers (128 floats
//   each), the rows of the other side are read at wave-uniform addresses (every lane reads the same row:
//   scalar / broadcast loads) and every product is a c-ascending fma chain on the vector ALU, keys (queries)
//   ascending -- the scalar loops the reference's einsum + softmax define, bit-reproducible, no atomics.
// Heads narrower than 128 arrive zero-padded (qarig.functional.attention); sqrt_d carries the model's own
// head dim.  softmax in base 2 like attention.hip; the saved LSE is in base-2 units and private to this
// family (forward and backward of one head dim always run on the same family).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qarig_common.h"

namespace qarig {

constexpr int WD = 128;

__device__ __forceinline__ float exp2w(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&r)[WD]) {
#pragma unroll
    for (int c = 0; c < WD; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        r[c] = v.x; r[c + 1] = v.y; r[c + 2] = v.z; r[c + 3] = v.w;
    }
}
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&r)[WD], float s) {
#pragma unroll
    for (int c = 0; c < WD; c += 4)
        *reinterpret_cast<float4*>(p + c) = make_float4(r[c] * s, r[c + 1] * s, r[c + 2] * s, r[c + 3] * s);
}
// sum_c a[c] * row[c], c ascending; `row` at a wave-uniform address
__device__ __forceinline__ float dot_row(const float (&a)[WD], const float* __restrict__ row) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < WD; ++c) s = fmaf(a[c], row[c], s);
    return s;
}

// grid (ceil(Sq / 64), H, N), 64 threads: lane = query
__global__ __launch_bounds__(64) void attn_wide_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, int Sq, int Sk, int H,
                                                           int causal, float c2, float* __restrict__ o,
                                                           float* __restrict__ lse) {
    const int h = blockIdx.y, n = blockIdx.z;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool live = i < Sq;
    const int64_t D = (int64_t)H * WD;
    float qr[WD], acc[WD];
    load_row(q + ((int64_t)n * Sq + (live ? i : Sq - 1)) * D + h * WD, qr);
#pragma unroll
    for (int c = 0; c < WD; ++c) acc[c] = 0.0f;
    float m = -INFINITY, l = 0.0f;
    const int jend = causal ? min(Sk, blockIdx.x * 64 + 64) : Sk;
    for (int j = 0; j < jend; ++j) {
        const float* kr = k + ((int64_t)n * Sk + j) * D + h * WD;
        const float* vr = v + ((int64_t)n * Sk + j) * D + h * WD;
        const float t = dot_row(qr, kr) * c2;
        if (!causal || j <= i) {
            const float mn = fmaxf(m, t);
            const float alpha = exp2w(m - mn), p = exp2w(t - mn);
            l = l * alpha + p;
#pragma unroll
            for (int c = 0; c < WD; ++c) acc[c] = fmaf(p, vr[c], acc[c] * alpha);
            m = mn;
        }
    }
    if (live) {
        store_row(o + ((int64_t)n * Sq + i) * D + h * WD, acc, 1.0f / l);
        lse[((int64_t)n * H + h) * Sq + i] = m + __builtin_amdgcn_logf(l);      // v_log_f32: log2
    }
}

// dQ and delta = rowsum(dO * O): lane = query
__global__ __launch_bounds__(64) void attn_wide_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ o,
                                                          const float* __restrict__ dO, const float* __restrict__ lse,
                                                          int Sq, int Sk, int H, int causal, float c2, float rsd,
                                                          float* __restrict__ dq, float* __restrict__ delta) {
    const int h = blockIdx.y, n = blockIdx.z;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const bool live = i < Sq;
    const int ic = live ? i : Sq - 1;
    const int64_t D = (int64_t)H * WD;
    const int64_t ro = ((int64_t)n * Sq + ic) * D + h * WD;
    float qr[WD], gr[WD], acc[WD];
    load_row(q + ro, qr);
    load_row(dO + ro, gr);
    float dl = 0.0f;
#pragma unroll
    for (int c = 0; c < WD; c += 4) {
        const float4 ov = *reinterpret_cast<const float4*>(o + ro + c);
        dl = fmaf(gr[c], ov.x, dl); dl = fmaf(gr[c + 1], ov.y, dl);
        dl = fmaf(gr[c + 2], ov.z, dl); dl = fmaf(gr[c + 3], ov.w, dl);
        acc[c] = acc[c + 1] = acc[c + 2] = acc[c + 3] = 0.0f;
    }
    const float L = lse[((int64_t)n * H + h) * Sq + ic];
    const int jend = causal ? min(Sk, blockIdx.x * 64 + 64) : Sk;
    for (int j = 0; j < jend; ++j) {
        const float* kr = k + ((int64_t)n * Sk + j) * D + h * WD;
        const float* vr = v + ((int64_t)n * Sk + j) * D + h * WD;
        const float p = exp2w(fmaf(dot_row(qr, kr), c2, -L));
        const float dp = dot_row(gr, vr);
        const float ds = (!causal || j <= i) ? p * (dp - dl) : 0.0f;
#pragma unroll
        for (int c = 0; c < WD; ++c) acc[c] = fmaf(ds, kr[c], acc[c]);
    }
    if (live) {
        store_row(dq + ro, acc, rsd);
        delta[((int64_t)n * H + h) * Sq + i] = dl;
    }
}

// dV (DK == false) or dK (DK == true): lane = key; queries ascending
template <bool DK>
__global__ __launch_bounds__(64) void attn_wide_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ dO,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta, int Sq, int Sk, int H,
                                                           int causal, float c2, float rsd, float* __restrict__ out) {
    const int h = blockIdx.y, n = blockIdx.z;
    const int j = blockIdx.x * 64 + threadIdx.x;
    const bool live = j < Sk;
    const int64_t D = (int64_t)H * WD;
    const int64_t ro = ((int64_t)n * Sk + (live ? j : Sk - 1)) * D + h * WD;
    float kr[WD], acc[WD];
    float vr[DK ? WD : 1];
    load_row(k + ro, kr);
    if constexpr (DK) load_row(v + ro, vr);
#pragma unroll
    for (int c = 0; c < WD; ++c) acc[c] = 0.0f;
    const int i0 = causal ? blockIdx.x * 64 : 0;      // queries in front of the tile's first key see none of its keys
    for (int i = i0; i < Sq; ++i) {
        const float* qr = q + ((int64_t)n * Sq + i) * D + h * WD;
        const float* gr = dO + ((int64_t)n * Sq + i) * D + h * WD;
        const float L = lse[((int64_t)n * H + h) * Sq + i];
        float p = exp2w(fmaf(dot_row(kr, qr), c2, -L));
        if (causal && j > i) p = 0.0f;
        if constexpr (DK) {
            const float ds = p * (dot_row(vr, gr) - delta[((int64_t)n * H + h) * Sq + i]);
#pragma unroll
            for (int c = 0; c < WD; ++c) acc[c] = fmaf(ds, qr[c], acc[c]);
        } else {
#pragma unroll
            for (int c = 0; c < WD; ++c) acc[c] = fmaf(p, gr[c], acc[c]);
        }
    }
    if (live) store_row(out + ro, acc, DK ? rsd : 1.0f);
}

}  // namespace qarig

using namespace qarig;

int qarig_attention_wide_fwd(const float* q, const float* k, const float* v, int N, int Sq, int Sk, int H,
                             int causal, float sqrt_d, float* o, float* lse, hipStream_t st) {
    const float c2 = 1.4426950408889634f / sqrt_d;
    hipLaunchKernelGGL(attn_wide_fwd_kernel, dim3((Sq + 63) / 64, H, N), dim3(64), 0, st, q, k, v, Sq, Sk, H, causal,
                       c2, o, lse);
    QARIG_CHECK_LAUNCH("attention (wide heads) forward");
    return QARIG_OK;
}

int qarig_attention_wide_bwd(const float* q, const float* k, const float* v, const float* o, const float* dO,
                             const float* lse, int N, int Sq, int Sk, int H, int causal, float sqrt_d, float* dq,
                             float* dk, float* dv, float* delta, hipStream_t st) {
    const float c2 = 1.4426950408889634f / sqrt_d, rsd = 1.0f / sqrt_d;
    hipLaunchKernelGGL(attn_wide_dq_kernel, dim3((Sq + 63) / 64, H, N), dim3(64), 0, st, q, k, v, o, dO, lse, Sq, Sk,
                       H, causal, c2, rsd, dq, delta);
    hipLaunchKernelGGL((attn_wide_dkv_kernel<false>), dim3((Sk + 63) / 64, H, N), dim3(64), 0, st, q, k, v, dO, lse,
                       delta, Sq, Sk, H, causal, c2, rsd, dv);
    hipLaunchKernelGGL((attn_wide_dkv_kernel<true>), dim3((Sk + 63) / 64, H, N), dim3(64), 0, st, q, k, v, dO, lse,
                       delta, Sq, Sk, H, causal, c2, rsd, dk);
    QARIG_CHECK_LAUNCH("attention (wide heads) backward");
    return QARIG_OK;
}
