// Fused multi-head attention, forward and backward, for the reference's
// AttentionLayer core (models/layers.py:433-474): split heads, QK^T / sqrt(d),
// causal mask (additive 2e9 then -inf == plain causal for finite logits), softmax,
// PV, merge heads.  q/k/v/o stay in the (N, S, H*d) layout the surrounding Linear
// layers produce -- no head permute, no (N,H,Sq,Sk) score tensor in HBM.
//
// The reference runs 64 heads on a 512-wide model: head dim 8.  That is too thin for
// MFMA (K=8 / N=8 tiles) and only ~2 % of the layer's FLOPs, so the kernel is a VALU
// one shaped for the wave: one lane per query row (q, o, m, l in registers), key/value
// rows are wave-uniform so they come through the scalar cache (s_load) and feed
// v_fma directly; online softmax over chunks of 8 keys; nothing crosses lanes.
// Backward = two such passes (lane per query for dQ, lane per key for dK/dV), scores
// recomputed from the saved log-sum-exp: deterministic, no atomics.
//
// Softmax runs in base 2: scores are scaled by c = log2(e)/sqrt(d) in one multiply and
// exponentiated with v_exp_f32 (2^x).  Against exp((q.k)/sqrt(d) - m) this perturbs each
// probability by <= ~1e-7 absolute (the product rounding is |x| * 2^-24 in the exponent
// and terms with large |x| are themselves tiny), two orders inside the 1e-5 tolerance
// stated for attention tensors; the saved LSE is kept in base-2 units (internal).
#include "qarig_common.h"

namespace qarig {

constexpr int KC = 8;  // keys per online-softmax chunk

struct AttnDims {
    int N, Sq, Sk, H, causal;
    float c2;   // log2(e) / sqrt(d)
    float rsd;  // 1 / sqrt(d)
};

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q,
                                                       const float* __restrict__ k,
                                                       const float* __restrict__ v, AttnDims a,
                                                       float* __restrict__ o,
                                                       float* __restrict__ lse) {
    const int lane = threadIdx.x & 63;
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int qtiles = (a.Sq + 63) >> 6;
    if (task >= a.N * a.H * qtiles) return;
    const int qt = task % qtiles, nh = task / qtiles;
    const int h = nh % a.H, n = nh / a.H;
    const int D = a.H * HD;
    const int i = qt * 64 + lane;
    const bool active = i < a.Sq;

    float qv[HD], ov[HD];
    const float* qp = q + ((int64_t)n * a.Sq + (active ? i : 0)) * D + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        qv[c] = active ? qp[c] : 0.0f;
        ov[c] = 0.0f;
    }
    float m = -INFINITY, l = 0.0f;
    const float* kb = k + (int64_t)n * a.Sk * D + h * HD;
    const float* vb = v + (int64_t)n * a.Sk * D + h * HD;
    const int jend = a.causal ? min(a.Sk, qt * 64 + 64) : a.Sk;

    for (int j0 = 0; j0 < jend; j0 += KC) {
        float s[KC];
        float mc = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) {
            const int j = j0 + jj;
            float t = -INFINITY;
            if (j < jend) {
                const float* kr = kb + (int64_t)j * D;
                float dot = 0.0f;
#pragma unroll
                for (int c = 0; c < HD; ++c) dot = fmaf(qv[c], kr[c], dot);
                t = dot * a.c2;
                if (a.causal && j > i) t = -INFINITY;
            }
            s[jj] = t;
            mc = fmaxf(mc, t);
        }
        const float mn = fmaxf(m, mc);
        if (mn == -INFINITY) continue;  // only lanes whose every key so far is masked
        const float alpha = exp2_fast(m - mn);
        l *= alpha;
#pragma unroll
        for (int c = 0; c < HD; ++c) ov[c] *= alpha;
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) {
            const int j = j0 + jj;
            if (j < jend) {
                const float p = exp2_fast(s[jj] - mn);
                l += p;
                const float* vr = vb + (int64_t)j * D;
#pragma unroll
                for (int c = 0; c < HD; ++c) ov[c] = fmaf(p, vr[c], ov[c]);
            }
        }
        m = mn;
    }
    if (active) {
        float* op = o + ((int64_t)n * a.Sq + i) * D + h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) op[c] = ov[c] / l;
        lse[((int64_t)n * a.H + h) * a.Sq + i] = m + log2f(l);   // base-2 units
    }
}

// dQ pass: lane per query.  Also emits delta[i] = sum_c dO[i][c]*O[i][c].
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ o, const float* __restrict__ dO, const float* __restrict__ lse,
    AttnDims a, float* __restrict__ dq, float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int qtiles = (a.Sq + 63) >> 6;
    if (task >= a.N * a.H * qtiles) return;
    const int qt = task % qtiles, nh = task / qtiles;
    const int h = nh % a.H, n = nh / a.H;
    const int D = a.H * HD;
    const int i = qt * 64 + lane;
    const bool active = i < a.Sq;
    const int64_t roff = ((int64_t)n * a.Sq + (active ? i : 0)) * D + h * HD;

    float qv[HD], dov[HD], acc[HD];
    float dl = 0.0f;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        qv[c] = active ? q[roff + c] : 0.0f;
        dov[c] = active ? dO[roff + c] : 0.0f;
        const float oc = active ? o[roff + c] : 0.0f;
        dl = fmaf(dov[c], oc, dl);
        acc[c] = 0.0f;
    }
    const float L = active ? lse[((int64_t)n * a.H + h) * a.Sq + i] : 0.0f;
    const float* kb = k + (int64_t)n * a.Sk * D + h * HD;
    const float* vb = v + (int64_t)n * a.Sk * D + h * HD;
    const int jend = a.causal ? min(a.Sk, qt * 64 + 64) : a.Sk;
    for (int j = 0; j < jend; ++j) {
        const float* kr = kb + (int64_t)j * D;
        const float* vr = vb + (int64_t)j * D;
        float dot = 0.0f, dp = 0.0f;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            dot = fmaf(qv[c], kr[c], dot);
            dp = fmaf(dov[c], vr[c], dp);
        }
        float p = exp2_fast(fmaf(dot, a.c2, -L));
        if (a.causal && j > i) p = 0.0f;
        const float ds = p * (dp - dl);
#pragma unroll
        for (int c = 0; c < HD; ++c) acc[c] = fmaf(ds, kr[c], acc[c]);
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[roff + c] = acc[c] * a.rsd;
        delta[((int64_t)n * a.H + h) * a.Sq + i] = dl;
    }
}

// dK/dV pass: lane per key; query rows are wave-uniform.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ delta,
    AttnDims a, float* __restrict__ dk, float* __restrict__ dv) {
    const int lane = threadIdx.x & 63;
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int ktiles = (a.Sk + 63) >> 6;
    if (task >= a.N * a.H * ktiles) return;
    const int kt = task % ktiles, nh = task / ktiles;
    const int h = nh % a.H, n = nh / a.H;
    const int D = a.H * HD;
    const int j = kt * 64 + lane;
    const bool active = j < a.Sk;
    const int64_t roff = ((int64_t)n * a.Sk + (active ? j : 0)) * D + h * HD;

    float kv[HD], vv[HD], dka[HD], dva[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        kv[c] = active ? k[roff + c] : 0.0f;
        vv[c] = active ? v[roff + c] : 0.0f;
        dka[c] = 0.0f;
        dva[c] = 0.0f;
    }
    const float* qb = q + (int64_t)n * a.Sq * D + h * HD;
    const float* dob = dO + (int64_t)n * a.Sq * D + h * HD;
    const float* lb = lse + ((int64_t)n * a.H + h) * a.Sq;
    const float* db = delta + ((int64_t)n * a.H + h) * a.Sq;
    const int ibeg = a.causal ? kt * 64 : 0;
    for (int i = ibeg; i < a.Sq; ++i) {
        const float* qr = qb + (int64_t)i * D;
        const float* dor = dob + (int64_t)i * D;
        float dot = 0.0f, dp = 0.0f;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            dot = fmaf(qr[c], kv[c], dot);
            dp = fmaf(dor[c], vv[c], dp);
        }
        float p = exp2_fast(fmaf(dot, a.c2, -lb[i]));
        if (a.causal && j > i) p = 0.0f;
        const float ds = p * (dp - db[i]);
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            dva[c] = fmaf(p, dor[c], dva[c]);
            dka[c] = fmaf(ds, qr[c], dka[c]);
        }
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            dk[roff + c] = dka[c] * a.rsd;
            dv[roff + c] = dva[c];
        }
    }
}

}  // namespace qarig

using namespace qarig;

#define QARIG_HD_DISPATCH(d, CALL)                                               \
    switch (d) {                                                                 \
        case 4: { constexpr int HD = 4; CALL; } break;                           \
        case 8: { constexpr int HD = 8; CALL; } break;                           \
        case 16: { constexpr int HD = 16; CALL; } break;                         \
        case 32: { constexpr int HD = 32; CALL; } break;                         \
        case 64: { constexpr int HD = 64; CALL; } break;                         \
        default:                                                                 \
            qarig_set_error("attention: head dim %d unsupported (4,8,16,32,64)", d); \
            return QARIG_ERR_ARG;                                                \
    }

static int attn_check(int N, int Sq, int Sk, int H, int d, int causal) {
    QARIG_CHECK_ARG(N > 0 && Sq > 0 && Sk > 0 && H > 0 && d > 0, "attention: bad extents");
    QARIG_CHECK_ARG(!causal || Sq == Sk, "attention: causal needs Sq == Sk (self-attention)");
    return QARIG_OK;
}

// q: (N,Sq,H*d); k,v: (N,Sk,H*d); o: (N,Sq,H*d); lse: (N,H,Sq).  sqrt_d is passed by
// the host as float(d ** 0.5), the divisor the reference uses (layers.py:446).
extern "C" int qarig_attention_fwd(const float* q, const float* k, const float* v, int N, int Sq,
                                   int Sk, int H, int d, int causal, float sqrt_d, float* o,
                                   float* lse, void* stream) {
    QARIG_CHECK_ARG(q && k && v && o && lse, "attention_fwd: null pointer");
    if (int e = attn_check(N, Sq, Sk, H, d, causal)) return e;
    AttnDims a{N, Sq, Sk, H, causal, 1.4426950408889634f / sqrt_d, 1.0f / sqrt_d};
    const int tasks = N * H * ((Sq + 63) / 64);
    dim3 grid((tasks + 3) / 4), block(256);
    QARIG_HD_DISPATCH(d, hipLaunchKernelGGL((attn_fwd_kernel<HD>), grid, block, 0,
                                            (hipStream_t)stream, q, k, v, a, o, lse));
    QARIG_CHECK_LAUNCH("attention_fwd");
    return QARIG_OK;
}

// delta: caller-provided (N,H,Sq) scratch.
extern "C" int qarig_attention_bwd(const float* q, const float* k, const float* v, const float* o,
                                   const float* dO, const float* lse, int N, int Sq, int Sk, int H,
                                   int d, int causal, float sqrt_d, float* dq, float* dk, float* dv,
                                   float* delta, void* stream) {
    QARIG_CHECK_ARG(q && k && v && o && dO && lse && dq && dk && dv && delta,
                    "attention_bwd: null pointer");
    if (int e = attn_check(N, Sq, Sk, H, d, causal)) return e;
    AttnDims a{N, Sq, Sk, H, causal, 1.4426950408889634f / sqrt_d, 1.0f / sqrt_d};
    const int qtasks = N * H * ((Sq + 63) / 64);
    const int ktasks = N * H * ((Sk + 63) / 64);
    dim3 block(256);
    QARIG_HD_DISPATCH(d, hipLaunchKernelGGL((attn_bwd_dq_kernel<HD>), dim3((qtasks + 3) / 4), block,
                                            0, (hipStream_t)stream, q, k, v, o, dO, lse, a, dq,
                                            delta));
    QARIG_CHECK_LAUNCH("attention_bwd dq");
    QARIG_HD_DISPATCH(d, hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD>), dim3((ktasks + 3) / 4), block,
                                            0, (hipStream_t)stream, q, k, v, dO, lse, delta, a, dk,
                                            dv));
    QARIG_CHECK_LAUNCH("attention_bwd dkv");
    return QARIG_OK;
}
