// Fused multi-head attention, forward and backward, for the reference's
// AttentionLayer core (models/layers.py:433-474): split heads, QK^T / sqrt(d),
// causal mask (additive 2e9 then -inf == plain causal for finite logits), softmax,
// PV, merge heads.  q/k/v/o stay in the (N, S, H*d) layout the surrounding Linear
// layers produce -- no head permute, no (N,H,Sq,Sk) score tensor in HBM.
//
// The reference runs 64 heads on a 512-wide model: head dim 8.  That is too thin for
// MFMA (K=8 / N=8 tiles) and only ~2 % of the layer's FLOPs, so the kernel is a VALU
// one shaped for the wave: one lane per query row (q, o, m, l in registers); key/value
// rows are staged through LDS per block and read back as wave-wide broadcasts (same
// address in every lane: conflict-free), online softmax over chunks of 8 keys; nothing
// crosses lanes.  (A first version fed K/V through s_load: each key's scalar-load latency
// sat exposed in front of its 8 FMAs -- 4x slower than this one.)
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

// Cooperative stage of `rows` consecutive (n,h) rows [row0, row0+rows) of a (N,S,H*HD)
// tensor into LDS as dst[r][HD]; rows at or beyond `limit` are zero-filled.
template <int HD>
__device__ __forceinline__ void stage_rows(const float* __restrict__ base, int D, int row0,
                                           int rows, int limit, float* dst) {
    constexpr int V4 = HD / 4;
    for (int idx = threadIdx.x; idx < rows * V4; idx += 256) {
        const int r = idx / V4, c4 = idx - r * V4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < limit)
            v = *reinterpret_cast<const float4*>(base + (int64_t)(row0 + r) * D + c4 * 4);
        *reinterpret_cast<float4*>(dst + r * HD + c4 * 4) = v;
    }
}

// Forward.  Block = 256 queries of one (n,h) (lane per query); K/V rows are staged through
// LDS in chunks of KCH keys (8 KB each) and read back as wave-wide broadcasts, so the
// inner loop is pure VALU with its operand reads pipelined (no per-key scalar-load wait).
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q,
                                                       const float* __restrict__ k,
                                                       const float* __restrict__ v, AttnDims a,
                                                       float* __restrict__ o,
                                                       float* __restrict__ lse) {
    constexpr int KCH = 2048 / HD;
    __shared__ __attribute__((aligned(16))) float Ks[KCH * HD];
    __shared__ __attribute__((aligned(16))) float Vs[KCH * HD];
    const int qblocks = (a.Sq + 255) >> 8;
    const int qb = blockIdx.x % qblocks, nh = blockIdx.x / qblocks;
    const int h = nh % a.H, n = nh / a.H;
    const int D = a.H * HD;
    const int i = qb * 256 + threadIdx.x;
    const bool active = i < a.Sq;
    const int wave_last = __builtin_amdgcn_readfirstlane(qb * 256 + (threadIdx.x >> 6) * 64 + 63);

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
    const int jend_blk = a.causal ? min(a.Sk, qb * 256 + 256) : a.Sk;

    for (int c0 = 0; c0 < jend_blk; c0 += KCH) {
        __syncthreads();
        stage_rows<HD>(kb, D, c0, KCH, a.Sk, Ks);
        stage_rows<HD>(vb, D, c0, KCH, a.Sk, Vs);
        __syncthreads();
        const int jw = a.causal ? min(c0 + KCH, min(a.Sk, wave_last + 1)) : min(c0 + KCH, a.Sk);
        for (int j0 = c0; j0 < jw; j0 += KC) {
            float s[KC];
            float mc = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                const int j = j0 + jj;
                const float* kr = Ks + (j - c0) * HD;
                float dot = 0.0f;
#pragma unroll
                for (int c = 0; c < HD; ++c) dot = fmaf(qv[c], kr[c], dot);
                float t = dot * a.c2;
                if (j >= jw || (a.causal && j > i)) t = -INFINITY;
                s[jj] = t;
                mc = fmaxf(mc, t);
            }
            const float mn = fmaxf(m, mc);
            const float msafe = mn == -INFINITY ? 0.0f : mn;  // fully masked so far: p = 0
            const float alpha = exp2_fast(m - msafe);
            l *= alpha;
#pragma unroll
            for (int c = 0; c < HD; ++c) ov[c] *= alpha;
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                const float p = exp2_fast(s[jj] - msafe);
                l += p;
                const float* vr = Vs + (j0 + jj - c0) * HD;
#pragma unroll
                for (int c = 0; c < HD; ++c) ov[c] = fmaf(p, vr[c], ov[c]);
            }
            m = mn;
        }
    }
    if (active) {
        float* op = o + ((int64_t)n * a.Sq + i) * D + h * HD;
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < HD; ++c) op[c] = ov[c] * inv;
        lse[((int64_t)n * a.H + h) * a.Sq + i] = m + log2f(l);   // base-2 units
    }
}

// dQ pass: lane per query, K/V through LDS.  Also emits delta[i] = sum_c dO[i][c]*O[i][c].
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ o, const float* __restrict__ dO, const float* __restrict__ lse,
    AttnDims a, float* __restrict__ dq, float* __restrict__ delta) {
    constexpr int KCH = 2048 / HD;
    __shared__ __attribute__((aligned(16))) float Ks[KCH * HD];
    __shared__ __attribute__((aligned(16))) float Vs[KCH * HD];
    const int qblocks = (a.Sq + 255) >> 8;
    const int qb = blockIdx.x % qblocks, nh = blockIdx.x / qblocks;
    const int h = nh % a.H, n = nh / a.H;
    const int D = a.H * HD;
    const int i = qb * 256 + threadIdx.x;
    const bool active = i < a.Sq;
    const int wave_last = __builtin_amdgcn_readfirstlane(qb * 256 + (threadIdx.x >> 6) * 64 + 63);
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
    const int jend_blk = a.causal ? min(a.Sk, qb * 256 + 256) : a.Sk;
    for (int c0 = 0; c0 < jend_blk; c0 += KCH) {
        __syncthreads();
        stage_rows<HD>(kb, D, c0, KCH, a.Sk, Ks);
        stage_rows<HD>(vb, D, c0, KCH, a.Sk, Vs);
        __syncthreads();
        const int jw = a.causal ? min(c0 + KCH, min(a.Sk, wave_last + 1)) : min(c0 + KCH, a.Sk);
#pragma unroll 4
        for (int j = c0; j < jw; ++j) {
            const float* kr = Ks + (j - c0) * HD;
            const float* vr = Vs + (j - c0) * HD;
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
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[roff + c] = acc[c] * a.rsd;
        delta[((int64_t)n * a.H + h) * a.Sq + i] = dl;
    }
}

// dK/dV pass: lane per key; Q, dO, LSE and delta rows come through LDS.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ delta,
    AttnDims a, float* __restrict__ dk, float* __restrict__ dv) {
    constexpr int QCH = 2048 / HD;
    __shared__ __attribute__((aligned(16))) float Qs[QCH * HD];
    __shared__ __attribute__((aligned(16))) float Gs[QCH * HD];
    __shared__ float Ls[QCH];
    __shared__ float Ds[QCH];
    const int kblocks = (a.Sk + 255) >> 8;
    const int kbk = blockIdx.x % kblocks, nh = blockIdx.x / kblocks;
    const int h = nh % a.H, n = nh / a.H;
    const int D = a.H * HD;
    const int j = kbk * 256 + threadIdx.x;
    const bool active = j < a.Sk;
    const int wave_first = __builtin_amdgcn_readfirstlane(kbk * 256 + (threadIdx.x >> 6) * 64);
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
    // causal: only queries i >= first key of the block matter
    const int ibeg_blk = a.causal ? (kbk * 256) / QCH * QCH : 0;
    for (int c0 = ibeg_blk; c0 < a.Sq; c0 += QCH) {
        __syncthreads();
        stage_rows<HD>(qb, D, c0, QCH, a.Sq, Qs);
        stage_rows<HD>(dob, D, c0, QCH, a.Sq, Gs);
        for (int r = threadIdx.x; r < QCH; r += 256) {
            Ls[r] = c0 + r < a.Sq ? lb[c0 + r] : 0.0f;
            Ds[r] = c0 + r < a.Sq ? db[c0 + r] : 0.0f;
        }
        __syncthreads();
        const int i0 = a.causal ? max(c0, wave_first) : c0;
        const int i1 = min(c0 + QCH, a.Sq);
#pragma unroll 4
        for (int i = i0; i < i1; ++i) {
            const float* qr = Qs + (i - c0) * HD;
            const float* dor = Gs + (i - c0) * HD;
            float dot = 0.0f, dp = 0.0f;
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                dot = fmaf(qr[c], kv[c], dot);
                dp = fmaf(dor[c], vv[c], dp);
            }
            float p = exp2_fast(fmaf(dot, a.c2, -Ls[i - c0]));
            if (a.causal && j > i) p = 0.0f;
            const float ds = p * (dp - Ds[i - c0]);
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                dva[c] = fmaf(p, dor[c], dva[c]);
                dka[c] = fmaf(ds, qr[c], dka[c]);
            }
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


// Single-query decode step against a key/value cache (generation with a KV cache: one new
// token per sequence).  One wave per (sequence, head): lanes own keys j = lane, lane+64, ...
// with a private online softmax, merged across the wave at the end.  When k_new/v_new are
// given they are appended to the cache at row `len` and attended as the last key, so the
// caller needs no separate cache-append launch.  `len` may come from device memory
// (len_dev) so that a captured hipGraph of the whole decode step can be replayed as the
// sequence grows.
template <int HD>
__global__ __launch_bounds__(64) void attn_decode_kernel(
    const float* __restrict__ q, const float* __restrict__ k_new, const float* __restrict__ v_new,
    float* __restrict__ kc, float* __restrict__ vc, int64_t bstride, int H, int len_arg,
    const int* __restrict__ len_dev, int max_len, float c2, const float* __restrict__ o_mul,
    float* __restrict__ o) {
    const int h = blockIdx.x % H, n = blockIdx.x / H;
    const int D = H * HD;
    const int lane = threadIdx.x;
    int L = len_dev ? *len_dev : len_arg;
    const bool app = k_new != nullptr;
    L = min(max(L, 0), app ? max_len - 1 : max_len);
    const int Sk = L + (app ? 1 : 0);
    const int64_t row = (int64_t)n * D + h * HD;
    float* kb = kc + (int64_t)n * bstride + h * HD;
    float* vb = vc + (int64_t)n * bstride + h * HD;
    if (app && lane < HD) {
        kb[(int64_t)L * D + lane] = k_new[row + lane];
        vb[(int64_t)L * D + lane] = v_new[row + lane];
    }
    float qv[HD], ov[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        qv[c] = q[row + c];
        ov[c] = 0.0f;
    }
    float m = -INFINITY, l = 0.0f;
    // U rows per lane are fetched back to back (the step is latency bound), then folded
    // into the running softmax in key order
    constexpr int U = HD <= 16 ? 4 : 2;
    for (int j0 = lane; j0 < Sk; j0 += 64 * U) {
        float kk[U][HD], vv[U][HD];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + 64 * u;
            const bool in = j < Sk;
            const bool fresh = app && j == L;   // the new row comes from its source, not the cache
            const float* kr = fresh ? k_new + row : kb + (int64_t)(in ? j : 0) * D;
            const float* vr = fresh ? v_new + row : vb + (int64_t)(in ? j : 0) * D;
#pragma unroll
            for (int c = 0; c < HD; c += 4) {
                const float4 a = *reinterpret_cast<const float4*>(kr + c);
                const float4 b = *reinterpret_cast<const float4*>(vr + c);
                kk[u][c] = a.x; kk[u][c + 1] = a.y; kk[u][c + 2] = a.z; kk[u][c + 3] = a.w;
                vv[u][c] = b.x; vv[u][c + 1] = b.y; vv[u][c + 2] = b.z; vv[u][c + 3] = b.w;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j0 + 64 * u >= Sk) break;
            float dot = 0.0f;
#pragma unroll
            for (int c = 0; c < HD; ++c) dot = fmaf(qv[c], kk[u][c], dot);
            const float t = dot * c2;
            const float mn = fmaxf(m, t);
            const float alpha = exp2_fast(m - mn);
            const float p = exp2_fast(t - mn);
            l = l * alpha + p;
#pragma unroll
            for (int c = 0; c < HD; ++c) ov[c] = fmaf(p, vv[u][c], ov[c] * alpha);
            m = mn;
        }
    }
    const float M = wave_max(m);
    const float sc = m == -INFINITY ? 0.0f : exp2_fast(m - M);
    l = wave_sum(l * sc);
    const float inv = 1.0f / l;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        const float t = wave_sum(ov[c] * sc);
        if (lane == 0) o[row + c] = o_mul ? (t * inv) * o_mul[row + c] : t * inv;
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
    dim3 grid(N * H * ((Sq + 255) / 256)), block(256);
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
    const int qblocks = N * H * ((Sq + 255) / 256);
    const int kblocks = N * H * ((Sk + 255) / 256);
    dim3 block(256);
    QARIG_HD_DISPATCH(d, hipLaunchKernelGGL((attn_bwd_dq_kernel<HD>), dim3(qblocks), block,
                                            0, (hipStream_t)stream, q, k, v, o, dO, lse, a, dq,
                                            delta));
    QARIG_CHECK_LAUNCH("attention_bwd dq");
    QARIG_HD_DISPATCH(d, hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD>), dim3(kblocks), block,
                                            0, (hipStream_t)stream, q, k, v, dO, lse, delta, a, dk,
                                            dv));
    QARIG_CHECK_LAUNCH("attention_bwd dkv");
    return QARIG_OK;
}

// One decode step with a KV cache (generation; no reference counterpart - the reference
// re-runs the whole window per token, generate_images.py:283-307).  q, k_new, v_new, o:
// (B, H*d); kcache/vcache: row j of sequence n at n*batch_stride + j*H*d, at least max_len
// rows.  Attends keys 0..len-1 from the cache plus, when k_new/v_new are non-NULL, the new
// key, which is also stored at row len.  With k_new == NULL (cross-attention against
// precomputed encoder K/V) the cache is read-only.  len_dev (device int, optional)
// overrides len so the call can sit inside a captured graph.  o_mul (B, H*d), optional: the
// output is multiplied elementwise by it (the residual layer's scale(cond), layers.py:293-295).
extern "C" int qarig_attention_decode(const float* q, const float* k_new, const float* v_new,
                                      float* kcache, float* vcache, int B, int H, int d, int len,
                                      const int* len_dev, int max_len, int64_t batch_stride,
                                      float sqrt_d, const float* o_mul, float* o, void* stream) {
    QARIG_CHECK_ARG(q && kcache && vcache && o, "attention_decode: null pointer");
    QARIG_CHECK_ARG((k_new == nullptr) == (v_new == nullptr),
                    "attention_decode: k_new and v_new go together");
    QARIG_CHECK_ARG(B > 0 && H > 0 && d > 0 && max_len > 0, "attention_decode: bad extents");
    QARIG_CHECK_ARG(batch_stride >= (int64_t)max_len * H * d,
                    "attention_decode: batch_stride smaller than max_len rows");
    if (!len_dev) {
        QARIG_CHECK_ARG(len >= 0 && (k_new ? len < max_len : (len > 0 && len <= max_len)),
                        "attention_decode: len out of range for the cache");
    }
    const float c2 = 1.4426950408889634f / sqrt_d;
    QARIG_HD_DISPATCH(d, hipLaunchKernelGGL((attn_decode_kernel<HD>), dim3(B * H), dim3(64), 0,
                                            (hipStream_t)stream, q, k_new, v_new, kcache, vcache,
                                            batch_stride, H, len, len_dev, max_len, c2, o_mul, o));
    QARIG_CHECK_LAUNCH("attention_decode");
    return QARIG_OK;
}
