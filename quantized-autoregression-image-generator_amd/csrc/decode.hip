// Single-token decode step (generate_images.py:256-345 evaluates the decoder once per sampled
// token): the Linear layers of ONE new row per sequence are weight-streaming matrix-vector
// products -- 1 to 4 MB of fp32 weights against <= 16 activation rows -- chained by data
// dependencies, ~80 launches per token.  Such a launch is not bandwidth-bound (16 KB per CU) but
// LATENCY-bound: its time is the number of dependent memory round trips on its critical path.
// decode_linear_kernel therefore issues EVERY load of the launch before it waits for anything:
// the activation rows, the LayerNorm operands, the epilogue operands of the lane's final output
// and the weights (non-temporal, fully coalesced: a workgroup streams a contiguous run of weight
// rows, 4 KB per load instruction); one wait, fma chains on the vector ALU, a halving butterfly
// across the wave, one LDS hand-over between the waves that share a column, epilogue, store.
// Every summation order is fixed (no atomics): results are run-to-run bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qarig_common.h"

namespace qarig {

struct DecLin {
    const float* X; int64_t ldx, x_gs;       // activations (M, K); group stride (0: shared by the groups)
    const float* W; int64_t ldw, w_gs;       // weights (N, K) per group, reduction-contiguous
    const float* bias; int64_t bias_gs;      // (N) per group, or null
    float* C; int64_t ldc, c_gs;             // out (M, N) per group
    const float* residual; int64_t ldr;      // (M, N) added before the activation, or null
    const float* mul; int64_t ldmul;         // (M, N) elementwise factor on the output, or null; ldmul 0: one row
    const float* gamma; const float* beta;   // LN = 1: nn.LayerNorm affine form (K)
    const float* scale; const float* shift;  // LN = 2: AdaLN rows (M, K) at ldmod; ldmod 0: one row for all
    int64_t ldmod;
    float eps;
    int M, N, K, act;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4_nt(const float* p) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 r = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
    return make_float4(r.x, r.y, r.z, r.w);
}

template <int CTRL>
__device__ __forceinline__ float dpp_lane(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// Sums V per-lane values over the 64 lanes of a wave in about 3 V vector instructions instead of 6 V
// shuffles: at every step a lane hands half of its values to a partner lane and keeps the sums of the
// other half.  The four steps inside a 16-lane row are DPP operands (row_mirror, row_half_mirror, two
// quad permutations: no LDS round trip); the two steps across rows are ds_bpermute exchanges, issued
// back to back per step.  Returns the number of the value whose wave total v[0] holds on this lane;
// `owner` is true on exactly one lane per value.  Fixed order: bit-reproducible.
template <int V>
__device__ __forceinline__ int wave_sum_multi(float (&v)[V], int lane, bool& owner) {
    static_assert(V >= 1 && V <= 64 && (V & (V - 1)) == 0, "power of two, at most the wave");
    int idx = 0, dup = 0;
#define QARIG_ROW_STEP(CTRL, BIT, N0)                                       \
    if constexpr ((N0) > 1) {                                               \
        constexpr int n = (N0) / 2;                                         \
        const bool up = (lane & (BIT)) != 0;                                \
        _Pragma("unroll") for (int i = 0; i < n; ++i) {                     \
            const float send = up ? v[i] : v[i + n];                        \
            const float keep = up ? v[i + n] : v[i];                        \
            v[i] = keep + dpp_lane<CTRL>(send);                             \
        }                                                                   \
        if (up) idx += n;                                                   \
    } else {                                                                \
        v[0] += dpp_lane<CTRL>(v[0]);                                       \
        dup |= (BIT);                                                       \
    }
    constexpr int N1 = V, N2 = N1 > 1 ? N1 / 2 : 1, N3 = N2 > 1 ? N2 / 2 : 1, N4 = N3 > 1 ? N3 / 2 : 1,
                  N5 = N4 > 1 ? N4 / 2 : 1, N6 = N5 > 1 ? N5 / 2 : 1;
    QARIG_ROW_STEP(0x140, 8, N1)    // row_mirror: lane i <-> 15 - i
    QARIG_ROW_STEP(0x141, 4, N2)    // row_half_mirror: i <-> 7 - i
    QARIG_ROW_STEP(0x4E, 2, N3)     // quad_perm [2,3,0,1]
    QARIG_ROW_STEP(0xB1, 1, N4)     // quad_perm [1,0,3,2]
#undef QARIG_ROW_STEP
#define QARIG_XROW_STEP(MASK, N0)                                           \
    if constexpr ((N0) > 1) {                                               \
        constexpr int n = (N0) / 2;                                         \
        const bool up = (lane & (MASK)) != 0;                               \
        float got[n];                                                       \
        _Pragma("unroll") for (int i = 0; i < n; ++i) got[i] = __shfl_xor(up ? v[i] : v[i + n], MASK, 64); \
        _Pragma("unroll") for (int i = 0; i < n; ++i) v[i] = (up ? v[i + n] : v[i]) + got[i];              \
        if (up) idx += n;                                                   \
    } else {                                                                \
        v[0] += __shfl_xor(v[0], MASK, 64);                                 \
        dup |= (MASK);                                                      \
    }
    QARIG_XROW_STEP(16, N5)
    QARIG_XROW_STEP(32, N6)
#undef QARIG_XROW_STEP
    owner = (lane & dup) == 0;
    return idx;
}

// MR: activation rows held (M padded up); LN: 0 none, 1 gamma/beta, 2 scale/shift rows, 3 one scale/shift
// row for every activation row (all rows of a decode step sit at the same window position);
// J: 16-B weight loads per thread; KS: 1-K chunks of a weight row per thread (K = 1024 KS floats
// when KS > 1).  256 threads.  K/4 = kq float4 per row:
//   KS == 1: kq in {64, 128, 256}; the workgroup's load pass j covers 256/kq whole rows,
//            thread t sits in row group t / kq at float4 t % kq; a column is summed over kq/64 waves;
//   KS  > 1: kq = 256 KS; passes j = c KS + s cover chunk s of row c; a column is summed over all 4 waves.
template <int MR, int LN, int J, int KS>
__global__ __launch_bounds__(256) void decode_linear_kernel(DecLin p) {
    static_assert(J % KS == 0 && (LN == 0 || KS == 1) && LN >= 0 && LN <= 3, "");
    constexpr int NC = J / KS;          // distinct columns per thread
    constexpr int V = NC * MR;          // partial sums per thread
    __shared__ float red[4][V];
    __shared__ float stat[4][MR];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int g = blockIdx.y;
    // kq = K/4 is a power of two (host-checked): shifts instead of integer divisions
    const int kqs = KS == 1 ? 31 - __builtin_clz(p.K >> 2) : 8;     // log2 kq (KS > 1: of the 256-float4 chunk)
    const int cgs = KS == 1 ? 8 - kqs : 0;                          // log2 CG, CG = 256 / kq row groups per pass
    const int WS = 4 >> cgs;                                        // waves that share a column
    const int cg = KS == 1 ? t >> kqs : 0;                          // wave-uniform
    const int kc = KS == 1 ? t & ((1 << kqs) - 1) : t;
    constexpr int ncs = NC == 1 ? 0 : (NC == 2 ? 1 : 2);
    const int cws = ncs + cgs;                                      // log2 CW, CW = NC * CG columns of this workgroup
    const int n0 = blockIdx.x << cws;
    const float* X = p.X + (int64_t)g * p.x_gs;
    const float* W = p.W + (int64_t)g * p.w_gs;

    // ---- every load of the launch, oldest first in the order they are needed.  No load sits behind a
    //      branch (the compiler waits for a conditional load where its value meets the alternative):
    //      rows / columns past the end re-read the last one, absent operands read X; what they produce
    //      is never stored.
    const int Ml = p.M - 1, Nl = p.N - 1;
    float4 xv[KS][MR];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int m = 0; m < MR; ++m)
            xv[s][m] = ld4(X + (int64_t)min(m, Ml) * p.ldx + 4 * (kc + 256 * s));
    float4 lg, lb, ls[LN == 2 ? MR : 1], lh[LN == 2 ? MR : 1];
    if (LN == 1 || LN == 3) {
        lg = ld4((LN == 1 ? p.gamma : p.scale) + 4 * kc);
        lb = ld4((LN == 1 ? p.beta : p.shift) + 4 * kc);
    }
    if (LN == 2) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int64_t mo = (int64_t)min(m, Ml) * p.ldmod + 4 * kc;
            ls[m] = ld4(p.scale + mo);
            lh[m] = ld4(p.shift + mo);
        }
    }
    // the output this thread will finish: row t / CW, column t % CW
    const int om = t >> cws, ocl = t & ((1 << cws) - 1);
    const int on = n0 + ocl;
    const bool oval = om < p.M && on < p.N;
    const int omc = min(om, Ml), onc = min(on, Nl);
    const float eb = *(p.bias ? p.bias + (int64_t)g * p.bias_gs + onc : X);
    const float er = *(p.residual ? p.residual + (int64_t)omc * p.ldr + onc : X);
    const float em = *(p.mul ? p.mul + (int64_t)omc * p.ldmul + onc : X);
    float4 wv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = j / KS, s = j % KS;
        wv[j] = ld4_nt(W + (int64_t)min(n0 + (c << cgs) + cg, Nl) * p.ldw + 4 * (kc + 256 * s));
    }
    __builtin_amdgcn_sched_barrier(0);      // nothing that waits for a load moves in front of the last issue

    // ---- LayerNorm of the rows on the way in (two passes: mean, then centred squares --
    //      layernorm_fwd_kernel's form); the weights are still in flight
    if (LN) {
        const float invK = 1.0f / (float)p.K;       // K is a power of two here: exact
        float s1[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) s1[m] = (xv[0][m].x + xv[0][m].y) + (xv[0][m].z + xv[0][m].w);
        bool own;
        int idx = wave_sum_multi<MR>(s1, lane, own);
        if (own) stat[w][idx] = s1[0];
        __syncthreads();
        float mean[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float a = stat[cg * WS][m];
#pragma nounroll
            for (int i = 1; i < WS; ++i) a += stat[cg * WS + i][m];
            mean[m] = a * invK;
        }
        __syncthreads();
        float s2[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float4& x = xv[0][m];
            x.x -= mean[m]; x.y -= mean[m]; x.z -= mean[m]; x.w -= mean[m];
            float q = x.x * x.x;
            q = fmaf(x.y, x.y, q); q = fmaf(x.z, x.z, q); q = fmaf(x.w, x.w, q);
            s2[m] = q;
        }
        idx = wave_sum_multi<MR>(s2, lane, own);
        if (own) stat[w][idx] = s2[0];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float a = stat[cg * WS][m];
#pragma nounroll
            for (int i = 1; i < WS; ++i) a += stat[cg * WS + i][m];
            // v_rsq_f32 (1 ulp) + one Newton step: within an ulp of 1 / sqrtf()
            const float var = a * invK + p.eps;
            float rstd = __builtin_amdgcn_rsqf(var);
            rstd = rstd * (1.5f - 0.5f * var * rstd * rstd);
            float4& x = xv[0][m];
            const float4 gg = LN == 2 ? ls[LN == 2 ? m : 0] : lg;
            const float4 hh = LN == 2 ? lh[LN == 2 ? m : 0] : lb;
            x.x = (x.x * rstd) * gg.x + hh.x; x.y = (x.y * rstd) * gg.y + hh.y;
            x.z = (x.z * rstd) * gg.z + hh.z; x.w = (x.w * rstd) * gg.w + hh.w;
        }
    }

    // ---- partial dot products: k ascending inside the thread's chunk(s)
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = j / KS, s = j % KS;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float a = acc[c * MR + m];
            a = fmaf(wv[j].x, xv[s][m].x, a); a = fmaf(wv[j].y, xv[s][m].y, a);
            a = fmaf(wv[j].z, xv[s][m].z, a); a = fmaf(wv[j].w, xv[s][m].w, a);
            acc[c * MR + m] = a;
        }
    }
    bool own;
    const int idx = wave_sum_multi<V>(acc, lane, own);
    if (own) red[w][idx] = acc[0];
    __syncthreads();
    if (oval) {
        const int c = ocl >> cgs, ocg = ocl & ((1 << cgs) - 1);
        float v = red[ocg * WS][c * MR + om];
#pragma nounroll
        for (int i = 1; i < WS; ++i) v += red[ocg * WS + i][c * MR + om];
        if (p.bias) v += eb;
        if (p.residual) v += er;
        v = p.act == ACT_SILU ? v * sigmoid_f(v) : act_fwd(v, p.act);
        if (p.mul) v *= em;
        p.C[(int64_t)g * p.c_gs + (int64_t)om * p.ldc + on] = v;
    }
}

}  // namespace qarig

using namespace qarig;

// Shapes the streaming kernel takes: M <= 16 rows, K/4 in {64, 128, 256} or K in {2048, 4096};
// a LayerNorm prologue only with K <= 1024 (a workgroup's threads cover whole rows).
extern "C" int qarig_decode_linear_supported(int M, int N, int K, int ln) {
    if (M < 1 || M > 16 || N < 1 || K < 256) return 0;
    if (K == 256 || K == 512 || K == 1024) return 1;
    return (K == 2048 || K == 4096) && !ln;
}

template <int MR, int LN>
static void launch_decode_linear(const DecLin& p, int groups, hipStream_t st) {
    const int kq = p.K / 4;
    auto grid = [&](int cw) { return dim3((p.N + cw - 1) / cw, groups); };
    auto wgs = [&](int cw) { return (int64_t)((p.N + cw - 1) / cw) * groups; };
#define QARIG_DL(J, KS, CW) hipLaunchKernelGGL((decode_linear_kernel<MR, LN, J, KS>), grid(CW), dim3(256), 0, st, p)
    if (kq <= 256) {
        const int CG = 256 / kq;
        // 16 KB of weights per workgroup while that leaves >= 256 workgroups, else fewer bytes each
        // (16 rows of per-row AdaLN operands leave no registers for four loads)
        if constexpr (!(MR == 16 && LN == 2)) {
            if (wgs(4 * CG) >= 256) { QARIG_DL(4, 1, 4 * CG); return; }
        }
        if (wgs(2 * CG) >= 256) QARIG_DL(2, 1, 2 * CG);
        else QARIG_DL(1, 1, CG);
    } else if constexpr (LN == 0) {
        if (kq == 512) {
            if (wgs(2) >= 256) QARIG_DL(4, 2, 2);
            else QARIG_DL(2, 2, 1);
        } else {
            QARIG_DL(4, 4, 1);
        }
    }
#undef QARIG_DL
}

/* C-ABI: see include/qarig.h */
extern "C" int qarig_decode_linear_f32(const float* X, int64_t ldx, int64_t x_gs, float eps,
                                       const float* gamma, const float* beta, const float* scale,
                                       const float* shift, int64_t ldmod, const float* W, int64_t ldw,
                                       int64_t w_gs, const float* bias, int64_t bias_gs,
                                       const float* residual, int64_t ldr, const float* mul,
                                       int64_t ldmul, float* C, int64_t ldc, int64_t c_gs, int groups,
                                       int M, int N, int K, int act, void* stream) {
    QARIG_CHECK_ARG(X && W && C, "decode_linear: null operand");
    QARIG_CHECK_ARG(groups >= 1 && groups <= 65535, "decode_linear: bad group count %d", groups);
    QARIG_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "decode_linear: gamma/beta pair");
    QARIG_CHECK_ARG((scale == nullptr) == (shift == nullptr), "decode_linear: scale/shift pair");
    QARIG_CHECK_ARG(!(gamma && scale), "decode_linear: affine and AdaLN forms are exclusive");
    const int ln = gamma ? 1 : (scale ? (ldmod == 0 ? 3 : 2) : 0);
    QARIG_CHECK_ARG(qarig_decode_linear_supported(M, N, K, ln),
                    "decode_linear: needs M <= 16 and K in {256, 512, 1024} (2048, 4096 without a "
                    "LayerNorm prologue) (M=%d N=%d K=%d)", M, N, K);
    QARIG_CHECK_DIMS("decode_linear", groups, N, K);
    QARIG_CHECK_ARG(act >= 0 && act <= 3, "decode_linear: bad activation id");
    QARIG_CHECK_ARG(!ln || eps > 0.0f, "decode_linear: eps must be positive");
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    QARIG_CHECK_ARG(al16(X) && al16(W) && ldx % 4 == 0 && ldw % 4 == 0 && x_gs % 4 == 0 && w_gs % 4 == 0 &&
                        al16(gamma) && al16(beta) && al16(scale) && al16(shift) && ldmod % 4 == 0,
                    "decode_linear: operands must be 16-B aligned");
    QARIG_CHECK_ARG(ldx >= K && ldw >= K && ldc >= N && (!residual || ldr >= N) && (!mul || ldmul == 0 || ldmul >= N) &&
                        (!scale || ldmod == 0 || ldmod >= K),
                    "decode_linear: a row stride is shorter than its row");
    const DecLin p{X, ldx, x_gs, W, ldw, w_gs, bias, bias_gs, C, ldc, c_gs, residual, ldr, mul, ldmul,
                   gamma, beta, scale, shift, ldmod, eps, M, N, K, act};
    hipStream_t st = (hipStream_t)stream;
#define QARIG_DL_LN(MR)                                                     \
    switch (ln) {                                                           \
        case 0: launch_decode_linear<MR, 0>(p, groups, st); break;          \
        case 1: launch_decode_linear<MR, 1>(p, groups, st); break;          \
        case 2: launch_decode_linear<MR, 2>(p, groups, st); break;          \
        default: launch_decode_linear<MR, 3>(p, groups, st); break;         \
    }
    if (M <= 4) { QARIG_DL_LN(4) } else { QARIG_DL_LN(16) }
#undef QARIG_DL_LN
    QARIG_CHECK_LAUNCH("decode_linear");
    return QARIG_OK;
}
