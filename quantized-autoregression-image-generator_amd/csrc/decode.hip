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

// The operands a launch needs to issue its loads come first, as individual kernel arguments: the first 16
// dwords of the kernel-argument segment are preloaded into SGPRs by the command processor
// (-mllvm -amdgpu-kernarg-preload-count=16 in build.py for this file), so the first loads do not wait for a
// scalar-cache miss on the argument segment the host has just written.  The rest travels as a struct.
struct DecLin {
    const float* bias; int64_t bias_gs;      // (N) per group, or null
    float* C; int64_t ldc, c_gs;             // out (M, N) per group
    const float* residual; int64_t ldr;      // (M, N) added before the activation, or null
    const float* mul; int64_t ldmul;         // (M, N) elementwise factor on the output, or null; ldmul 0: one row
    const float* gamma; const float* beta;   // LN = 1: nn.LayerNorm affine form (K)
    const float* scale; const float* shift;  // LN = 2: AdaLN rows (M, K) at ldmod; LN = 3: one row for all
    int64_t ldmod;
    float eps;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4_nt(const float* p) {
#ifdef QARIG_DECODE_NO_NT       // ablation build (tools/): weights through the default cache policy
    return ld4(p);
#else
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 r = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
    return make_float4(r.x, r.y, r.z, r.w);
#endif
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
__global__ __launch_bounds__(256) void decode_linear_kernel(const float* __restrict__ Xb, const float* __restrict__ Wb,
                                                            int64_t ldx, int64_t ldw, int M, int N, int K, int act,
                                                            int64_t x_gs, int64_t w_gs, DecLin p) {
    static_assert(J % KS == 0 && (LN == 0 || KS == 1) && LN >= 0 && LN <= 3, "");
    constexpr int NC = J / KS;          // distinct columns per thread
    constexpr int V = NC * MR;          // partial sums per thread
    __shared__ float red[4][V];
    __shared__ float stat[4][MR];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int g = blockIdx.y;
    // kq = K/4 is a power of two (host-checked): shifts instead of integer divisions
    const int kqs = KS == 1 ? 31 - __builtin_clz(K >> 2) : 8;     // log2 kq (KS > 1: of the 256-float4 chunk)
    const int cgs = KS == 1 ? 8 - kqs : 0;                          // log2 CG, CG = 256 / kq row groups per pass
    const int WS = 4 >> cgs;                                        // waves that share a column
    const int cg = KS == 1 ? t >> kqs : 0;                          // wave-uniform
    const int kc = KS == 1 ? t & ((1 << kqs) - 1) : t;
    constexpr int ncs = NC == 1 ? 0 : (NC == 2 ? 1 : 2);
    const int cws = ncs + cgs;                                      // log2 CW, CW = NC * CG columns of this workgroup
    const int n0 = blockIdx.x << cws;
    const float* X = Xb + (int64_t)g * x_gs;
    const float* W = Wb + (int64_t)g * w_gs;

    // ---- every load of the launch, oldest first in the order they are needed.  No load sits behind a
    //      branch (the compiler waits for a conditional load where its value meets the alternative):
    //      rows / columns past the end re-read the last one, absent operands read X; what they produce
    //      is never stored.
    const int Ml = M - 1, Nl = N - 1;
    float4 xv[KS][MR];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int m = 0; m < MR; ++m)
            xv[s][m] = ld4(X + (int64_t)min(m, Ml) * ldx + 4 * (kc + 256 * s));
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
    const bool oval = om < M && on < N;
    const int omc = min(om, Ml), onc = min(on, Nl);
    const float eb = *(p.bias ? p.bias + (int64_t)g * p.bias_gs + onc : X);
    const float er = *(p.residual ? p.residual + (int64_t)omc * p.ldr + onc : X);
    const float em = *(p.mul ? p.mul + (int64_t)omc * p.ldmul + onc : X);
    float4 wv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = j / KS, s = j % KS;
        wv[j] = ld4_nt(W + (int64_t)min(n0 + (c << cgs) + cg, Nl) * ldw + 4 * (kc + 256 * s));
    }
    __builtin_amdgcn_sched_barrier(0);      // nothing that waits for a load moves in front of the last issue

    // ---- LayerNorm of the rows on the way in (two passes: mean, then centred squares --
    //      layernorm_fwd_kernel's form); the weights are still in flight
    if (LN) {
        const float invK = 1.0f / (float)K;       // K is a power of two here: exact
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
            // v_rsq_f32: 1 ulp of 1 / sqrt() -- 6e-8 relative on the row, far inside the 2e-6 of the parity tests
            // (every vector instruction of this single-wave-per-SIMD kernel is on the launch's critical path)
            const float rstd = __builtin_amdgcn_rsqf(a * invK + p.eps);
            float4& x = xv[0][m];
            const float4 gg = LN == 2 ? ls[LN == 2 ? m : 0] : lg;
            const float4 hh = LN == 2 ? lh[LN == 2 ? m : 0] : lb;
            x.x = fmaf(x.x * rstd, gg.x, hh.x); x.y = fmaf(x.y * rstd, gg.y, hh.y);
            x.z = fmaf(x.z * rstd, gg.z, hh.z); x.w = fmaf(x.w * rstd, gg.w, hh.w);
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
        v = act == ACT_SILU ? v * sigmoid_f(v) : act_fwd(v, act);
        if (p.mul) v *= em;
        p.C[(int64_t)g * p.c_gs + (int64_t)om * p.ldc + on] = v;
    }
}


// Which value's total lane `lane` holds after wave_sum_multi<V>, and whether it is that value's owner -- known
// from the lane number alone, so a lane can request the epilogue operands of its output before anything else.
template <int V>
__device__ __forceinline__ int wave_sum_multi_index(int lane, bool& owner) {
    int idx = 0, dup = 0, n = V;
#pragma unroll
    for (int step = 0; step < 6; ++step) {
        const int bit = step == 0 ? 8 : (step == 1 ? 4 : (step == 2 ? 2 : (step == 3 ? 1 : (step == 4 ? 16 : 32))));
        if (n > 1) {
            n >>= 1;
            if (lane & bit) idx += n;
        } else {
            dup |= bit;
        }
    }
    owner = (lane & dup) == 0;
    return idx;
}

// 5 ... 16 activation rows (the candidate chunks of a few images as one batch): the ROWS are split over the four
// waves of the workgroup -- wave w owns rows 4w .. 4w+3 -- and every wave streams the workgroup's whole run of
// weight rows (CW columns; the three later waves hit the lines the first one brought in).  Against holding all 16
// rows in every lane (decode_linear_kernel<16, ...>: 2,057 instructions in its LayerNorm form, every one of them on
// the critical path of a single-wave-per-SIMD launch) a lane now carries 4 rows: a quarter of the LayerNorm and
// statistics arithmetic, sums that stay inside the wave (no LDS hand-over, no workgroup barrier at all), the same
// fma count.  KW: 256-float chunks of a weight row per lane (K = 256 KW); a lane's loads j = c KW + s cover chunk
// s of column c.
template <int LN, int CW, int KW>
__global__ __launch_bounds__(256) void decode_linear_rows_kernel(const float* __restrict__ Xb, const float* __restrict__ Wb,
                                                                 int64_t ldx, int64_t ldw, int M, int N, int K, int act,
                                                                 int64_t x_gs, int64_t w_gs, DecLin p) {
    static_assert(LN >= 0 && LN <= 3 && CW * 4 <= 64, "");
    constexpr int V = CW * 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (4 * w >= M) return;                 // a whole wave without rows (no barrier below)
    const int g = blockIdx.y;
    const int n0 = blockIdx.x * CW;
    const float* X = Xb + (int64_t)g * x_gs;
    const float* W = Wb + (int64_t)g * w_gs;
    const int Ml = M - 1, Nl = N - 1;
    float4 xv[KW][4];
#pragma unroll
    for (int s = 0; s < KW; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m)
            xv[s][m] = ld4(X + (int64_t)min(4 * w + m, Ml) * ldx + 4 * (lane + 64 * s));
    float4 lg[(LN == 1 || LN == 3) ? KW : 1], lb[(LN == 1 || LN == 3) ? KW : 1];
    float4 ls[LN == 2 ? KW : 1][4], lh[LN == 2 ? KW : 1][4];
    if (LN == 1 || LN == 3) {
#pragma unroll
        for (int s = 0; s < KW; ++s) {
            lg[s] = ld4((LN == 1 ? p.gamma : p.scale) + 4 * (lane + 64 * s));
            lb[s] = ld4((LN == 1 ? p.beta : p.shift) + 4 * (lane + 64 * s));
        }
    }
    if (LN == 2) {
#pragma unroll
        for (int s = 0; s < KW; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int64_t mo = (int64_t)min(4 * w + m, Ml) * p.ldmod + 4 * (lane + 64 * s);
                ls[s][m] = ld4(p.scale + mo);
                lh[s][m] = ld4(p.shift + mo);
            }
    }
    bool own;
    const int oidx = wave_sum_multi_index<V>(lane, own);
    const int om = 4 * w + (oidx & 3), on = n0 + (oidx >> 2);
    const bool oval = own && om < M && on < N;
    const int omc = min(om, Ml), onc = min(on, Nl);
    const float eb = *(p.bias ? p.bias + (int64_t)g * p.bias_gs + onc : X);
    const float er = *(p.residual ? p.residual + (int64_t)omc * p.ldr + onc : X);
    const float em = *(p.mul ? p.mul + (int64_t)omc * p.ldmul + onc : X);
    float4 wv[CW * KW];
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int s = 0; s < KW; ++s)
            wv[c * KW + s] = ld4_nt(W + (int64_t)min(n0 + c, Nl) * ldw + 4 * (lane + 64 * s));
    __builtin_amdgcn_sched_barrier(0);

    if (LN) {
        const float invK = 1.0f / (float)K;
        float s1[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float a = 0.0f;
#pragma unroll
            for (int s = 0; s < KW; ++s) a += (xv[s][m].x + xv[s][m].y) + (xv[s][m].z + xv[s][m].w);
            s1[m] = a;
        }
        bool o4;
        wave_sum_multi<4>(s1, lane, o4);          // value m's total on the lanes with (lane >> 2) & 3 == m
        float mean[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) mean[m] = __shfl(s1[0], 4 * m, 64) * invK;
        float s2[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float q = 0.0f;
#pragma unroll
            for (int s = 0; s < KW; ++s) {
                float4& x = xv[s][m];
                x.x -= mean[m]; x.y -= mean[m]; x.z -= mean[m]; x.w -= mean[m];
                q = fmaf(x.x, x.x, q); q = fmaf(x.y, x.y, q); q = fmaf(x.z, x.z, q); q = fmaf(x.w, x.w, q);
            }
            s2[m] = q;
        }
        wave_sum_multi<4>(s2, lane, o4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float rstd = __builtin_amdgcn_rsqf(__shfl(s2[0], 4 * m, 64) * invK + p.eps);
#pragma unroll
            for (int s = 0; s < KW; ++s) {
                float4& x = xv[s][m];
                const float4 gg = LN == 2 ? ls[LN == 2 ? s : 0][m] : lg[LN == 2 ? 0 : s];
                const float4 hh = LN == 2 ? lh[LN == 2 ? s : 0][m] : lb[LN == 2 ? 0 : s];
                x.x = fmaf(x.x * rstd, gg.x, hh.x); x.y = fmaf(x.y * rstd, gg.y, hh.y);
                x.z = fmaf(x.z * rstd, gg.z, hh.z); x.w = fmaf(x.w * rstd, gg.w, hh.w);
            }
        }
    }

    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
        for (int s = 0; s < KW; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 wq = wv[c * KW + s];
                float a = acc[c * 4 + m];
                a = fmaf(wq.x, xv[s][m].x, a); a = fmaf(wq.y, xv[s][m].y, a);
                a = fmaf(wq.z, xv[s][m].z, a); a = fmaf(wq.w, xv[s][m].w, a);
                acc[c * 4 + m] = a;
            }
    bool own2;
    wave_sum_multi<V>(acc, lane, own2);        // lands where wave_sum_multi_index said
    if (oval) {
        float v = acc[0];
        if (p.bias) v += eb;
        if (p.residual) v += er;
        v = act == ACT_SILU ? v * sigmoid_f(v) : act_fwd(v, act);
        if (p.mul) v *= em;
        p.C[(int64_t)g * p.c_gs + (int64_t)om * p.ldc + on] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Device-resident state of the sampling loop (generate_images.py:256-345): the host replays captured
// graphs and never reads a token back between them.  int32 control words:
enum DecCtl : int {
    CTL_LEN = 0,     // window index of the token the next decoder step evaluates
    CTL_CUR = 1,     // window index at which the current chunk of beam_width tokens starts
    CTL_DRAW = 2,    // sampling draws made so far (row of the uniform / forced-token / probability-log buffers)
    CTL_CAND = 3,    // candidate chunks of the current chunk position evaluated so far
    CTL_TOK = 4,     // decoder steps since the candidate began = the chunk slot the next draw fills
    CTL_WORDS = 8
};

// First launch of a step: x[b] = table[ids[b]] + pe[len] (Transformer.py:154-167: embedding + the
// sinusoid of the window index) and, when the stage keeps a per-position table of every projection of
// `cond` (ScaleLayer / ShiftLayer of all layers, models/layers.py:100-153, 258-304: cond depends on
// the token's position alone and the positions of a stage are known before its loop), the copy of
// this position's row of it into the buffer the step's launches read.
__global__ __launch_bounds__(256) void decode_embed_kernel(const int64_t* __restrict__ ids, int B, int D, int V,
                                                           const float* __restrict__ table,
                                                           const float* __restrict__ pe, int* __restrict__ ctl,
                                                           int len_arg, int max_len,
                                                           const float* __restrict__ proj_table, int64_t PD,
                                                           float* __restrict__ x, float* __restrict__ proj_row,
                                                           int* __restrict__ bad) {
    int L = ctl ? ctl[CTL_LEN] : len_arg;
    L = min(max(L, 0), max_len - 1);
    const int64_t nx = (int64_t)B * D / 4, total = nx + PD / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        if (i < nx) {
            const int b = (int)(i / (D / 4)), c = (int)(i - (int64_t)b * (D / 4)) * 4;
            const int64_t id = ids[b];
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (id < 0 || id >= V) {
                if (c == 0) atomicExch(bad, 1);
            } else {
                t = ld4(table + id * D + c);
                if (pe) {
                    const float4 q = ld4(pe + (int64_t)L * D + c);
                    t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                }
            }
            *reinterpret_cast<float4*>(x + (int64_t)b * D + c) = t;
        } else {
            const int64_t j = (i - nx) * 4;
            *reinterpret_cast<float4*>(proj_row + j) = ld4(proj_table + (int64_t)L * PD + j);
        }
    }
    // a step has begun: the draw behind it fills the next slot of the chunk (no workgroup of this launch reads it)
    if (ctl && blockIdx.x == 0 && threadIdx.x == 0) ctl[CTL_TOK] += 1;
}

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
template <int CTRL> __device__ __forceinline__ float dpp_max_step(float x) { return fmaxf(x, dpp_lane<CTRL>(x)); }
__device__ __forceinline__ float wave_max_dpp(float x) {
    x = dpp_max_step<0x140>(x); x = dpp_max_step<0x141>(x); x = dpp_max_step<0x4E>(x); x = dpp_max_step<0xB1>(x);
    x = fmaxf(x, __shfl_xor(x, 16, 64));
    return fmaxf(x, __shfl_xor(x, 32, 64));
}

// One query row per sequence against its cached keys / values: a wave per (sequence, head), the four
// waves of a workgroup on four adjacent heads (they read the same 128-B lines at head dim 8), a lane
// per key.  Rows are loaded without waiting for the length word (rows past it hold finite data -- the
// cache is zero-initialised -- and are masked afterwards); the running softmax of a lane's keys is
// combined over the wave with one DPP maximum and one multi-value sum.
template <int HD>
__global__ __launch_bounds__(256) void decode_attention_kernel(
    const float* __restrict__ q, const float* __restrict__ k_new, const float* __restrict__ v_new,
    float* __restrict__ kc, float* __restrict__ vc, int64_t bstride, int64_t hstride, int64_t rstride, int H,
    int len_arg, const int* __restrict__ ctl, int max_len, float c2, const float* __restrict__ o_mul,
    int64_t ldmul, float* __restrict__ o) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int hq = (H + 3) >> 2;
    const int n = blockIdx.x / hq, h = (blockIdx.x - n * hq) * 4 + w;
    if (h >= H) return;                      // whole wave; no workgroup barrier below
    const int D = H * HD;
    const bool app = k_new != nullptr;
    const int64_t row = (int64_t)n * D + h * HD;
    // cache row j of head h: n * bstride + h * hstride + j * rstride (row-major rows of H * d floats, or
    // head-major -- DecodeCache's layout: a head's keys contiguous, a wave-level load is one 2-KB run at d = 8)
    float* kb = kc + (int64_t)n * bstride + (int64_t)h * hstride;
    float* vb = vc + (int64_t)n * bstride + (int64_t)h * hstride;
    constexpr int U = HD <= 16 ? 4 : (HD <= 32 ? 2 : 1);     // keys per lane and pass (128 registers of rows)
    float qv[HD], kn[HD], vn[HD];
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
        const float4 a = ld4(q + row + c);
        qv[c] = a.x; qv[c + 1] = a.y; qv[c + 2] = a.z; qv[c + 3] = a.w;
        const float4 b = ld4((app ? k_new : q) + row + c), d = ld4((app ? v_new : q) + row + c);
        kn[c] = b.x; kn[c + 1] = b.y; kn[c + 2] = b.z; kn[c + 3] = b.w;
        vn[c] = d.x; vn[c + 1] = d.y; vn[c + 2] = d.z; vn[c + 3] = d.w;
    }
    float4 kk[U][HD / 4], vv[U][HD / 4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = min(lane + 64 * u, max_len - 1);
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
            kk[u][c] = ld4(kb + (int64_t)j * rstride + 4 * c);
            vv[u][c] = ld4(vb + (int64_t)j * rstride + 4 * c);
        }
    }
    const float mq = *(o_mul ? o_mul + (int64_t)n * ldmul + h * HD + min(lane, HD - 1) : q + row);
    int L = ctl ? ctl[CTL_LEN] : len_arg;
    L = min(max(L, 0), app ? max_len - 1 : max_len);
    const int Sk = L + (app ? 1 : 0);
    if (app && lane < HD) {                  // the new row joins the cache (off the critical path)
        kb[(int64_t)L * rstride + lane] = k_new[row + lane];
        vb[(int64_t)L * rstride + lane] = v_new[row + lane];
    }
    float m = -INFINITY, l = 0.0f, ov[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) ov[c] = 0.0f;
    for (int j0 = 0; j0 < Sk; j0 += 64 * U) {
        if (j0 > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(j0 + lane + 64 * u, max_len - 1);
#pragma unroll
                for (int c = 0; c < HD / 4; ++c) {
                    kk[u][c] = ld4(kb + (int64_t)j * rstride + 4 * c);
                    vv[u][c] = ld4(vb + (int64_t)j * rstride + 4 * c);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + lane + 64 * u;
            const bool fresh = app && j == L;      // the new row comes from its source, not from the cache
            float dot = 0.0f;
#pragma unroll
            for (int c = 0; c < HD / 4; ++c) {
                dot = fmaf(qv[4 * c], fresh ? kn[4 * c] : kk[u][c].x, dot);
                dot = fmaf(qv[4 * c + 1], fresh ? kn[4 * c + 1] : kk[u][c].y, dot);
                dot = fmaf(qv[4 * c + 2], fresh ? kn[4 * c + 2] : kk[u][c].z, dot);
                dot = fmaf(qv[4 * c + 3], fresh ? kn[4 * c + 3] : kk[u][c].w, dot);
            }
            if (j < Sk) {
                const float t = dot * c2;
                const float mn = fmaxf(m, t);
                const float alpha = exp2_fast(m - mn);
                const float pr = exp2_fast(t - mn);
                l = l * alpha + pr;
#pragma unroll
                for (int c = 0; c < HD / 4; ++c) {
                    ov[4 * c] = fmaf(pr, fresh ? vn[4 * c] : vv[u][c].x, ov[4 * c] * alpha);
                    ov[4 * c + 1] = fmaf(pr, fresh ? vn[4 * c + 1] : vv[u][c].y, ov[4 * c + 1] * alpha);
                    ov[4 * c + 2] = fmaf(pr, fresh ? vn[4 * c + 2] : vv[u][c].z, ov[4 * c + 2] * alpha);
                    ov[4 * c + 3] = fmaf(pr, fresh ? vn[4 * c + 3] : vv[u][c].w, ov[4 * c + 3] * alpha);
                }
                m = mn;
            }
        }
    }
    const float Mx = wave_max_dpp(m);
    const float sc = m == -INFINITY ? 0.0f : exp2_fast(m - Mx);
    // {l, o[0..HD)} summed over the wave together: value number 0 is l, 1 + c is o[c]
    constexpr int NV = HD < 8 ? 8 : (HD < 16 ? 16 : (HD < 32 ? 32 : 64));
    static_assert(HD + 1 <= NV || HD == 64, "");
    if constexpr (HD < 64) {
        float r[NV];
        r[0] = l * sc;
#pragma unroll
        for (int c = 0; c < HD; ++c) r[1 + c] = ov[c] * sc;
#pragma unroll
        for (int c = HD + 1; c < NV; ++c) r[c] = 0.0f;
        bool own;
        const int idx = wave_sum_multi<NV>(r, lane, own);
        // value 0 (l) ends on lane 0 (no bit of its number set); lane c < HD holds o_mul[c]
        const float lt = __shfl(r[0], 0, 64);
        const float mv = __shfl(mq, min(max(idx - 1, 0), HD - 1), 64);
        if (own && idx >= 1 && idx <= HD) {
            const float t = r[0] / lt;
            o[row + idx - 1] = o_mul ? t * mv : t;
        }
    } else {
        float lsum[1] = {l * sc};
        bool own;
        wave_sum_multi<1>(lsum, lane, own);
        float r[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) r[c] = ov[c] * sc;
        const int idx = wave_sum_multi<64>(r, lane, own);
        const float mv = __shfl(mq, idx, 64);
        const float t = r[0] / lsum[0];
        o[row + idx] = o_mul ? t * mv : t;
    }
}

__device__ __forceinline__ int wave_min_int(int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = min(x, __shfl_xor(x, o, 64));
    return x;
}
__device__ __forceinline__ int wave_max_int(int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = max(x, __shfl_xor(x, o, 64));
    return x;
}

// One sampling draw per row, as generate_images.py:289-304 makes it (train_quantized_transformer.py:
// 626-636 in "train" mode): probs = softmax(logits / T); generate mode zeroes the <end> probability;
// a token is drawn in proportion to probs -- by inverse CDF from ONE uniform of the caller's generator
// (torch.multinomial consumes its generator differently: the same distribution, not the same stream) --
// the running product of the chosen probabilities is updated, train mode maps a drawn <end> to 0, and the
// token (+ shift: the base stage's vocabulary offset) is appended: to ids (the next step's input) and to
// slot `slot` of the row's chunk.  forced (optional, >= 0 entries): tokens to take instead of drawing
// (tests replay the reference's recorded draws); probs_log (optional): receives the row it sampled from.
// A wave per row; every sum runs in a fixed order.
__global__ __launch_bounds__(64) void decode_sample_kernel(
    const float* __restrict__ logits, int64_t ldl, int B, int V, float temperature, int end_token,
    int generate_mode, int64_t shift, const float* __restrict__ uniforms, const int64_t* __restrict__ forced,
    int* __restrict__ ctl, int slot, int bw, int max_draws, int inc_len, int beams, int64_t* __restrict__ ids,
    int64_t* __restrict__ chunk, float* __restrict__ comb, float* __restrict__ probs_log) {
    __shared__ float part[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (slot < 0) slot = min(max(ctl[CTL_TOK], 0), bw - 1);     // the slot the device counts
    // beams == 0: row b draws from column b of draw row ctl[2] + slot (B columns).  beams > 0: the rows are
    // `beams` candidate chunks per image evaluated as one batch, and every draw keeps the number the reference's
    // candidate-after-candidate loop gives it (generate_images.py:262-304): row b = image * beams + candidate
    // reads column `image` of draw row ctl[2] + candidate * bw + slot (B / beams columns)
    const int dcol = beams > 0 ? b / beams : b, dcols = beams > 0 ? B / beams : B;
    const int d = min(max(ctl[CTL_DRAW] + (beams > 0 ? (b - dcol * beams) * bw : 0) + slot, 0), max_draws - 1);
    const float* z = logits + (int64_t)b * ldl;
    const int per = (V + 63) / 64, i0 = min(V, lane * per), i1 = min(V, i0 + per);
    float mx = -INFINITY;
    for (int i = i0; i < i1; ++i) mx = fmaxf(mx, z[i] / temperature);
    mx = wave_max_dpp(mx);
    float s[1] = {0.0f};
    for (int i = i0; i < i1; ++i) s[0] += expf(z[i] / temperature - mx);
    bool own;
    wave_sum_multi<1>(s, lane, own);
    const float S = s[0];
    auto prob = [&](int i) {
        const float pr = expf(z[i] / temperature - mx) / S;
        return (generate_mode && i == end_token) ? 0.0f : pr;
    };
    float ps = 0.0f;
    int last_nz = -1;
    float* lg = probs_log ? probs_log + ((int64_t)d * dcols + dcol) * V : nullptr;
    for (int i = i0; i < i1; ++i) {
        const float pr = prob(i);
        ps += pr;
        if (pr > 0.0f) last_nz = i;
        if (lg) lg[i] = pr;
    }
    part[lane] = ps;
    __syncthreads();
    float base = 0.0f, total = 0.0f;
    for (int k = 0; k < 64; ++k) {
        if (k == lane) base = total;
        total += part[k];
    }
    const float target = uniforms[(int64_t)d * dcols + dcol] * total;
    int cand = 0x7fffffff;
    float run = base;
    for (int i = i0; i < i1; ++i) {
        const float pr = prob(i);
        run += pr;
        if (pr > 0.0f && run > target && cand == 0x7fffffff) cand = i;
    }
    cand = wave_min_int(cand);
    const int fallback = wave_max_int(last_nz);       // rounding left the target at or past the total
    int64_t nxt = cand == 0x7fffffff ? max(fallback, 0) : cand;
    if (forced) {
        const int64_t f = forced[(int64_t)d * dcols + dcol];
        if (f >= 0 && f < V) nxt = f;
    }
    if (lane == 0) {
        comb[b] *= prob((int)nxt);
        if (!generate_mode && nxt == end_token) nxt = 0;      // reference HACK: <end> -> index 0
        ids[b] = nxt + shift;
        chunk[(int64_t)b * bw + slot] = nxt + shift;
        if (b == 0 && inc_len) ctl[CTL_LEN] += 1;             // no other workgroup of this launch reads it
    }
}

// After a candidate chunk (all rows have drawn beam_width tokens): per image, the beam with the largest
// probability product (first one on ties: torch.argmax) competes with the best chunk kept so far --
// generate_images.py:325-337 keeps the earlier candidate unless the new product is larger -- and the
// counters move on: next candidate, the draws it consumed, the cache position back at the chunk start.
// take[n] = 1 + winning beam when the new chunk replaces the kept one, else 0.
__global__ __launch_bounds__(64) void decode_decide_kernel(int* __restrict__ ctl, int N, int NB, int bw, int draws,
                                                           float* __restrict__ comb,
                                                           const int64_t* __restrict__ chunk,
                                                           float* __restrict__ best_p,
                                                           int64_t* __restrict__ best_chunk,
                                                           int* __restrict__ take) {
    const int cand = ctl[CTL_CAND];
    for (int n = threadIdx.x; n < N; n += 64) {
        int pb = 0;
        float c = comb[(int64_t)n * NB];
        for (int k = 1; k < NB; ++k)
            if (comb[(int64_t)n * NB + k] > c) { c = comb[(int64_t)n * NB + k]; pb = k; }
        const bool tk = cand == 0 || !(best_p[n] >= c);
        take[n] = tk ? pb + 1 : 0;
        if (tk) {
            best_p[n] = c;
            for (int j = 0; j < bw; ++j) best_chunk[(int64_t)n * bw + j] = chunk[((int64_t)n * NB + pb) * bw + j];
        }
        for (int k = 0; k < NB; ++k) comb[(int64_t)n * NB + k] = 1.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl[CTL_CAND] = cand + 1;
        ctl[CTL_DRAW] += draws;
        ctl[CTL_LEN] = ctl[CTL_CUR];
        ctl[CTL_TOK] = 0;
    }
}

// Cache rows [cur, cur + R) of every layer, keys and values: restore == 0 saves the winning beam's rows
// of the images whose chunk was just kept (take[n] > 0) into the staging copy; restore == 1 writes the
// staged rows into EVERY beam of every image (the kept chunk becomes the common prefix).
// kv (layers * 2, N * NB, H, max_len, d) head-major; staged (layers * 2, N, H, R, d).
__global__ __launch_bounds__(256) void decode_rows_kernel(const int* __restrict__ ctl, float* __restrict__ kv,
                                                          float* __restrict__ staged,
                                                          const int* __restrict__ take, int layers2, int N,
                                                          int NB, int H, int R, int d, int max_len, int restore) {
    const int cur = ctl[CTL_CUR];
    const int64_t dq = d / 4;
    const int64_t per = (restore ? (int64_t)NB : 1) * H * R * dq;
    const int64_t total = (int64_t)layers2 * N * per;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r_ = i;
        const int c = (int)(r_ % dq); r_ /= dq;
        const int r = (int)(r_ % R); r_ /= R;
        const int h = (int)(r_ % H); r_ /= H;
        int bm = 0;
        if (restore) { bm = (int)(r_ % NB); r_ /= NB; }
        const int n = (int)(r_ % N);
        const int l2 = (int)(r_ / N);
        const int row = cur + r;
        if (row < 0 || row >= max_len) continue;
        float* st = staged + ((((int64_t)l2 * N + n) * H + h) * R + r) * d + 4 * c;
        if (restore) {
            float* dst = kv + ((((int64_t)l2 * N * NB + (int64_t)n * NB + bm) * H + h) * max_len + row) * d + 4 * c;
            *reinterpret_cast<float4*>(dst) = ld4(st);
        } else {
            const int tk = take[n];
            if (tk <= 0) continue;
            const float* src = kv + ((((int64_t)l2 * N * NB + (int64_t)n * NB + (tk - 1)) * H + h) * max_len + row) * d + 4 * c;
            *reinterpret_cast<float4*>(st) = ld4(src);
        }
    }
}

// The kept chunk becomes part of the sequence: tokens[n][cur + j] = best_chunk[n][j]; its last token is
// the input of the step that produces the next chunk's first logits, at window index cur + bw - 1.
__global__ __launch_bounds__(64) void decode_commit_kernel(int* __restrict__ ctl, int N, int NB, int bw,
                                                           const int64_t* __restrict__ best_chunk,
                                                           int64_t* __restrict__ tokens, int64_t ldt,
                                                           int64_t* __restrict__ ids) {
    const int cur = ctl[CTL_CUR];
    for (int n = threadIdx.x; n < N; n += 64) {
        for (int j = 0; j < bw; ++j)
            if (cur + j >= 0 && cur + j < ldt) tokens[(int64_t)n * ldt + cur + j] = best_chunk[(int64_t)n * bw + j];
        for (int k = 0; k < NB; ++k) ids[(int64_t)n * NB + k] = best_chunk[(int64_t)n * bw + bw - 1];
    }
    __syncthreads();
    if (threadIdx.x == 0) ctl[CTL_LEN] = cur + bw - 1;
}

// Behind that step: the next chunk starts.
__global__ void decode_advance_kernel(int* __restrict__ ctl, int bw) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctl[CTL_CUR] += bw;
        ctl[CTL_LEN] = ctl[CTL_CUR];
        ctl[CTL_CAND] = 0;
        ctl[CTL_TOK] = 0;
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

struct DecLinHead { const float* X; const float* W; int64_t ldx, ldw; int M, N, K, act; int64_t x_gs, w_gs; };

template <int MR, int LN>
static void launch_decode_linear(const DecLinHead& h, const DecLin& p, int groups, hipStream_t st) {
    const int kq = h.K / 4;
    auto grid = [&](int cw) { return dim3((h.N + cw - 1) / cw, groups); };
    auto wgs = [&](int cw) { return (int64_t)((h.N + cw - 1) / cw) * groups; };
#define QARIG_DL(J, KS, CW)                                                                                     \
    hipLaunchKernelGGL((decode_linear_kernel<MR, LN, J, KS>), grid(CW), dim3(256), 0, st, h.X, h.W, h.ldx, h.ldw, \
                       h.M, h.N, h.K, h.act, h.x_gs, h.w_gs, p)
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
// 5 ... 16 rows, K <= 2048: the row-split kernel.  CW columns per workgroup: 16 KB of weights while that leaves
// >= 256 workgroups, else a quarter of it.
template <int LN>
static bool launch_decode_linear_rows(const DecLinHead& h, const DecLin& p, int groups, hipStream_t st) {
    const int kw = h.K / 256;
    // one wave per 4 rows; <= 4 rows: single-wave workgroups, so four times the workgroups for the bytes per CU
    const int threads = 64 * ((h.M + 3) / 4);
    const int64_t want = h.M <= 4 ? 1024 : 256;
    auto wgs = [&](int cw) { return (int64_t)((h.N + cw - 1) / cw) * groups; };
#define QARIG_DR(CW, KW)                                                                                                  \
    hipLaunchKernelGGL((decode_linear_rows_kernel<LN, CW, KW>), dim3((h.N + CW - 1) / CW, groups), dim3(threads), 0, st, \
                       h.X, h.W, h.ldx, h.ldw, h.M, h.N, h.K, h.act, h.x_gs, h.w_gs, p)
    switch (kw) {
        case 1: if (wgs(16) >= want) QARIG_DR(16, 1); else QARIG_DR(4, 1); return true;
        case 2: if (wgs(8) >= want) QARIG_DR(8, 2); else QARIG_DR(2, 2); return true;
        case 4: if (wgs(4) >= want) QARIG_DR(4, 4); else QARIG_DR(1, 4); return true;
        case 8:
            if constexpr (LN == 0) {
                if (wgs(2) >= want) QARIG_DR(2, 8); else QARIG_DR(1, 8);
                return true;
            }
            return false;
        default: return false;
    }
#undef QARIG_DR
}

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
    const DecLinHead h{X, W, ldx, ldw, M, N, K, act, x_gs, w_gs};
    const DecLin p{bias, bias_gs, C, ldc, c_gs, residual, ldr, mul, ldmul, gamma, beta, scale, shift, ldmod, eps};
    hipStream_t st = (hipStream_t)stream;
#define QARIG_DL_LN(MR)                                                     \
    switch (ln) {                                                           \
        case 0: launch_decode_linear<MR, 0>(h, p, groups, st); break;          \
        case 1: launch_decode_linear<MR, 1>(h, p, groups, st); break;          \
        case 2: launch_decode_linear<MR, 2>(h, p, groups, st); break;          \
        default: launch_decode_linear<MR, 3>(h, p, groups, st); break;         \
    }
    bool done = false;
    if ((M > 4 && g_qarig_opt.decode_rows != 0) || g_qarig_opt.decode_rows == 2) {
        switch (ln) {
            case 0: done = launch_decode_linear_rows<0>(h, p, groups, st); break;
            case 1: done = launch_decode_linear_rows<1>(h, p, groups, st); break;
            case 2: done = launch_decode_linear_rows<2>(h, p, groups, st); break;
            default: done = launch_decode_linear_rows<3>(h, p, groups, st); break;
        }
    }
    if (done) {
    } else if (M <= 4) { QARIG_DL_LN(4) } else { QARIG_DL_LN(16) }
#undef QARIG_DL_LN
    QARIG_CHECK_LAUNCH("decode_linear");
    return QARIG_OK;
}


extern "C" int qarig_decode_embed(const int64_t* ids, int B, int D, int V, const float* table,
                                  const float* pe, int* ctl, int len, int max_len,
                                  const float* proj_table, int64_t proj_row_floats, float* x,
                                  float* proj_row, int* bad_flag, void* stream) {
    QARIG_CHECK_ARG(ids && table && x && bad_flag, "decode_embed: null pointer");
    QARIG_CHECK_ARG(B > 0 && D > 0 && V > 0 && max_len > 0 && D % 4 == 0, "decode_embed: bad extents (D %% 4 == 0)");
    QARIG_CHECK_DIMS("decode_embed", B, D);
    QARIG_CHECK_DIMS("decode_embed", V, D);
    QARIG_CHECK_DIMS("decode_embed", max_len, D);
    QARIG_CHECK_ARG(ctl || (len >= 0 && len < max_len), "decode_embed: len out of range");
    QARIG_CHECK_ARG(proj_row_floats >= 0 && proj_row_floats % 4 == 0 && proj_row_floats <= (1LL << 30) &&
                        (proj_row_floats == 0 || (proj_table && proj_row)),
                    "decode_embed: projection row must be a multiple of 4 floats with both pointers given");
    QARIG_CHECK_ARG(proj_row_floats == 0 || qarig_dims_ok({max_len, proj_row_floats}), "decode_embed: table too large");
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    QARIG_CHECK_ARG(al16(table) && al16(pe) && al16(x) && al16(proj_table) && al16(proj_row),
                    "decode_embed: operands must be 16-B aligned");
    const int64_t items = (int64_t)B * D / 4 + proj_row_floats / 4;
    const int blocks = (int)((items + 255) / 256 < 1024 ? (items + 255) / 256 : 1024);
    hipLaunchKernelGGL(decode_embed_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, B, D, V, table, pe,
                       ctl, len, max_len, proj_table, proj_row_floats, x, proj_row, bad_flag);
    QARIG_CHECK_LAUNCH("decode_embed");
    return QARIG_OK;
}

extern "C" int qarig_decode_attention(const float* q, const float* k_new, const float* v_new, float* kcache,
                                      float* vcache, int B, int H, int d, int len, const int* len_dev,
                                      int max_len, int64_t batch_stride, int64_t head_stride, int64_t row_stride,
                                      float sqrt_d, const float* o_mul, int64_t ldmul, float* o, void* stream) {
    QARIG_CHECK_ARG(q && kcache && vcache && o, "decode_attention: null pointer");
    QARIG_CHECK_ARG((k_new == nullptr) == (v_new == nullptr), "decode_attention: k_new and v_new go together");
    QARIG_CHECK_ARG(B > 0 && H > 0 && d > 0 && max_len > 0 && sqrt_d > 0.0f, "decode_attention: bad extents");
    QARIG_CHECK_DIMS("decode_attention", B, H, max_len);
    QARIG_CHECK_ARG(d == 4 || d == 8 || d == 16 || d == 32 || d == 64,
                    "decode_attention: head dim %d unsupported (4,8,16,32,64)", d);
    QARIG_CHECK_ARG((long long)B * H < (1LL << 31), "decode_attention: bad extents");
    const bool row_major = head_stride == d && row_stride == (int64_t)H * d && batch_stride >= (int64_t)max_len * H * d;
    const bool head_major = row_stride == d && head_stride >= (int64_t)max_len * d && head_stride <= (1LL << 40) &&
                            batch_stride >= (int64_t)H * head_stride;
    QARIG_CHECK_ARG((row_major || head_major) && batch_stride % 4 == 0 && head_stride % 4 == 0,
                    "decode_attention: cache strides are neither row-major (rows of H * d) nor head-major "
                    "(a head's max_len rows of d), or shorter than max_len rows");
    QARIG_CHECK_ARG(!o_mul || ldmul == 0 || ldmul >= (int64_t)H * d, "decode_attention: ldmul shorter than a row");
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    QARIG_CHECK_ARG(al16(q) && al16(k_new) && al16(v_new) && al16(kcache) && al16(vcache),
                    "decode_attention: operands must be 16-B aligned");
    if (!len_dev) {
        QARIG_CHECK_ARG(len >= 0 && (k_new ? len < max_len : (len > 0 && len <= max_len)),
                        "decode_attention: len out of range for the cache");
    }
    const float c2 = 1.4426950408889634f / sqrt_d;
    const dim3 grid(B * ((H + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define QARIG_DA(HD)                                                                                          \
    hipLaunchKernelGGL((decode_attention_kernel<HD>), grid, block, 0, st, q, k_new, v_new, kcache, vcache,    \
                       batch_stride, head_stride, row_stride, H, len, len_dev, max_len, c2, o_mul, ldmul, o)
    switch (d) {
        case 4: QARIG_DA(4); break;
        case 8: QARIG_DA(8); break;
        case 16: QARIG_DA(16); break;
        case 32: QARIG_DA(32); break;
        default: QARIG_DA(64); break;
    }
#undef QARIG_DA
    QARIG_CHECK_LAUNCH("decode_attention");
    return QARIG_OK;
}

extern "C" int qarig_decode_sample(const float* logits, int64_t ldl, int B, int V, float temperature,
                                   int end_token, int generate_mode, int64_t shift, const float* uniforms,
                                   const int64_t* forced, int* ctl, int slot, int beam_width, int max_draws,
                                   int inc_len, int beams, int64_t* ids, int64_t* chunk, float* comb,
                                   float* probs_log, void* stream) {
    QARIG_CHECK_ARG(logits && uniforms && ctl && ids && chunk && comb, "decode_sample: null pointer");
    QARIG_CHECK_ARG(B > 0 && V > 0 && beam_width > 0 && max_draws > 0 && slot >= -1 && slot < beam_width,
                    "decode_sample: bad extents");
    QARIG_CHECK_DIMS("decode_sample", B, V);
    QARIG_CHECK_DIMS("decode_sample", max_draws, B);
    QARIG_CHECK_DIMS("decode_sample", B, beam_width);
    QARIG_CHECK_ARG(ldl >= V && temperature > 0.0f, "decode_sample: ldl < V or temperature <= 0");
    QARIG_CHECK_ARG(beams >= 0 && (beams == 0 || B % beams == 0), "decode_sample: beams must divide the rows");
    QARIG_CHECK_ARG(!probs_log || qarig_dims_ok({max_draws, B, V}), "decode_sample: probability log too large");
    hipLaunchKernelGGL(decode_sample_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, ldl, B, V,
                       temperature, end_token, generate_mode, shift, uniforms, forced, ctl, slot, beam_width,
                       max_draws, inc_len, beams, ids, chunk, comb, probs_log);
    QARIG_CHECK_LAUNCH("decode_sample");
    return QARIG_OK;
}

extern "C" int qarig_decode_decide(int* ctl, int N, int NB, int beam_width, int draws, float* comb,
                                   const int64_t* chunk, float* best_p, int64_t* best_chunk, int* take, void* stream) {
    QARIG_CHECK_ARG(ctl && comb && chunk && best_p && best_chunk && take, "decode_decide: null pointer");
    QARIG_CHECK_ARG(N > 0 && NB > 0 && beam_width > 0 && draws >= 0 && draws <= (1 << 24), "decode_decide: bad extents");
    QARIG_CHECK_DIMS("decode_decide", N, NB, beam_width);
    hipLaunchKernelGGL(decode_decide_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, N, NB, beam_width, draws,
                       comb, chunk, best_p, best_chunk, take);
    QARIG_CHECK_LAUNCH("decode_decide");
    return QARIG_OK;
}

extern "C" int qarig_decode_rows(const int* ctl, float* kv, float* staged, const int* take, int layers2, int N,
                                 int NB, int H, int R, int d, int max_len, int restore, void* stream) {
    QARIG_CHECK_ARG(ctl && kv && staged && (restore || take), "decode_rows: null pointer");
    QARIG_CHECK_ARG(layers2 > 0 && N > 0 && NB > 0 && H > 0 && R > 0 && d > 0 && d % 4 == 0 && max_len > 0,
                    "decode_rows: bad extents (d %% 4 == 0)");
    QARIG_CHECK_DIMS("decode_rows", layers2, N, NB, max_len);
    QARIG_CHECK_DIMS("decode_rows", layers2, N, NB, R);
    QARIG_CHECK_DIMS("decode_rows", max_len, H, d);
    QARIG_CHECK_DIMS("decode_rows", R, H, d);
    QARIG_CHECK_ARG(qarig_dims_ok({layers2, (long long)N * NB, (long long)max_len * H * d}, 1LL << 40, 1LL << 44),
                    "decode_rows: cache too large");
    QARIG_CHECK_ARG((((uintptr_t)kv | (uintptr_t)staged) & 15) == 0, "decode_rows: operands must be 16-B aligned");
    const int64_t total = (int64_t)layers2 * N * (restore ? NB : 1) * H * R * (d / 4);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(decode_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ctl, kv, staged, take,
                       layers2, N, NB, H, R, d, max_len, restore);
    QARIG_CHECK_LAUNCH("decode_rows");
    return QARIG_OK;
}

extern "C" int qarig_decode_commit(int* ctl, int N, int NB, int beam_width, const int64_t* best_chunk,
                                   int64_t* tokens, int64_t ldt, int64_t* ids, void* stream) {
    QARIG_CHECK_ARG(ctl && best_chunk && tokens && ids, "decode_commit: null pointer");
    QARIG_CHECK_ARG(N > 0 && NB > 0 && beam_width > 0 && ldt > 0, "decode_commit: bad extents");
    QARIG_CHECK_DIMS("decode_commit", N, NB, beam_width);
    QARIG_CHECK_ARG(qarig_dims_ok({N, ldt}, 1LL << 40), "decode_commit: token buffer too large");
    hipLaunchKernelGGL(decode_commit_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, N, NB, beam_width,
                       best_chunk, tokens, ldt, ids);
    QARIG_CHECK_LAUNCH("decode_commit");
    return QARIG_OK;
}

extern "C" int qarig_decode_advance(int* ctl, int beam_width, void* stream) {
    QARIG_CHECK_ARG(ctl && beam_width > 0, "decode_advance: bad arguments");
    hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, beam_width);
    QARIG_CHECK_LAUNCH("decode_advance");
    return QARIG_OK;
}
