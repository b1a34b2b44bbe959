// The GEMM epilogue shared by the fp32 (gemm.hip) and reduced-precision (gemm_lp.hip) kernels:
// bias, residual, saved pre-activation, activation, `* act'(z)` backward fusion, split-K slabs,
// through a 16-B-wide LDS-staged store path.  Accumulators are the 2x2 32x32 MFMA tiles of a
// wave's 64x64 sub-tile (the C/D register map is the same for the f32 and bf16 MFMAs).
#pragma once
#include <type_traits>
#include "qarig_common.h"

namespace qarig {

struct GemmEpilogue {
    float* C; int64_t ldc;
    const float* bias;                    // [N], per column, or null
    const float* residual; int64_t ldr;   // [M][N] added before the activation, or null
    float* preact; int64_t ldp;           // receives acc+bias+residual, or null
    int act;                              // activation applied to what goes to C
    const float* gradz; int64_t ldz;      // C *= act'(gradz[m][n]) (backward fusion), or null
    int gact;
    float* rowsum;                        // [splitk][M]: sum_k A(m,k) per K split, or null
    unsigned short* Cb; int64_t ldcb;     // optional bf16 copy of what goes to C (consumer GEMMs), or null
    unsigned short* Pb; int64_t ldpb;     // optional bf16 copy of the saved pre-activation, or null
    const unsigned short* gradzb; int64_t ldzb;   // gradz given in bf16 (reduced-precision mode), or null
    // fp8 operands (gemm_f8_kernel): the accumulator is multiplied by *alpha_a * *alpha_b (the two
    // per-tensor dequantisation factors, device scalars written by the cast kernels) first; or null
    const float* alpha_a; const float* alpha_b;
};

// operand / output pointers of a grouped launch (gemm.hip gemm_dma_pf_grouped_kernel, gemm_x3.hip)
constexpr int GEMM_MAX_GROUPS = 16;
struct GemmGroupPtrs {
    const float* A[GEMM_MAX_GROUPS];
    const float* B[GEMM_MAX_GROUPS];
    float* C[GEMM_MAX_GROUPS];
    const float* bias[GEMM_MAX_GROUPS];
    const float* residual[GEMM_MAX_GROUPS];
    float* preact[GEMM_MAX_GROUPS];
    const float* gradz[GEMM_MAX_GROUPS];
    float* rowsum[GEMM_MAX_GROUPS];
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned int hi16) { return __uint_as_float(hi16 << 16); }

// two fp32 -> one dword of two bf16 (round to nearest even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t f = {a, b};
    const bf16x2_t r = __builtin_convertvector(f, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}

// Wide epilogue (interior tiles): the wave's 64x64 tile goes through its private 8 KB of
// LDS in two 32-row halves, so that every global access of the epilogue (C, preact,
// residual, gradz, slabs) is a 16-B-per-lane, 256-B-per-row dwordx4 instead of 4x as many
// 4-B accesses (the epilogue is store-issue bound otherwise).  `lds`: >= 32 KB, free.
// General form: the wave's tile starts at (mb, nb); `stage` = this wave's 32 x 64 floats of LDS.
// Accumulators of a wave's 64x64 sub-tile as 4x4 tiles of 16x16 (v_mfma_f32_16x16x32_bf16: lane l
// holds rows 4 (l >> 4) + r, r = 0..3, of column l & 15); the epilogue only differs in how a
// 32-row half is written to the staging buffer.
struct Acc16 {
    f32x4 t[4][4];   // [16-row tile][16-column tile]
};
template <int NJ>
__device__ __forceinline__ void acc_stage_half(const Acc& acc, int i, float* stage, int lane) {
    constexpr int EL = 32 * NJ;
    const int cl = lane & 31;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            stage[acc_row(r, lane) * EL + j * 32 + cl] = acc.t[i][j][r];
}
template <int NJ>
__device__ __forceinline__ void acc_stage_half(const Acc16& acc, int i, float* stage, int lane) {
    static_assert(NJ == 2, "16x16 accumulators cover the full 64 columns");
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                stage[(h * 16 + 4 * (lane >> 4) + r) * 64 + tj * 16 + (lane & 15)] = acc.t[2 * i + h][tj][r];
}

// PRE: what the epilogue requests from memory BEFORE it stages a half's accumulators (see
// gemm_epilogue_wave_t): 0 nothing, 1 a bf16 saved pre-activation (16 registers: the 16-wave kernels with
// 128 registers per lane), 2 also fp32 pre-activations and the residual (64 more: 4-wave kernels).
template <int NJ = 2, class AccT = Acc, int PRE = 0>
__device__ __forceinline__ void gemm_epilogue_wave(const AccT& acc, const GemmEpilogue& ep, float* stage,
                                                   int mb, int nb, int M, int N, int splitk,
                                                   float* slabs, int i_begin = 0, int i_end = 2);

// Activation ids as template arguments: ACT_RUNTIME keeps the per-element switch on the runtime id.
// The epilogue is specialised for the combinations the Transformer uses (plain, SiLU forward, SiLU
// backward fusion): with the switch inside the element loop every output value costs two scalar
// branch trees, and a K = 512 tile spends ~15 % of its time there (fp32 MFMA and the rest of the
// instruction stream do not overlap on this chip).
constexpr int ACT_RUNTIME = -1;
template <int A>
__device__ __forceinline__ float act_fwd_t(float x, int act) {
    if constexpr (A == ACT_RUNTIME) return act_fwd(x, act);
    else if constexpr (A == ACT_SILU) return x * sigmoid_f(x);
    else return x;
}
template <int A>
__device__ __forceinline__ float act_grad_t(float x, int act) {
    if constexpr (A == ACT_RUNTIME) return act_grad(x, act);
    else if constexpr (A == ACT_SILU) {
        const float sg = sigmoid_f(x);
        return sg * (1.0f + x * (1.0f - sg));
    } else return 1.0f;
}
template <int NJ, int A, int GA, class AccT, int PRE>
__device__ __forceinline__ void gemm_epilogue_wave_t(const AccT& acc, const GemmEpilogue& ep, float* stage,
                                                     int mb, int nb, int M, int N, int splitk,
                                                     float* slabs, int i_begin, int i_end);

template <int PRE = 0>
__device__ __forceinline__ void gemm_epilogue_wide(const Acc& acc, const GemmEpilogue& ep, float* lds,
                                                   int m0, int n0, int M, int N, int splitk,
                                                   float* slabs) {
    const int wave = threadIdx.x >> 6;
    // 4 waves x 32 x 64 floats = 32 KB
    gemm_epilogue_wave<2, Acc, PRE>(acc, ep, lds + wave * (32 * 64), m0 + (wave >> 1) * 64, n0 + (wave & 1) * 64,
                                    M, N, splitk, slabs);
}

// [i_begin, i_end): which 32-row halves of the wave's tile this call stores (wave-uniform; the
// paired kernel gives each of its two waves per tile one half).
template <int NJ, class AccT, int PRE>
__device__ __forceinline__ void gemm_epilogue_wave(const AccT& acc, const GemmEpilogue& ep, float* stage,
                                                   int mb, int nb, int M, int N, int splitk,
                                                   float* slabs, int i_begin, int i_end) {
    const bool grad = ep.gradz != nullptr || ep.gradzb != nullptr;
    if (ep.act == ACT_NONE && !grad)
        gemm_epilogue_wave_t<NJ, ACT_NONE, ACT_NONE, AccT, PRE>(acc, ep, stage, mb, nb, M, N, splitk, slabs, i_begin, i_end);
    else if (ep.act == ACT_SILU && !grad)
        gemm_epilogue_wave_t<NJ, ACT_SILU, ACT_NONE, AccT, PRE>(acc, ep, stage, mb, nb, M, N, splitk, slabs, i_begin, i_end);
    else if (ep.act == ACT_NONE && ep.gact == ACT_SILU)
        gemm_epilogue_wave_t<NJ, ACT_NONE, ACT_SILU, AccT, PRE>(acc, ep, stage, mb, nb, M, N, splitk, slabs, i_begin, i_end);
    else
        gemm_epilogue_wave_t<NJ, ACT_RUNTIME, ACT_RUNTIME, AccT, PRE>(acc, ep, stage, mb, nb, M, N, splitk, slabs, i_begin, i_end);
}

template <int NJ, int A, int GA, class AccT, int PRE>
__device__ __forceinline__ void gemm_epilogue_wave_t(const AccT& acc, const GemmEpilogue& ep, float* stage,
                                                     int mb, int nb, int M, int N, int splitk,
                                                     float* slabs, int i_begin, int i_end) {
    const int lane = threadIdx.x & 63;
    // unpadded rows are conflict-free for both the b32 writes (half-waves hit different rows)
    // and the b128 reads.  The wave's tile is 64 rows x 32*NJ columns (NJ = 1: accumulators t[i][0])
    constexpr int EL = 32 * NJ;
    constexpr int LPR = 8 * NJ;            // lanes per staged row (4 columns each)
    const int er = lane / LPR, ec = (lane % LPR) * 4;
    const int gc = nb + ec;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ep.bias && splitk == 1) bv = *reinterpret_cast<const float4*>(ep.bias + gc);
    const float alpha = ep.alpha_a ? *ep.alpha_a * *ep.alpha_b : 1.0f;
    // Operands the epilogue READS (the saved pre-activation of the backward fusion, the residual) are
    // requested before a half's accumulators are staged, all its rows at once: issued inside the row loop
    // each load waits out its own HBM latency, and a one-workgroup-per-CU kernel has nothing else to run
    // meanwhile (dH = dY W2 * act'(t1) at 32768 x 2048 x 512 in bf16: 164 us against 109 us for the
    // forward product of the same shape, which reads nothing here).
    constexpr int NIT = 4 * NJ;
    constexpr bool WIDE_PRE = PRE >= 2;
    const bool ingest = splitk == 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i < i_begin || i >= i_end) continue;
        uint2 zb[PRE >= 1 ? NIT : 1];
        float4 zf[WIDE_PRE ? NIT : 1], rf[WIDE_PRE ? NIT : 1];
        if constexpr (PRE >= 1 && GA != ACT_NONE) {
            if (ingest && ep.gradzb && !ep.gradz) {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    zb[it] = *reinterpret_cast<const uint2*>(
                        ep.gradzb + (int64_t)(mb + i * 32 + it * (64 / LPR) + er) * ep.ldzb + gc);
            }
        }
        if constexpr (WIDE_PRE) {
            if (ingest) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int64_t row = mb + i * 32 + it * (64 / LPR) + er;
                    if (ep.residual) rf[it] = *reinterpret_cast<const float4*>(ep.residual + row * ep.ldr + gc);
                    if constexpr (GA != ACT_NONE)
                        if (ep.gradz) zf[it] = *reinterpret_cast<const float4*>(ep.gradz + row * ep.ldz + gc);
                }
            }
        }
        acc_stage_half<NJ>(acc, i, stage, lane);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int lr = it * (64 / LPR) + er;
            const int64_t row = mb + i * 32 + lr;
            float4 t = *reinterpret_cast<const float4*>(stage + lr * EL + ec);
            if (splitk > 1) {
                *reinterpret_cast<float4*>(slabs + ((int64_t)blockIdx.z * M + row) * N + gc) = t;
                continue;
            }
            if (ep.alpha_a) { t.x *= alpha; t.y *= alpha; t.z *= alpha; t.w *= alpha; }
            t.x += bv.x; t.y += bv.y; t.z += bv.z; t.w += bv.w;
            if (ep.residual) {
                float4 rv;
                if constexpr (WIDE_PRE) rv = rf[it];
                else rv = *reinterpret_cast<const float4*>(ep.residual + row * ep.ldr + gc);
                t.x += rv.x; t.y += rv.y; t.z += rv.z; t.w += rv.w;
            }
            if (ep.preact) *reinterpret_cast<float4*>(ep.preact + row * ep.ldp + gc) = t;
            if (ep.Pb)
                *reinterpret_cast<uint2*>(ep.Pb + row * ep.ldpb + gc) =
                    make_uint2(pack_bf16x2(t.x, t.y), pack_bf16x2(t.z, t.w));
            float4 y = make_float4(act_fwd_t<A>(t.x, ep.act), act_fwd_t<A>(t.y, ep.act),
                                   act_fwd_t<A>(t.z, ep.act), act_fwd_t<A>(t.w, ep.act));
            if constexpr (GA != ACT_NONE) {   // (a gradz with gact = none multiplies by 1)
                if (ep.gradz) {
                    float4 z;
                    if constexpr (WIDE_PRE) z = zf[it];
                    else z = *reinterpret_cast<const float4*>(ep.gradz + row * ep.ldz + gc);
                    y.x *= act_grad_t<GA>(z.x, ep.gact); y.y *= act_grad_t<GA>(z.y, ep.gact);
                    y.z *= act_grad_t<GA>(z.z, ep.gact); y.w *= act_grad_t<GA>(z.w, ep.gact);
                } else if (ep.gradzb) {
                    uint2 z2;
                    if constexpr (PRE >= 1) z2 = zb[it];
                    else z2 = *reinterpret_cast<const uint2*>(ep.gradzb + row * ep.ldzb + gc);
                    y.x *= act_grad_t<GA>(bf16_bits_to_f32(z2.x & 0xffffu), ep.gact);
                    y.y *= act_grad_t<GA>(bf16_bits_to_f32(z2.x >> 16), ep.gact);
                    y.z *= act_grad_t<GA>(bf16_bits_to_f32(z2.y & 0xffffu), ep.gact);
                    y.w *= act_grad_t<GA>(bf16_bits_to_f32(z2.y >> 16), ep.gact);
                }
            }
            if (ep.C) *reinterpret_cast<float4*>(ep.C + row * ep.ldc + gc) = y;
            if (ep.Cb)
                *reinterpret_cast<uint2*>(ep.Cb + row * ep.ldcb + gc) =
                    make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
        }
        __syncthreads();
    }
}

}  // namespace qarig
