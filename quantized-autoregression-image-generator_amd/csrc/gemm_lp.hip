// Reduced-precision GEMM for BASELINE config 5 (opt-in; never the fp32 parity path):
// bf16 operands IN HBM, products on v_mfma_f32_32x32x16_bf16, fp32 accumulation, the same
// fused fp32 epilogue as the fp32 kernels (gemm_epilogue.h) plus optional bf16 copies of the
// outputs for the next GEMM.  Replaces the Linear contractions of models/layers.py:234-254,
// 330-340, 389-418 when qarig.ops precision is "bf16".
//
//   NT:  C[M,N] = sum_k A[m][k] * B[n][k]      A (M,K), B (N,K) bf16, reduction-contiguous
//        (forward  x W^T;  input gradient  dT W  with the W^T shadow as B)
//   TN:  C[M,N] = sum_k A[k][m] * B[k][n]      A (K,M), B (K,N) bf16, reduction-major
//        (weight gradient  dT^T x: both operands are the row-major activations as they lie)
//   NN:  C[M,N] = sum_k A[m][k] * B[k][n]      A (M,K) reduction-contiguous, B (K,N) reduction-major
//        (input gradient  dT W  on the weight shadow AS STORED (N_out,K_in): no W^T copy)
//
// Tile 128 x 128 x 64, 4 waves (2x2), each wave 2x2 accumulators of 32x32; the operand tiles
// go global -> LDS by global_load_lds_dwordx4 (no VGPR staging) into a 2-stage ring, one
// barrier per k-tile (the structure of gemm_dma_kernel).  The LDS image is lane-linear, so
// the bank swizzles sit on the SOURCE address and on the read (guide rule 21):
//   NT tile [128 rows][64 k]   (128-B rows):  16-B chunk c of row r at chunk c ^ ((r >> 1) & 7): two
//        rows share one 256-B bank line, and a ds_read_b128 lane group (16 lanes, NOT contiguous:
//        rows {0-3,12-15,20-27} or {4-11,16-19,28-31} of the fragment) must land on 16 distinct
//        16-B slots of it -- (r & 7) leaves every read 2-way conflicted (SQ_LDS_BANK_CONFLICT was
//        44 % of the LDS cycles);
//        fragments = one ds_read_b128 per (32-row tile, 16-deep k-step);
//   TN tile [64 k][128 x]      (256-B rows):  chunk c of k-row r at c ^ (((r&3)<<2) | ((r>>2)&3));
//        fragments = two ds_read_b64_tr_b16 (the hardware transpose read: 4 k-rows x 16 columns
//        per 16-lane group, delivered column-major), so the reduction-major activations feed
//        the MFMA without a transposed copy in HBM.
#include <stdlib.h>

#include "gemm_epilogue.h"

namespace qarig {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_lp;
typedef __attribute__((address_space(1))) const void* glb_ptr_lp;
typedef unsigned short bf16_t;   // storage type of a bf16 element

constexpr int LBK = 64;                         // reduction depth per staged tile
constexpr int LP_OP = 128 * LBK;                // bf16 elements per operand tile (16 KB)
constexpr int LP_STAGE = 2 * LP_OP;             // A then B

__device__ __forceinline__ unsigned lds_addr_lp(const bf16_t* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const bf16_t*)p;
}
__device__ __forceinline__ int tn_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// This wave's share (4 x 1 KiB) of one operand tile.
template <bool TN>
__device__ __forceinline__ void lp_stage(const bf16_t* __restrict__ P, int64_t ld, int x0, int k0,
                                         bf16_t* tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wave * 4 + i;
        const bf16_t* src;
        if (!TN) {   // [x][k]: 8 rows x 128 B per instruction
            const int r = q * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            src = P + (int64_t)(x0 + r) * ld + k0 + c * 8;
        } else {     // [k][x]: 4 k-rows x 256 B per instruction
            const int r = q * 4 + (lane >> 4);
            const int c = (lane & 15) ^ tn_swz(r);
            src = P + (int64_t)(k0 + r) * ld + x0 + c * 8;
        }
        __builtin_amdgcn_global_load_lds((glb_ptr_lp)src, (lds_ptr_lp)(tile + q * 512), 16, 0, 0);
    }
}

// Fragment of the 32-row (column) tile starting at x0 for k-step ks (16 deep): lane l holds
// element j = operand(x0 + (l & 31), k = 16 ks + 8 (l >> 5) + j).  The LDS reads are inline asm
// (hipcc would otherwise drain the DMA queue in front of every LDS read); their destination
// registers are touched again only in value(), which callers invoke behind the lgkmcnt wait.
typedef short s16x4_lp __attribute__((ext_vector_type(4)));
typedef short s16x8_lp __attribute__((ext_vector_type(8)));
template <bool TN> struct LpFrag {
    bf16x8 v;
    __device__ __forceinline__ bf16x8 value() const { return v; }
};
template <> struct LpFrag<true> {
    s16x4_lp lo, hi;
    __device__ __forceinline__ bf16x8 value() const {
        const s16x8_lp both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, both);
    }
};
template <bool TN>
__device__ __forceinline__ void lp_frag(const bf16_t* tile, int x0, int ks, int lane, LpFrag<TN>& f) {
    if constexpr (!TN) {
        const int r = x0 + (lane & 31);
        const int c = (ks * 2 + (lane >> 5)) ^ ((r >> 1) & 7);
        const unsigned a = lds_addr_lp(tile) + r * 128 + (c << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.v) : "v"(a));
    } else {
        const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
        const int ch = ((x0 + 16 * (g & 1)) >> 3) + (p >> 1);
        const int r0 = ks * 16 + 8 * (g >> 1) + q;
        const int r1 = r0 + 4;
        const unsigned base = lds_addr_lp(tile) + 8 * (p & 1);
        const unsigned a0 = base + 256 * r0 + ((ch ^ tn_swz(r0)) << 4);
        const unsigned a1 = base + 256 * r1 + ((ch ^ tn_swz(r1)) << 4);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1));
    }
}

template <bool TNA, bool TNB>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_lp_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                              const bf16_t* __restrict__ B, int64_t ldb,
                                                              GemmEpilogue ep, int M, int N, int K,
                                                              int tiles_n, int splitk, float* slabs) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * LP_STAGE];   // 64 KB: 2 workgroups per CU
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = K / splitk;           // host guarantees per % LBK == 0
        k_begin = blockIdx.z * per;
        k_end = k_begin + per;
    }
    const int nk = (k_end - k_begin) / LBK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    Acc acc;
    acc_zero(acc);
    if (nk > 0) {
        lp_stage<TNA>(A, lda, m0, k_begin, lds, wave, lane);
        lp_stage<TNB>(B, ldb, n0, k_begin, lds + LP_OP, wave, lane);
        for (int kt = 0; kt < nk; ++kt) {
            // this wave's DMAs of tile kt have landed, then (barrier) everybody's; the same barrier
            // retires all reads of tile kt-1, whose stage is refilled right after it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int st = kt & 1;
            if (kt + 1 < nk) {
                const int kn = k_begin + (kt + 1) * LBK;
                lp_stage<TNA>(A, lda, m0, kn, lds + (st ^ 1) * LP_STAGE, wave, lane);
                lp_stage<TNB>(B, ldb, n0, kn, lds + (st ^ 1) * LP_STAGE + LP_OP, wave, lane);
            }
            const bf16_t* ta = lds + st * LP_STAGE;
            const bf16_t* tb = ta + LP_OP;
            // all 16 fragments of the stage are requested, then one wait: the other resident
            // waves' MFMAs cover the reads (interleaving reads per k-step measured 8-17 % slower)
            LpFrag<TNA> fa[4][2];
            LpFrag<TNB> fb[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    lp_frag<TNA>(ta, wm * 64 + i * 32, ks, lane, fa[ks][i]);
                    lp_frag<TNB>(tb, wn * 64 + i * 32, ks, lane, fb[ks][i]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc.t[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            fa[ks][i].value(), fb[ks][j].value(), acc.t[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                      // ring no longer in use: the epilogue stages through it
    gemm_epilogue_wide<2>(acc, ep, reinterpret_cast<float*>(lds), m0, n0, M, N, splitk, slabs);
}

// ---------------------------------------------------------------------------------
// 256 x 256 x 64 tiles, 16 waves (4 x 4, each the same 64 x 64 sub-tile as above), one workgroup
// per CU with a 2-stage ring of 64 KB stages (128 KB of LDS).  Why: the 128 x 128 kernel keeps
// 2 x 32 KB of operands in flight per CU; at ~1 us of DMA latency that bounds it near 1 PF however
// the loop is scheduled (slope of time against K at M = 32768: 0.137 us per k).  A 256 x 256 tile
// does twice the MFMA work per byte, so the same 64 KB in flight covers twice the rate.
// Operand images: NT tiles are [256 rows][64 k] (the small tile's image with more rows); TN
// tiles [64 k][256 x] with 512-B rows, chunk c of k-row r at c ^ tn_swz(r) (the swizzle permutes
// inside 256-B halves, which is what the transposing reads need).
constexpr int LPB = 256;                         // tile extent
constexpr int LPB_OP = LPB * LBK;                // bf16 elements per operand tile (32 KB)
constexpr int LPB_STAGE = 2 * LPB_OP;            // 64 KB

template <bool TN>
__device__ __forceinline__ void lpb_stage(const bf16_t* __restrict__ P, int64_t ld, int x0, int k0,
                                          bf16_t* tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave * 2 + i;              // 32 x 1 KiB per operand tile
        const bf16_t* src;
        if (!TN) {   // [x][k]: 8 rows x 128 B per instruction
            const int r = q * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            src = P + (int64_t)(x0 + r) * ld + k0 + c * 8;
        } else {     // [k][x]: 2 k-rows x 512 B per instruction
            const int r = q * 2 + (lane >> 5);
            const int c = (lane & 31) ^ tn_swz(r);
            src = P + (int64_t)(k0 + r) * ld + x0 + c * 8;
        }
        __builtin_amdgcn_global_load_lds((glb_ptr_lp)src, (lds_ptr_lp)(tile + q * 512), 16, 0, 0);
    }
}
template <bool TN>
__device__ __forceinline__ void lpb_frag(const bf16_t* tile, int x0, int ks, int lane, LpFrag<TN>& f) {
    if constexpr (!TN) {
        lp_frag<false>(tile, x0, ks, lane, f);   // same image, more rows
    } else {
        const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
        const int ch = ((x0 + 16 * (g & 1)) >> 3) + (p >> 1);
        const int r0 = ks * 16 + 8 * (g >> 1) + q;
        const int r1 = r0 + 4;
        const unsigned base = lds_addr_lp(tile) + 8 * (p & 1);
        const unsigned a0 = base + 512 * r0 + ((ch ^ tn_swz(r0)) << 4);
        const unsigned a1 = base + 512 * r1 + ((ch ^ tn_swz(r1)) << 4);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1));
    }
}

template <bool TNA, bool TNB>
__global__ __launch_bounds__(1024, 1) void gemm_lp_big_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                              const bf16_t* __restrict__ B, int64_t ldb,
                                                              GemmEpilogue ep, int M, int N, int K,
                                                              int tiles_n, int splitk, float* slabs) {
    extern __shared__ __attribute__((aligned(16))) bf16_t ldsb[];   // 2 x 64 KB
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * LPB, n0 = tn * LPB;
    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = K / splitk;
        k_begin = blockIdx.z * per;
        k_end = k_begin + per;
    }
    const int nk = (k_end - k_begin) / LBK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    Acc acc;
    acc_zero(acc);
    if (nk > 0) {
        lpb_stage<TNA>(A, lda, m0, k_begin, ldsb, wave, lane);
        lpb_stage<TNB>(B, ldb, n0, k_begin, ldsb + LPB_OP, wave, lane);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int st = kt & 1;
            if (kt + 1 < nk) {
                const int kn = k_begin + (kt + 1) * LBK;
                lpb_stage<TNA>(A, lda, m0, kn, ldsb + (st ^ 1) * LPB_STAGE, wave, lane);
                lpb_stage<TNB>(B, ldb, n0, kn, ldsb + (st ^ 1) * LPB_STAGE + LPB_OP, wave, lane);
            }
            const bf16_t* ta = ldsb + st * LPB_STAGE;
            const bf16_t* tb = ta + LPB_OP;
            // two k-steps of fragments at a time (4 waves per SIMD: 128 VGPRs each)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                LpFrag<TNA> fa[2][2];
                LpFrag<TNB> fb[2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        lpb_frag<TNA>(ta, wm * 64 + i * 32, 2 * h + ks, lane, fa[ks][i]);
                        lpb_frag<TNB>(tb, wn * 64 + i * 32, 2 * h + ks, lane, fb[ks][i]);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc.t[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                fa[ks][i].value(), fb[ks][j].value(), acc.t[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();                      // ring idle: 16 x 8 KB of epilogue staging
    gemm_epilogue_wave<2, Acc, 1>(acc, ep, reinterpret_cast<float*>(ldsb) + wave * (32 * 64), m0 + wm * 64,
                                  n0 + wn * 64, M, N, splitk, slabs);
}

// The 256 x 256 tile on v_mfma_f32_16x16x32_bf16 (QARIG_LP_MFMA16=1): same bytes, same LDS images,
// same cycles per FLOP as the 32x32x16 form; the chip is reported to hold a higher clock on this
// shape under load (MI355X_MICROARCH.md, DVFS give-back item 7), so both exist and wall time decides.
// A/B operand of one MFMA: lane l holds row (column) x0 + (l & 15), k = 32 ks + 8 (l >> 4) ... + 7.
template <bool TN, int PITCH = 512>
__device__ __forceinline__ void lp16_frag(const bf16_t* tile, int x0, int ks, int lane, LpFrag<TN>& f) {
    if constexpr (!TN) {
        const int r = x0 + (lane & 15);
        const int c = (ks * 4 + (lane >> 4)) ^ ((r >> 1) & 7);
        const unsigned a = lds_addr_lp(tile) + r * 128 + (c << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.v) : "v"(a));
    } else {   // PITCH-byte k-rows: two transposing reads of 4 k-rows x 16 columns per 16-lane group
        const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
        const int ch = (x0 >> 3) + (p >> 1);
        const int r0 = ks * 32 + 8 * g + q;
        const int r1 = r0 + 4;
        const unsigned base = lds_addr_lp(tile) + 8 * (p & 1);
        const unsigned a0 = base + PITCH * r0 + ((ch ^ tn_swz(r0)) << 4);
        const unsigned a1 = base + PITCH * r1 + ((ch ^ tn_swz(r1)) << 4);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1));
    }
}

// The 128 x 128 tile (4 waves, two workgroups per CU) on the same MFMA shape.
template <bool TNA, bool TNB>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_lp16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                const bf16_t* __restrict__ B, int64_t ldb,
                                                                GemmEpilogue ep, int M, int N, int K,
                                                                int tiles_n, int splitk, float* slabs) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * LP_STAGE];   // 64 KB: 2 workgroups per CU
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = K / splitk;
        k_begin = blockIdx.z * per;
        k_end = k_begin + per;
    }
    const int nk = (k_end - k_begin) / LBK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    Acc16 acc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc.t[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nk > 0) {
        lp_stage<TNA>(A, lda, m0, k_begin, lds, wave, lane);
        lp_stage<TNB>(B, ldb, n0, k_begin, lds + LP_OP, wave, lane);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int st = kt & 1;
            if (kt + 1 < nk) {
                const int kn = k_begin + (kt + 1) * LBK;
                lp_stage<TNA>(A, lda, m0, kn, lds + (st ^ 1) * LP_STAGE, wave, lane);
                lp_stage<TNB>(B, ldb, n0, kn, lds + (st ^ 1) * LP_STAGE + LP_OP, wave, lane);
            }
            const bf16_t* ta = lds + st * LP_STAGE;
            const bf16_t* tb = ta + LP_OP;
            LpFrag<TNA> fa[2][4];
            LpFrag<TNB> fb[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lp16_frag<TNA, 256>(ta, wm * 64 + i * 16, ks, lane, fa[ks][i]);
                    lp16_frag<TNB, 256>(tb, wn * 64 + i * 16, ks, lane, fb[ks][i]);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc.t[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i].value(), fb[ks][j].value(),
                                                                              acc.t[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    gemm_epilogue_wave<2, Acc16, 2>(acc, ep, reinterpret_cast<float*>(lds) + wave * (32 * 64), m0 + wm * 64,
                                    n0 + wn * 64, M, N, splitk, slabs);
}

template <bool TNA, bool TNB>
__global__ __launch_bounds__(1024, 1) void gemm_lp_big16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                const bf16_t* __restrict__ B, int64_t ldb,
                                                                GemmEpilogue ep, int M, int N, int K,
                                                                int tiles_n, int splitk, float* slabs) {
    extern __shared__ __attribute__((aligned(16))) bf16_t ldsm[];   // 2 x 64 KB
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * LPB, n0 = tn * LPB;
    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = K / splitk;
        k_begin = blockIdx.z * per;
        k_end = k_begin + per;
    }
    const int nk = (k_end - k_begin) / LBK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    Acc16 acc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc.t[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nk > 0) {
        lpb_stage<TNA>(A, lda, m0, k_begin, ldsm, wave, lane);
        lpb_stage<TNB>(B, ldb, n0, k_begin, ldsm + LPB_OP, wave, lane);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int st = kt & 1;
            if (kt + 1 < nk) {
                const int kn = k_begin + (kt + 1) * LBK;
                lpb_stage<TNA>(A, lda, m0, kn, ldsm + (st ^ 1) * LPB_STAGE, wave, lane);
                lpb_stage<TNB>(B, ldb, n0, kn, ldsm + (st ^ 1) * LPB_STAGE + LPB_OP, wave, lane);
            }
            const bf16_t* ta = ldsm + st * LPB_STAGE;
            const bf16_t* tb = ta + LPB_OP;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {           // 32-deep steps
                LpFrag<TNA> fa[4];
                LpFrag<TNB> fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lp16_frag<TNA>(ta, wm * 64 + i * 16, ks, lane, fa[i]);
                    lp16_frag<TNB>(tb, wn * 64 + i * 16, ks, lane, fb[i]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc.t[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i].value(), fb[j].value(),
                                                                              acc.t[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
    gemm_epilogue_wave<2, Acc16, 1>(acc, ep, reinterpret_cast<float*>(ldsm) + wave * (32 * 64), m0 + wm * 64,
                                    n0 + wn * 64, M, N, splitk, slabs);
}

// ---------------------------------------------------------------------------------
// fp8 (OCP e4m3) operands: BASELINE config 5 names the fp8 MFMA.  NT layout only (the forward
// products x W^T): A (M,K) and B (N,K) are bytes, per-tensor scaled by the cast kernels below;
// products on v_mfma_f32_32x32x64_f8f6f4 (64 deep per instruction, twice the bf16 rate), fp32
// accumulation, the accumulator multiplied by the two dequantisation factors in the shared
// epilogue.  A 128-deep k-tile is 128 bytes per row: the LDS image, its DMA and its chunk
// swizzle are exactly those of the bf16 NT tile (lp_stage<false> on the bytes).
// Fragment for k-step ks (64 deep): lane l holds the 32 bytes k = 64 ks + 32 (l >> 5) ... + 31 of
// row x0 + (l & 31) = two 16-B chunks.
typedef int i32x8_f8 __attribute__((ext_vector_type(8)));
typedef int i32x4_f8 __attribute__((ext_vector_type(4)));
struct F8Frag {
    i32x4_f8 lo, hi;
    __device__ __forceinline__ i32x8_f8 value() const {
        return i32x8_f8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
};
__device__ __forceinline__ void f8_frag(const bf16_t* tile, int x0, int ks, int lane, F8Frag& f) {
    const int r = x0 + (lane & 31);
    const int c0 = ks * 4 + (lane >> 5) * 2;
    const unsigned base = lds_addr_lp(tile) + r * 128;
    const unsigned a0 = base + (((c0) ^ ((r >> 1) & 7)) << 4), a1 = base + (((c0 + 1) ^ ((r >> 1) & 7)) << 4);
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.lo) : "v"(a0));
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.hi) : "v"(a1));
}

__global__ __launch_bounds__(NTHREADS, 2) void gemm_f8_kernel(const unsigned char* __restrict__ A, int64_t lda,
                                                              const unsigned char* __restrict__ B, int64_t ldb,
                                                              GemmEpilogue ep, int M, int N, int K, int tiles_n) {
    constexpr int FBK = 128;                                             // bytes = elements per k-tile
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * LP_STAGE];   // 64 KB: 2 workgroups per CU
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = K / FBK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // the byte matrices seen as 2-byte elements: leading dimensions and k offsets halve
    const bf16_t* A2 = reinterpret_cast<const bf16_t*>(A);
    const bf16_t* B2 = reinterpret_cast<const bf16_t*>(B);
    const int64_t lda2 = lda / 2, ldb2 = ldb / 2;

    Acc acc;
    acc_zero(acc);
    if (nk > 0) {
        lp_stage<false>(A2, lda2, m0, 0, lds, wave, lane);
        lp_stage<false>(B2, ldb2, n0, 0, lds + LP_OP, wave, lane);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int st = kt & 1;
            if (kt + 1 < nk) {
                const int kn = (kt + 1) * (FBK / 2);
                lp_stage<false>(A2, lda2, m0, kn, lds + (st ^ 1) * LP_STAGE, wave, lane);
                lp_stage<false>(B2, ldb2, n0, kn, lds + (st ^ 1) * LP_STAGE + LP_OP, wave, lane);
            }
            const bf16_t* ta = lds + st * LP_STAGE;
            const bf16_t* tb = ta + LP_OP;
            F8Frag fa[2][2], fb[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f8_frag(ta, wm * 64 + i * 32, ks, lane, fa[ks][i]);
                    f8_frag(tb, wn * 64 + i * 32, ks, lane, fb[ks][i]);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc.t[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            fa[ks][i].value(), fb[ks][j].value(), acc.t[i][j], 0, 0, 0, 0, 0, 0);
        }
    }
    __syncthreads();
    gemm_epilogue_wide<2>(acc, ep, reinterpret_cast<float*>(lds), m0, n0, M, N, 1, nullptr);
}

// The e4m3 product on the 256 x 256 tile of gemm_lp_big_kernel (128 bytes deep per k-tile: the
// same 64 KB stage images, lpb_stage<false> on the bytes).
__global__ __launch_bounds__(1024, 1) void gemm_f8_big_kernel(const unsigned char* __restrict__ A, int64_t lda,
                                                              const unsigned char* __restrict__ B, int64_t ldb,
                                                              GemmEpilogue ep, int M, int N, int K, int tiles_n) {
    constexpr int FBK = 128;
    extern __shared__ __attribute__((aligned(16))) bf16_t ldsf[];   // 2 x 64 KB
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * LPB, n0 = tn * LPB;
    const int nk = K / FBK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bf16_t* A2 = reinterpret_cast<const bf16_t*>(A);
    const bf16_t* B2 = reinterpret_cast<const bf16_t*>(B);
    const int64_t lda2 = lda / 2, ldb2 = ldb / 2;

    Acc acc;
    acc_zero(acc);
    if (nk > 0) {
        lpb_stage<false>(A2, lda2, m0, 0, ldsf, wave, lane);
        lpb_stage<false>(B2, ldb2, n0, 0, ldsf + LPB_OP, wave, lane);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int st = kt & 1;
            if (kt + 1 < nk) {
                const int kn = (kt + 1) * (FBK / 2);
                lpb_stage<false>(A2, lda2, m0, kn, ldsf + (st ^ 1) * LPB_STAGE, wave, lane);
                lpb_stage<false>(B2, ldb2, n0, kn, ldsf + (st ^ 1) * LPB_STAGE + LPB_OP, wave, lane);
            }
            const bf16_t* ta = ldsf + st * LPB_STAGE;
            const bf16_t* tb = ta + LPB_OP;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                F8Frag fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f8_frag(ta, wm * 64 + i * 32, ks, lane, fa[i]);
                    f8_frag(tb, wn * 64 + i * 32, ks, lane, fb[i]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc.t[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            fa[i].value(), fb[j].value(), acc.t[i][j], 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();
    gemm_epilogue_wave(acc, ep, reinterpret_cast<float*>(ldsf) + wave * (32 * 64), m0 + wm * 64, n0 + wn * 64,
                       M, N, 1, nullptr);
}

// |x| maximum of a tensor as the bit pattern of a non-negative float (they order like
// unsigned integers): *amax_bits must be zero before the launch.
// With `bf` the bf16 copy of x (what the backward products read) is written in the same pass.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n4,
                                                   unsigned* __restrict__ amax_bits, uint2* __restrict__ bf) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (bf) bf[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
    m = wave_max(m);
    __shared__ float wmax[4];              // one atomic per workgroup: they all hit one L2 line
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}
// dst = e4m3(src * 448 / amax) (round to nearest even, saturating), *inv_scale = amax / 448: the
// factor that takes products of the quantised values back.  amax = 0 quantises with scale 1.
constexpr float F8_MAX = 448.0f;
__global__ __launch_bounds__(256) void cast_fp8_kernel(const float* __restrict__ src, int64_t n4,
                                                       const unsigned* __restrict__ amax_bits,
                                                       unsigned* __restrict__ dst, float* __restrict__ inv_scale) {
    const float amax = __uint_as_float(*amax_bits);
    const float scale = amax > 0.0f ? F8_MAX / amax : 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = amax > 0.0f ? amax / F8_MAX : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        int p = __builtin_amdgcn_cvt_pk_fp8_f32(v.x * scale, v.y * scale, 0, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(v.z * scale, v.w * scale, p, true);
        dst[i] = (unsigned)p;
    }
}

// dst[i] = bf16(src[i]), round to nearest even; n % 8 == 0, 16-B aligned.
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i];
        const float4 b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        reinterpret_cast<uint4*>(dst)[i] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w),
                                                      pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
    }
}
__global__ void cast_bf16_tail_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                      int64_t begin, int64_t n) {
    const int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (bf16_t)(pack_bf16x2(src[i], 0.0f) & 0xffffu);
}

// Cast with the column sums riding along: dst = bf16(src) for src (M, N) fp32 and
// part[blockIdx.y][n] = sum of the block's CS_ROWS rows of column n (fixed order).  The bias
// gradient of a Linear layer is the column sum of the same dT that the weight-gradient GEMM
// needs in bf16, so one pass over dT yields both.
constexpr int CS_ROWS = 64;     // rows per block: 4 row groups of 16, combined in group order
__global__ __launch_bounds__(512) void cast_colsum_kernel(const float* __restrict__ src, int64_t ld,
                                                          int M, int N, bf16_t* __restrict__ dst,
                                                          float* __restrict__ part) {
    __shared__ float4 red[3][128];
    const int ct = threadIdx.x & 127, rg = threadIdx.x >> 7;
    const int c = (blockIdx.x * 128 + ct) * 4;
    const bool in = c < N;
    const int r0 = blockIdx.y * CS_ROWS + rg * 16, r1 = min(M, r0 + 16);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in)
        for (int r = r0; r < r1; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + c);
            if (dst)
                *reinterpret_cast<uint2*>(dst + (int64_t)r * N + c) =
                    make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    if (rg > 0) red[rg - 1][ct] = s;
    __syncthreads();
    if (rg == 0 && in) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4 t = red[g][ct];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * N + c) = s;
    }
}
// the same sums from a bf16 tensor (dT1 of an MLP exists only in bf16 in this mode)
__global__ __launch_bounds__(512) void colsum_bf16_kernel(const bf16_t* __restrict__ src, int64_t ld,
                                                          int M, int N, float* __restrict__ part) {
    __shared__ float4 red[3][128];
    const int ct = threadIdx.x & 127, rg = threadIdx.x >> 7;
    const int c = (blockIdx.x * 128 + ct) * 4;
    const bool in = c < N;
    const int r0 = blockIdx.y * CS_ROWS + rg * 16, r1 = min(M, r0 + 16);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in)
        for (int r = r0; r < r1; ++r) {
            const uint2 v = *reinterpret_cast<const uint2*>(src + (int64_t)r * ld + c);
            s.x += bf16_bits_to_f32(v.x & 0xffffu); s.y += bf16_bits_to_f32(v.x >> 16);
            s.z += bf16_bits_to_f32(v.y & 0xffffu); s.w += bf16_bits_to_f32(v.y >> 16);
        }
    if (rg > 0) red[rg - 1][ct] = s;
    __syncthreads();
    if (rg == 0 && in) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4 t = red[g][ct];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * N + c) = s;
    }
}
// out[n] (+)= sum over the `chunks` partial rows, in a fixed order: 4 interleaved chunk groups per
// column summed in parallel, then combined in group order (a serial walk over 128+ chunks is
// latency-bound: 15 us per bias gradient)
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int chunks,
                                                            int N, float* __restrict__ out, int accumulate) {
    __shared__ float red[3][64];
    const int ct = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + ct;
    float s = 0.0f;
    if (c < N)
        for (int z0 = g; z0 < chunks; z0 += 32) {      // 8 loads in flight, added in chunk order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int z = z0 + 4 * u;
                v[u] = z < chunks ? part[(int64_t)z * N + c] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    if (g > 0) red[g - 1][ct] = s;
    __syncthreads();
    if (g == 0 && c < N) {
        s = ((s + red[0][ct]) + red[1][ct]) + red[2][ct];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// dst[c][r] = bf16(src[r][c]) for src (R, C) row-major with leading dimension lds_: the W^T
// shadow of a Linear weight.  64 x 64 tiles through LDS, coalesced on both sides.
__global__ __launch_bounds__(256) void cast_transpose_bf16_kernel(const float* __restrict__ src,
                                                                  int64_t ld, int R, int C,
                                                                  bf16_t* __restrict__ dst) {
    __shared__ float t[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        t[r][c] = (r0 + r < R && c0 + c < C) ? src[(int64_t)(r0 + r) * ld + c0 + c] : 0.0f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
        const int c = idx >> 5, r = (idx & 31) * 2;
        if (c0 + c < C && r0 + r < R) {
            const uint32_t v = pack_bf16x2(t[r][c], t[r + 1][c]);
            if (r0 + r + 1 < R && ((R & 1) == 0))
                *reinterpret_cast<uint32_t*>(dst + (int64_t)(c0 + c) * R + r0 + r) = v;
            else {
                dst[(int64_t)(c0 + c) * R + r0 + r] = (bf16_t)(v & 0xffffu);
                if (r0 + r + 1 < R) dst[(int64_t)(c0 + c) * R + r0 + r + 1] = (bf16_t)(v >> 16);
            }
        }
    }
}
}  // namespace qarig

using namespace qarig;

extern "C" int qarig_slab_reduce_f32(const float* slabs, float* out, int64_t ldc, int M, int N,
                                     int nslab, int accumulate, void* stream);

// fp32 -> bf16 (round to nearest even) of n contiguous elements; dst holds n 16-bit values.
// New entry (no reference counterpart): the operand conversion of the reduced-precision mode.
extern "C" int qarig_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
    QARIG_CHECK_ARG(src && dst && n > 0 && n <= (1LL << 40), "cast_bf16: bad arguments");
    QARIG_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "cast_bf16: 16-B aligned buffers");
    const int64_t n8 = n / 8;
    if (n8 > 0) {
        int blocks = (int)((n8 + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                           (bf16_t*)dst, n8);
    }
    if (n8 * 8 < n)
        hipLaunchKernelGGL(cast_bf16_tail_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src,
                           (bf16_t*)dst, n8 * 8, n);
    QARIG_CHECK_LAUNCH("cast_bf16");
    return QARIG_OK;
}

// dst (C, R) bf16 = transpose of src (R, C) fp32 (row stride ld).
extern "C" int qarig_cast_transpose_bf16(const float* src, int64_t ld, int R, int C, void* dst,
                                         void* stream) {
    QARIG_CHECK_ARG(src && dst && R > 0 && C > 0 && ld >= C, "cast_transpose_bf16: bad arguments");
    QARIG_CHECK_DIMS("cast_transpose_bf16", R, C);
    QARIG_CHECK_ARG(((uintptr_t)dst & 3) == 0, "cast_transpose_bf16: 4-B aligned destination");
    hipLaunchKernelGGL(cast_transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0,
                       (hipStream_t)stream, src, ld, R, C, (bf16_t*)dst);
    QARIG_CHECK_LAUNCH("cast_transpose_bf16");
    return QARIG_OK;
}

extern "C" size_t qarig_cast_colsum_workspace_bytes(int M, int N) {
    if (M < 1 || N < 1 || M > (1 << 24) || N > (1 << 24)) return 0;
    return (size_t)((M + CS_ROWS - 1) / CS_ROWS) * N * sizeof(float);
}

// colsum[n] (+)= sum_m src[m][n], and (dst != NULL) dst = bf16(src); src fp32 (src_is_bf16 = 0) or
// bf16 (1; dst must be NULL).  N % 4 == 0, rows 16-B (8-B for bf16) aligned.  The bias gradient
// of nn.Linear (models/layers.py:243-250) in the reduced-precision mode.
extern "C" int qarig_cast_colsum(const void* src, int64_t ld, int src_is_bf16, int M, int N, void* dst,
                                 float* colsum, int accumulate, void* workspace, size_t ws_bytes,
                                 void* stream) {
    QARIG_CHECK_ARG(src && colsum && M > 0 && N > 0 && N % 4 == 0 && ld >= N && ld % 4 == 0,
                    "cast_colsum: bad arguments");
    QARIG_CHECK_DIMS("cast_colsum", M, N);
    QARIG_CHECK_ARG(!(src_is_bf16 && dst), "cast_colsum: a bf16 source is not re-cast");
    QARIG_CHECK_ARG(((uintptr_t)src & (src_is_bf16 ? 7 : 15)) == 0 && (!dst || ((uintptr_t)dst & 7) == 0),
                    "cast_colsum: misaligned buffers");
    if (!workspace || ws_bytes < qarig_cast_colsum_workspace_bytes(M, N)) {
        qarig_set_error("cast_colsum: workspace too small");
        return QARIG_ERR_WORKSPACE;
    }
    const int chunks = (M + CS_ROWS - 1) / CS_ROWS;
    dim3 grid((N / 4 + 127) / 128, chunks), block(512);
    hipStream_t st = (hipStream_t)stream;
    if (src_is_bf16)
        hipLaunchKernelGGL(colsum_bf16_kernel, grid, block, 0, st, (const bf16_t*)src, ld, M, N, (float*)workspace);
    else
        hipLaunchKernelGGL(cast_colsum_kernel, grid, block, 0, st, (const float*)src, ld, M, N, (bf16_t*)dst,
                           (float*)workspace);
    QARIG_CHECK_LAUNCH("cast_colsum");
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 63) / 64), dim3(256), 0, st, (const float*)workspace,
                       chunks, N, colsum, accumulate);
    QARIG_CHECK_LAUNCH("cast_colsum reduce");
    return QARIG_OK;
}

extern "C" size_t qarig_gemm_lp_workspace_bytes(int M, int N, int splitk) {
    if (M < 1 || N < 1 || splitk > (1 << 16)) return 0;
    return splitk > 1 ? (size_t)splitk * M * N * sizeof(float) : 0;
}

// 1 when the shape can run on the reduced-precision kernel (else the caller uses qarig_gemm_f32).
extern "C" int qarig_gemm_lp_supported(int M, int N, int K, int splitk) {
    if (splitk < 1) splitk = 1;
    if (!qarig_dims_ok({M, N}) || !qarig_dims_ok({M, K}) || !qarig_dims_ok({N, K}) || splitk > 4096) return 0;
    return M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % splitk == 0 &&
           (K / splitk) % LBK == 0;
}

// C (fp32, optional) / Cb (bf16, optional) = epilogue(A B^T) with bf16 operands.
// layout 0 = NT (A (M,K), B (N,K)), 1 = TN (A (K,M), B (K,N)); lda/ldb in elements.
// Epilogue arguments as qarig_gemm_f32 (bias, residual, preact, act, gradz/gact, splitk,
// accumulate); Cb / Pb: optional bf16 copies of the output / of the saved pre-activation.
extern "C" int qarig_gemm_lp(const void* A, int64_t lda, const void* B, int64_t ldb, int layout,
                             float* C, int64_t ldc, int M, int N, int K, const float* bias,
                             const float* residual, int64_t ldr, float* preact, int64_t ldp, int act,
                             const void* gradz, int64_t ldz, int gradz_is_bf16, int gact, int splitk,
                             int accumulate, void* Cb, int64_t ldcb, void* Pb, int64_t ldpb,
                             void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(A && B && (C || Cb), "gemm_lp: null operand");
    QARIG_CHECK_ARG(layout >= 0 && layout <= 2, "gemm_lp: layout must be 0 (NT), 1 (TN) or 2 (NN)");
    QARIG_CHECK_ARG(act >= 0 && act <= 3 && gact >= 0 && gact <= 3, "gemm_lp: bad activation id");
    if (splitk < 1) splitk = 1;
    QARIG_CHECK_ARG(qarig_gemm_lp_supported(M, N, K, splitk),
                    "gemm_lp: needs M,N %% 128 == 0 and K/splitk %% 64 == 0 (M=%d N=%d K=%d splitk=%d)",
                    M, N, K, splitk);
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    QARIG_CHECK_ARG(al16(A) && al16(B) && lda % 8 == 0 && ldb % 8 == 0, "gemm_lp: operands 16-B aligned, ld %% 8");
    auto ok4 = [&](const void* p, int64_t ld) { return !p || (al16(p) && ld % 4 == 0); };
    QARIG_CHECK_ARG(ok4(C, ldc) && ok4(bias, 4) && ok4(residual, ldr) && ok4(preact, ldp) &&
                        (gradz_is_bf16 ? (!gradz || (((uintptr_t)gradz & 7) == 0 && ldz % 4 == 0)) : ok4(gradz, ldz)),
                    "gemm_lp: fp32 epilogue tensors 16-B aligned, ld %% 4");
    QARIG_CHECK_ARG((!Cb || (((uintptr_t)Cb & 7) == 0 && ldcb % 4 == 0)) &&
                        (!Pb || (((uintptr_t)Pb & 7) == 0 && ldpb % 4 == 0)),
                    "gemm_lp: bf16 outputs 8-B aligned, ld %% 4");
    if (accumulate) {
        QARIG_CHECK_ARG(C && !bias && !residual && !preact && !gradz && act == ACT_NONE && !Cb && !Pb,
                        "gemm_lp: accumulate supports the plain epilogue only");
        if (splitk == 1) { residual = C; ldr = ldc; }
    }
    if (splitk > 1) {
        QARIG_CHECK_ARG(C && !bias && !residual && !preact && !gradz && act == ACT_NONE && !Cb && !Pb,
                        "gemm_lp: split-K supports the plain epilogue only");
        if (!workspace || ws_bytes < qarig_gemm_lp_workspace_bytes(M, N, splitk)) {
            qarig_set_error("gemm_lp: workspace too small");
            return QARIG_ERR_WORKSPACE;
        }
    }
    const int tiles_m = M / BM, tiles_n = N / BN;
    dim3 grid(tiles_m * tiles_n, 1, splitk), block(NTHREADS);
    GemmEpilogue ep{C, ldc, bias, residual, ldr, preact, ldp, act,
                    gradz_is_bf16 ? nullptr : (const float*)gradz, ldz, gact, nullptr,
                    (unsigned short*)Cb, ldcb, (unsigned short*)Pb, ldpb,
                    gradz_is_bf16 ? (const unsigned short*)gradz : nullptr, ldz};
    hipStream_t st = (hipStream_t)stream;
    // 256 x 256 tiles where they still give every CU a workgroup; option lp_big = 0 / 1 overrides
    const int big_env = g_qarig_opt.lp_big;
    const long big_tiles = (long)(M / LPB) * (N / LPB) * splitk;
    if (M % LPB == 0 && N % LPB == 0 && big_env != 0 && (big_env == 1 || big_tiles >= 224)) {
        constexpr int BIG_LDS = 2 * LPB_STAGE * (int)sizeof(bf16_t);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)gemm_lp_big_kernel<false, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
            (void)hipFuncSetAttribute((const void*)gemm_lp_big_kernel<true, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
            (void)hipFuncSetAttribute((const void*)gemm_lp_big_kernel<false, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
            attr_set = true;
        }
        dim3 gridb((M / LPB) * (N / LPB), 1, splitk), blockb(1024);
        const int tnb = N / LPB;
        const int m16_env = g_qarig_opt.lp_mfma16;
        if (m16_env) {
            static bool attr16 = false;
            if (!attr16) {
                (void)hipFuncSetAttribute((const void*)gemm_lp_big16_kernel<false, false>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_lp_big16_kernel<true, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_lp_big16_kernel<false, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
                attr16 = true;
            }
            if (layout == 0)
                hipLaunchKernelGGL((gemm_lp_big16_kernel<false, false>), gridb, blockb, BIG_LDS, st, (const bf16_t*)A,
                                   lda, (const bf16_t*)B, ldb, ep, M, N, K, tnb, splitk, (float*)workspace);
            else if (layout == 1)
                hipLaunchKernelGGL((gemm_lp_big16_kernel<true, true>), gridb, blockb, BIG_LDS, st, (const bf16_t*)A,
                                   lda, (const bf16_t*)B, ldb, ep, M, N, K, tnb, splitk, (float*)workspace);
            else
                hipLaunchKernelGGL((gemm_lp_big16_kernel<false, true>), gridb, blockb, BIG_LDS, st, (const bf16_t*)A,
                                   lda, (const bf16_t*)B, ldb, ep, M, N, K, tnb, splitk, (float*)workspace);
            QARIG_CHECK_LAUNCH("gemm_lp big16");
            if (splitk > 1)
                return qarig_slab_reduce_f32((const float*)workspace, C, ldc, M, N, splitk, accumulate, stream);
            return QARIG_OK;
        }
        if (layout == 0)
            hipLaunchKernelGGL((gemm_lp_big_kernel<false, false>), gridb, blockb, BIG_LDS, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, ep, M, N, K, tnb, splitk, (float*)workspace);
        else if (layout == 1)
            hipLaunchKernelGGL((gemm_lp_big_kernel<true, true>), gridb, blockb, BIG_LDS, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, ep, M, N, K, tnb, splitk, (float*)workspace);
        else
            hipLaunchKernelGGL((gemm_lp_big_kernel<false, true>), gridb, blockb, BIG_LDS, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, ep, M, N, K, tnb, splitk, (float*)workspace);
        QARIG_CHECK_LAUNCH("gemm_lp big");
        if (splitk > 1)
            return qarig_slab_reduce_f32((const float*)workspace, C, ldc, M, N, splitk, accumulate, stream);
        return QARIG_OK;
    }
    const int m16_small = g_qarig_opt.lp_mfma16;
    if (m16_small) {
        if (layout == 0)
            hipLaunchKernelGGL((gemm_lp16_kernel<false, false>), grid, block, 0, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, ep, M, N, K, tiles_n, splitk, (float*)workspace);
        else if (layout == 1)
            hipLaunchKernelGGL((gemm_lp16_kernel<true, true>), grid, block, 0, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, ep, M, N, K, tiles_n, splitk, (float*)workspace);
        else
            hipLaunchKernelGGL((gemm_lp16_kernel<false, true>), grid, block, 0, st, (const bf16_t*)A, lda,
                               (const bf16_t*)B, ldb, ep, M, N, K, tiles_n, splitk, (float*)workspace);
    } else if (layout == 0)
        hipLaunchKernelGGL((gemm_lp_kernel<false, false>), grid, block, 0, st, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, ep, M, N, K, tiles_n, splitk, (float*)workspace);
    else if (layout == 1)
        hipLaunchKernelGGL((gemm_lp_kernel<true, true>), grid, block, 0, st, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, ep, M, N, K, tiles_n, splitk, (float*)workspace);
    else
        hipLaunchKernelGGL((gemm_lp_kernel<false, true>), grid, block, 0, st, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, ep, M, N, K, tiles_n, splitk, (float*)workspace);
    QARIG_CHECK_LAUNCH("gemm_lp");
    if (splitk > 1)
        return qarig_slab_reduce_f32((const float*)workspace, C, ldc, M, N, splitk, accumulate, stream);
    return QARIG_OK;
}

// ---- fp8 (e4m3) operands: see gemm_f8_kernel ------------------------------------------------
extern "C" int qarig_gemm_f8_supported(int M, int N, int K) {
    return M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % 128 == 0;
}

// dst (n bytes, e4m3) = quantised src (n floats) with one scale for the tensor; scratch[0] receives
// the |x| maximum (as float bits; zeroed here), inv_scale[0] the dequantisation factor; bf16_dst
// (optional, n elements) the bf16 copy of src, written by the pass that finds the maximum.
extern "C" int qarig_cast_fp8(const float* src, int64_t n, void* dst, float* inv_scale, void* scratch,
                              void* bf16_dst, void* stream) {
    QARIG_CHECK_ARG(src && dst && inv_scale && scratch, "cast_fp8: null pointer");
    QARIG_CHECK_ARG(((uintptr_t)bf16_dst & 7) == 0, "cast_fp8: bf16 copy 8-B aligned");
    QARIG_CHECK_ARG(n > 0 && n % 4 == 0 && n < (1LL << 40), "cast_fp8: n must be a positive multiple of 4");
    QARIG_CHECK_ARG((((uintptr_t)src & 15) | ((uintptr_t)dst & 3) | ((uintptr_t)scratch & 3)) == 0,
                    "cast_fp8: src 16-B aligned, dst 4-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
    if (hipMemsetAsync(scratch, 0, 4, st) != hipSuccess) {
        qarig_set_error("cast_fp8: memset failed");
        return QARIG_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, st, src, n4, (unsigned*)scratch,
                       (uint2*)bf16_dst);
    QARIG_CHECK_LAUNCH("cast_fp8 amax");
    hipLaunchKernelGGL(cast_fp8_kernel, dim3(blocks), dim3(256), 0, st, src, n4, (const unsigned*)scratch,
                       (unsigned*)dst, inv_scale);
    QARIG_CHECK_LAUNCH("cast_fp8");
    return QARIG_OK;
}

// C[M,N] = epilogue(inv_a * inv_b * sum_k A8[m][k] B8[n][k]): A8 (M,K), B8 (N,K) e4m3 bytes,
// inv_a / inv_b the device scalars qarig_cast_fp8 wrote.  Epilogue options as qarig_gemm_lp
// (no backward fusion, no split-K, no accumulate).
extern "C" int qarig_gemm_f8(const void* A, int64_t lda, const void* B, int64_t ldb, const float* inv_a,
                             const float* inv_b, float* C, int64_t ldc, int M, int N, int K,
                             const float* bias, const float* residual, int64_t ldr, float* preact,
                             int64_t ldp, int act, void* Cb, int64_t ldcb, void* Pb, int64_t ldpb,
                             void* stream) {
    QARIG_CHECK_ARG(A && B && inv_a && inv_b && (C || Cb), "gemm_f8: null operand");
    QARIG_CHECK_ARG(act >= 0 && act <= 3, "gemm_f8: bad activation id");
    QARIG_CHECK_ARG(qarig_gemm_f8_supported(M, N, K),
                    "gemm_f8: needs M,N %% 128 == 0 and K %% 128 == 0 (M=%d N=%d K=%d)", M, N, K);
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    QARIG_CHECK_ARG(al16(A) && al16(B) && lda % 16 == 0 && ldb % 16 == 0, "gemm_f8: operands 16-B aligned, ld %% 16");
    auto ok4 = [&](const void* p, int64_t ld) { return !p || (al16(p) && ld % 4 == 0); };
    QARIG_CHECK_ARG(ok4(C, ldc) && ok4(bias, 4) && ok4(residual, ldr) && ok4(preact, ldp),
                    "gemm_f8: fp32 epilogue tensors 16-B aligned, ld %% 4");
    QARIG_CHECK_ARG((!Cb || (((uintptr_t)Cb & 7) == 0 && ldcb % 4 == 0)) &&
                        (!Pb || (((uintptr_t)Pb & 7) == 0 && ldpb % 4 == 0)),
                    "gemm_f8: bf16 outputs 8-B aligned, ld %% 4");
    const int tiles_m = M / BM, tiles_n = N / BN;
    GemmEpilogue ep{C, ldc, bias, residual, ldr, preact, ldp, act, nullptr, 0, 0, nullptr,
                    (unsigned short*)Cb, ldcb, (unsigned short*)Pb, ldpb, nullptr, 0, inv_a, inv_b};
    const int big_env = g_qarig_opt.lp_big;
    if (M % LPB == 0 && N % LPB == 0 && big_env != 0 && (big_env == 1 || (long)(M / LPB) * (N / LPB) >= 224)) {
        constexpr int BIG_LDS = 2 * LPB_STAGE * (int)sizeof(bf16_t);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)gemm_f8_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      BIG_LDS);
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_f8_big_kernel, dim3((M / LPB) * (N / LPB)), dim3(1024), BIG_LDS,
                           (hipStream_t)stream, (const unsigned char*)A, lda, (const unsigned char*)B, ldb, ep, M,
                           N, K, N / LPB);
        QARIG_CHECK_LAUNCH("gemm_f8 big");
        return QARIG_OK;
    }
    hipLaunchKernelGGL(gemm_f8_kernel, dim3(tiles_m * tiles_n), dim3(NTHREADS), 0, (hipStream_t)stream,
                       (const unsigned char*)A, lda, (const unsigned char*)B, ldb, ep, M, N, K, tiles_n);
    QARIG_CHECK_LAUNCH("gemm_f8");
    return QARIG_OK;
}
