// fp32 GEMM on the bf16 matrix pipe (opt-in: option gemm_x3; reported under its own dtype tag, never as the fp32
// headline).  Same operands, same epilogue and same results-within-tolerance as the fp32-MFMA kernels of gemm.hip
// for the Linear products (reference models/layers.py:234-254, 330-340, 389-418): both operands stay fp32 in HBM;
// on their way into LDS every value is split into three bf16 pieces that add up to it EXACTLY (3 x 8 significant
// bits: h = bf16(v), m = bf16(v - h), l = bf16(v - h - m), as the BMU coarse pass does), and
//     x w = hh + hm + mh + hl + lh + mm      (dropped: ml, lm, ll <= 3 x 2^-24 |x||w|)
// is six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block, accumulated in fp32.  The bf16 MFMA runs 16 x the fp32
// MFMA's rate, so six products are 2.7 x the fp32 peak in matrix cycles, and -- unlike the fp32 MFMA on this chip --
// it co-executes with the vector ALU that does the splitting.  Accuracy, measured (tools/gemm_x3_probe.hip,
// profiles/r04_gemm_x3_probe.log): the matrix core adds the 16 products of an instruction in a wide adder and rounds
// once, so this chain rounds K / 16 x 6 times where the fp32 fma chain rounds K times: max|err| / max|ref| against
// fp64 8.3e-7 (rms 9.0e-8) at K = 512 where the fp32 chain has 5.1e-7 (1.1e-7); 2.1e-6 (4.0e-7) against 2.3e-6
// (4.5e-7) at K = 8192 -- inside the 2e-6 sqrt(K / 512) the parity tests ask of the fp32 kernels, which
// tests/test_gpu_switches.py runs under this option.  Non-finite operands give NaN where the fp32 chain gives +-inf
// (inf - inf in the split).
//
// 128 x 128 tile, 256 threads = 4 waves in a 2 x 2 arrangement of 64 x 64 (2 x 2 accumulators of 32 x 32), k-tiles
// of 32.  Threads 0-127 stage A, 128-255 stage B: 8 float4 loads per thread and k-tile (a row's 32 k for a
// reduction-contiguous operand [X][K]; 8 k-rows of 4 consecutive x for a tile-contiguous one [K][X]) land in
// registers one whole k-tile before they are split (11 vector instructions per pair of values, under the MFMAs) and
// written as 16-B fragment units -- (row, 8 consecutive k) of one piece -- in the order the MFMA fragments are read:
// [piece][32-row tile][16-k step][k half][row], so a fragment is ONE conflict-free ds_read_b128.  One LDS stage
// (48 KB) and two barriers per k-tile, THREE workgroups per CU (<= 168 registers).  Per 16-k step a wave reads 12
// fragments for 24 MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qarig_common.h"
#include "gemm_epilogue.h"

namespace qarig {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int x3_u32x4 __attribute__((ext_vector_type(4)));

constexpr int XK = 32;                       // reduction depth per staged tile
constexpr int X3_PLANE = 4 * 2 * 2 * 32;     // 16-B units of one piece of one operand tile: [tile][k step][k half][row]
constexpr int X3_OP = 3 * X3_PLANE;          // ... of an operand tile (24 KB)
constexpr int X3_STAGE = 2 * X3_OP;          // A then B (48 KB)
constexpr int X3_LDS = X3_STAGE * 16;        // ONE stage: two workgroups per CU (see the kernel)

__device__ __forceinline__ uint32_t x3_pack(float a, float b) {   // a in the low half
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
}
// (a, b) = H + M + L piecewise (round-to-nearest remainders)
__device__ __forceinline__ void x3_split(float a, float b, uint32_t& H, uint32_t& M, uint32_t& L) {
    H = x3_pack(a, b);
    const float ra = a - __uint_as_float(H << 16), rb = b - __uint_as_float(H & 0xffff0000u);
    M = x3_pack(ra, rb);
    L = x3_pack(ra - __uint_as_float(M << 16), rb - __uint_as_float(M & 0xffff0000u));
}

// The 32 values one staging thread holds of an operand tile, as 8 float4.
//   KC ([X][K]): thread x = row x of the tile, f[j] = k 4j .. 4j+3
//   XC ([K][X]): thread (q = t & 31, o = t >> 5): f[j] = rows 4q .. 4q+3 at k = 8 o + j
template <bool KC>
__device__ __forceinline__ void x3_load(f32x4 (&f)[8], const float* __restrict__ P, int64_t ld, int x0, int k0, int t) {
    // (vector-typed values: as HIP float4 structs the staged values were taken apart into scalars and, in the
    //  kernels whose two operands differ in layout, loaded dword by dword: 3.25 x the load instructions)
    if (KC) {
        const f32x4* p = reinterpret_cast<const f32x4*>(P + (int64_t)(x0 + t) * ld + k0);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = p[j];
    } else {
        const float* p = P + (int64_t)(k0 + 8 * (t >> 5)) * ld + x0 + 4 * (t & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = *reinterpret_cast<const f32x4*>(p + (int64_t)j * ld);
    }
}
// Where row r of a 32-row tile sits inside its block of 32 units: r with bit 1 ^= bit 3 and bit 0 ^= bit 4.  Three
// access patterns must be conflict-free (the guide's LDS table: ds_write_b128 is served in groups of 8 contiguous
// lanes over 32 banks, ds_read_b128 in the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} over 64):
//   * the fragment reads and the row-per-thread writers (lane = row r): 8 contiguous r keep distinct positions
//     mod 8 (the low three bits are XORed with a constant of the group), and both read groups cover every
//     position mod 16 once;
//   * the tile-contiguous writers hold rows 4a .. 4a+3 and store row 4a + i with instruction i: its 8 lanes
//     a = 0..7 land on positions whose low bits are (a & 1, ((a >> 1) & 1) ^ i1, (a >> 2) ^ i0 ...) -- distinct mod 8;
//     with the identity they hit two bank groups four times each (SQ_LDS_BANK_CONFLICT 37.7 M cycles per launch
//     at 16384 x 512 x 2048).
__device__ __forceinline__ int x3_pos(int tile, int r) { return r ^ ((r >> 2) & 2) ^ (r >> 4); }

template <int PL = X3_PLANE>
__device__ __forceinline__ void x3_put(x3_u32x4* op, int unit, const float (&v)[8]) {
    x3_u32x4 H, M, L;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t h, m, l;
        x3_split(v[2 * q], v[2 * q + 1], h, m, l);
        H[q] = h; M[q] = m; L[q] = l;
    }
    op[unit] = H;
    op[PL + unit] = M;
    op[2 * PL + unit] = L;
}
template <bool KC>
__device__ __forceinline__ void x3_store(x3_u32x4* op, const f32x4 (&f)[8], int t) {
    if (KC) {
        const int tile = t >> 5, r = t & 31;
#pragma unroll
        for (int u = 0; u < 4; ++u) {                      // u = 2 ks + k half: k = 8u .. 8u+7
            const float v[8] = {f[2 * u][0], f[2 * u][1], f[2 * u][2], f[2 * u][3],
                                f[2 * u + 1][0], f[2 * u + 1][1], f[2 * u + 1][2], f[2 * u + 1][3]};
            x3_put(op, (tile * 4 + u) * 32 + x3_pos(tile, r), v);
        }
    } else {
        const int q = t & 31, o = t >> 5;                  // o = 2 ks + k half
        const int tile = q >> 3, r0 = (4 * q) & 31;
        {
            const float v[8] = {f[0][0], f[1][0], f[2][0], f[3][0], f[4][0], f[5][0], f[6][0], f[7][0]};
            x3_put(op, (tile * 4 + o) * 32 + x3_pos(tile, r0 + 0), v);
        }
        {
            const float v[8] = {f[0][1], f[1][1], f[2][1], f[3][1], f[4][1], f[5][1], f[6][1], f[7][1]};
            x3_put(op, (tile * 4 + o) * 32 + x3_pos(tile, r0 + 1), v);
        }
        {
            const float v[8] = {f[0][2], f[1][2], f[2][2], f[3][2], f[4][2], f[5][2], f[6][2], f[7][2]};
            x3_put(op, (tile * 4 + o) * 32 + x3_pos(tile, r0 + 2), v);
        }
        {
            const float v[8] = {f[0][3], f[1][3], f[2][3], f[3][3], f[4][3], f[5][3], f[6][3], f[7][3]};
            x3_put(op, (tile * 4 + o) * 32 + x3_pos(tile, r0 + 3), v);
        }
    }
}

// One 128 x 128 output tile: rows m0.., columns n0.., reduction [k_begin, k_begin + nk * XK).  rs_dst: where the
// tile's 128 row sums of A go (null: none).  ep_splitk / ep_slabs: what the epilogue is told (slabs when > 1).
template <bool AKC, bool BKC>
__device__ __forceinline__ void x3_tile(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                        int64_t ldb, const GemmEpilogue& ep, int M, int N, int m0, int n0,
                                        int k_begin, int nk, float* __restrict__ rs_dst, int ep_splitk,
                                        float* __restrict__ ep_slabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char x3_smem[];
    x3_u32x4* lds = reinterpret_cast<x3_u32x4*>(x3_smem);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool stage_a = wave < 2;                // waves 0-1 stage A, 2-3 stage B (a scalar branch: `wave` is an SGPR)
    const int ts = t & 127;
    Acc acc;
    acc_zero(acc);
    // sum_k A(m, k) of this split (the bias gradient riding on a weight-gradient product: tile-contiguous A, column
    // tiles 0 only): a staging thread keeps the sums of its four rows over its k rows
    const bool do_rs = !AKC && rs_dst != nullptr && stage_a;
    float4 rs4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x4 f[8];
    auto load = [&](int kt) {
        const int k0 = k_begin + kt * XK;
        if (stage_a) x3_load<AKC>(f, A, lda, m0, k0, ts);
        else x3_load<BKC>(f, B, ldb, n0, k0, ts);
    };
    auto store = [&]() {
        if (stage_a) {
            if (do_rs) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { rs4.x += f[j][0]; rs4.y += f[j][1]; rs4.z += f[j][2]; rs4.w += f[j][3]; }
            }
            x3_store<AKC>(lds, f, ts);
        } else {
            x3_store<BKC>(lds + X3_OP, f, ts);
        }
    };
    // One LDS stage and two barriers per k-tile, THREE workgroups per CU: while one splits and writes its next tile
    // (vector ALU, LDS writes) the others run their MFMAs -- the co-execution the bf16 MFMA allows, obtained from the
    // hardware's choice between three waves per SIMD rather than from an instruction order the compiler would have
    // to keep (measured: two stages and one workgroup per CU 88-140 TF-equivalent, one stage and two 142-167, three
    // 147-180).  The global loads of tile kt + 1 are issued behind the split of tile kt and have the whole MFMA phase
    // to land.
    if (nk > 0) load(0);
    // unit of this lane inside a (tile, k step) block of 64 (x3_pos: the same in every tile)
    const int fl = (lane >> 5) * 32 + x3_pos(0, lane & 31);
    const x3_u32x4* sa = lds;
    const x3_u32x4* sb = lds + X3_OP;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();      // every wave has finished reading tile kt - 1
        store();              // tile kt: registers -> three bf16 pieces -> LDS
        if (kt + 1 < nk) load(kt + 1);
        __syncthreads();      // tile kt is in LDS
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x3_bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = __builtin_bit_cast(x3_bf16x8, sa[p * X3_PLANE + ((wm * 2 + i) * 2 + ks) * 64 + fl]);
                    b[i][p] = __builtin_bit_cast(x3_bf16x8, sb[p * X3_PLANE + ((wn * 2 + i) * 2 + ks) * 64 + fl]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc.t[i][j];                // small products first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc.t[i][j] = c;
                }
        }
    }
    __syncthreads();
    float* scratch = reinterpret_cast<float*>(x3_smem);
    if (!AKC && rs_dst != nullptr) {
        // four staging threads (k octets o = 0..3) hold partial sums of the same four rows: summed in o order
        if (stage_a) *reinterpret_cast<float4*>(scratch + (ts >> 5) * 128 + 4 * (ts & 31)) = rs4;
        __syncthreads();
        if (t < 128) rs_dst[t] = ((scratch[t] + scratch[128 + t]) + scratch[256 + t]) + scratch[384 + t];
        __syncthreads();
    }
    gemm_epilogue_wide<0>(acc, ep, scratch, m0, n0, M, N, ep_splitk, ep_slabs);
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 3) void gemm_x3_kernel(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ B, int64_t ldb, GemmEpilogue ep,
                                                         int M, int N, int K, int tiles_n, int splitk,
                                                         float* __restrict__ slabs) {
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = ((K + splitk - 1) / splitk + XK - 1) / XK * XK;
        k_begin = blockIdx.z * per;
        k_end = min(K, k_begin + per);
    }
    float* rs_dst = (ep.rowsum != nullptr && tn == 0) ? ep.rowsum + (int64_t)blockIdx.z * M + m0 : nullptr;
    x3_tile<AKC, BKC>(A, lda, B, ldb, ep, M, N, m0, n0, k_begin, (k_end - k_begin) / XK, rs_dst, splitk, slabs);
}

// The grouped form (gemm.hip gemm_dma_pf_grouped_kernel: `groups` problems of one shape in one flat grid,
// workgroup -> (group, reduction split, tile); slab index = g * splitk + z).
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 3) void gemm_x3_grouped_kernel(GemmGroupPtrs gp, int64_t lda, int64_t ldb,
                                                                 GemmEpilogue ep, int M, int N, int K, int tiles_n,
                                                                 int tiles, int splitk, float* __restrict__ slabs,
                                                                 float* __restrict__ rs_part) {
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int per_g = tiles * splitk;
    const int g = lin / per_g;
    const int r = lin - g * per_g;
    const int z = r / tiles;
    const int tile = r - z * tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int per = K / splitk;                       // host: whole 32-deep tiles per split
    ep.C = gp.C[g];
    ep.bias = gp.bias[g];
    ep.residual = gp.residual[g];
    ep.preact = gp.preact[g];
    ep.gradz = gp.gradz[g];
    const int slab = g * splitk + z;
    float* my_slab = slabs ? slabs + (int64_t)slab * M * N : nullptr;
    float* rs_dst = (rs_part != nullptr && tn == 0) ? rs_part + (int64_t)slab * M + m0 : nullptr;
    // (1-D grid: blockIdx.z == 0, so the epilogue's slab offset is the one folded into my_slab)
    x3_tile<AKC, BKC>(gp.A[g], lda, gp.B[g], ldb, ep, M, N, m0, n0, z * per, per / XK, rs_dst, my_slab ? 2 : 1, my_slab);
}


// ---- the 64 x 64-tile form: products whose 128 x 128 tiling leaves most of the chip idle (a 2,048-row shard's
// 512 -> 512 Linear products: 64 tiles of 128, 256 of 64).  Same pieces, units, swizzle and MFMA order; a wave owns one
// 32 x 32 accumulator; an operand tile is 64 rows x 32 k = 16 values per staging thread:
//   KC ([X][K]): thread (row = t & 63, h = t >> 6): 4 16-B loads, k = 16 h .. 16 h + 15 -> the octets 2h, 2h + 1
//   XC ([K][X]): thread (q = t & 31, o = t >> 5): 8 8-B loads, rows 2q, 2q + 1 at k = 8 o + j -> octet o of both rows
constexpr int X3H_PLANE = 2 * 2 * 2 * 32;     // units of one piece of one 64-row operand tile
constexpr int X3H_OP = 3 * X3H_PLANE;         // 12 KB
constexpr int X3H_LDS = 2 * X3H_OP * 16;      // 24 KB: A then B

template <bool KC>
__device__ __forceinline__ void x3h_load(f32x4 (&f)[4], const float* __restrict__ P, int64_t ld, int x0, int k0, int t) {
    if (KC) {
        const f32x4* p = reinterpret_cast<const f32x4*>(P + (int64_t)(x0 + (t & 63)) * ld + k0 + 16 * (t >> 6));
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = p[j];
    } else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const float* p = P + (int64_t)(k0 + 8 * (t >> 5)) * ld + x0 + 2 * (t & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) {                     // f[j] = (rows 2q, 2q+1 at k 2j), (the same at k 2j + 1)
            const f32x2 a = *reinterpret_cast<const f32x2*>(p + (int64_t)(2 * j) * ld);
            const f32x2 b = *reinterpret_cast<const f32x2*>(p + (int64_t)(2 * j + 1) * ld);
            f[j] = f32x4{a[0], a[1], b[0], b[1]};
        }
    }
}
template <bool KC>
__device__ __forceinline__ void x3h_store(x3_u32x4* op, const f32x4 (&f)[4], int t) {
    if (KC) {
        const int row = t & 63, h = t >> 6, tile = row >> 5, r = row & 31;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float v[8] = {f[2 * u][0], f[2 * u][1], f[2 * u][2], f[2 * u][3],
                                f[2 * u + 1][0], f[2 * u + 1][1], f[2 * u + 1][2], f[2 * u + 1][3]};
            x3_put<X3H_PLANE>(op, (tile * 4 + 2 * h + u) * 32 + x3_pos(tile, r), v);
        }
    } else {
        const int q = t & 31, o = t >> 5, tile = q >> 4, r0 = (2 * q) & 31;
        {
            const float v[8] = {f[0][0], f[0][2], f[1][0], f[1][2], f[2][0], f[2][2], f[3][0], f[3][2]};
            x3_put<X3H_PLANE>(op, (tile * 4 + o) * 32 + x3_pos(tile, r0), v);
        }
        {
            const float v[8] = {f[0][1], f[0][3], f[1][1], f[1][3], f[2][1], f[2][3], f[3][1], f[3][3]};
            x3_put<X3H_PLANE>(op, (tile * 4 + o) * 32 + x3_pos(tile, r0 + 1), v);
        }
    }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 4) void gemm_x3_half_kernel(const float* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ B, int64_t ldb,
                                                              GemmEpilogue ep, int M, int N, int K, int tiles_n,
                                                              int splitk, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char x3_smem[];
    x3_u32x4* lds = reinterpret_cast<x3_u32x4*>(x3_smem);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;
    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = ((K + splitk - 1) / splitk + XK - 1) / XK * XK;
        k_begin = blockIdx.z * per;
        k_end = min(K, k_begin + per);
    }
    const int nk = (k_end - k_begin) / XK;
    const int wm = wave >> 1, wn = wave & 1;
    const bool stage_a = wave < 2;
    const int ts = t & 127;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const bool want_rs = !AKC && ep.rowsum != nullptr && tn == 0;
    const bool do_rs = want_rs && stage_a;
    float rs0 = 0.0f, rs1 = 0.0f;           // rows 2q, 2q + 1 over this thread's k rows

    f32x4 f[4];
    auto load = [&](int kt) {
        const int k0 = k_begin + kt * XK;
        if (stage_a) x3h_load<AKC>(f, A, lda, m0, k0, ts);
        else x3h_load<BKC>(f, B, ldb, n0, k0, ts);
    };
    auto store = [&]() {
        if (stage_a) {
            if (do_rs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { rs0 += f[j][0] + f[j][2]; rs1 += f[j][1] + f[j][3]; }
            }
            x3h_store<AKC>(lds, f, ts);
        } else {
            x3h_store<BKC>(lds + X3H_OP, f, ts);
        }
    };
    if (nk > 0) load(0);
    const int fl = (lane >> 5) * 32 + x3_pos(0, lane & 31);
    const x3_u32x4* sa = lds;
    const x3_u32x4* sb = lds + X3H_OP;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        store();
        if (kt + 1 < nk) load(kt + 1);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x3_bf16x8 a[3], b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = __builtin_bit_cast(x3_bf16x8, sa[p * X3H_PLANE + (wm * 2 + ks) * 64 + fl]);
                b[p] = __builtin_bit_cast(x3_bf16x8, sb[p * X3H_PLANE + (wn * 2 + ks) * 64 + fl]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);      // small products first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    __syncthreads();
    float* scratch = reinterpret_cast<float*>(x3_smem);
    if (want_rs) {
        // four staging threads (k octets o = 0..3) hold partial sums of the same two rows: summed in o order
        if (stage_a) { scratch[(ts >> 5) * 64 + 2 * (ts & 31)] = rs0; scratch[(ts >> 5) * 64 + 2 * (ts & 31) + 1] = rs1; }
        __syncthreads();
        if (t < 64)
            ep.rowsum[(int64_t)blockIdx.z * M + m0 + t] = ((scratch[t] + scratch[64 + t]) + scratch[128 + t]) + scratch[192 + t];
        __syncthreads();
    }
    Acc one;
    one.t[0][0] = acc;
    gemm_epilogue_wave<1, Acc, 0>(one, ep, scratch + wave * (32 * 32), m0 + wm * 32, n0 + wn * 32, M, N, splitk, slabs, 0, 1);
}

}  // namespace qarig

using namespace qarig;

// the 64 x 64-tile form: whole 64-tiles and 32-deep k-tiles per split, fewer than 192 tiles of 128 x 128
int qarig_gemm_x3_half_ok(int M, int N, int K, int splitk) {
    if (M < 64 || N < 64 || K < XK || M % 64 || N % 64 || K % XK || splitk > K || splitk < 1) return 0;
    if (splitk > 1) {
        const long per = (((long)K + splitk - 1) / splitk + XK - 1) / XK * XK;
        if (K % per) return 0;
    }
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    return t128 < 192;
}

int qarig_gemm_x3_half_launch(const float* A, int64_t lda, int a_kcontig, const float* B, int64_t ldb, int b_kcontig,
                              const GemmEpilogue& ep, int M, int N, int K, int splitk, float* slabs, hipStream_t st) {
    const int tiles_n = N / 64;
    const dim3 grid((M / 64) * tiles_n, 1, splitk), block(256);
#define QARIG_X3H(AK, BK_)                                                                                         \
    hipLaunchKernelGGL((gemm_x3_half_kernel<AK, BK_>), grid, block, X3H_LDS, st, A, lda, B, ldb, ep, M, N, K, tiles_n, splitk, slabs)
    if (a_kcontig && b_kcontig) QARIG_X3H(true, true);
    else if (a_kcontig) QARIG_X3H(true, false);
    else if (b_kcontig) QARIG_X3H(false, true);
    else QARIG_X3H(false, false);
#undef QARIG_X3H
    return QARIG_OK;
}

// shapes the kernel takes: whole 128 x 128 tiles, whole 32-deep k-tiles per split, 16-B aligned operands and rows
extern "C" int qarig_gemm_x3_ok(int M, int N, int K, int splitk) {
    if (M < 128 || N < 128 || K < XK || M % BM || N % BN || K % XK || splitk > K) return 0;
    if (splitk > 1) {
        const long per = (((long)K + splitk - 1) / splitk + XK - 1) / XK * XK;
        if (K % per) return 0;
    }
    return 1;
}

int qarig_gemm_x3_grouped_launch(const GemmGroupPtrs& gp, int64_t lda, int a_kcontig, int64_t ldb, int b_kcontig,
                                 const GemmEpilogue& ep, int M, int N, int K, int tiles_n, int tiles, int splitk,
                                 float* slabs, float* rs_part, unsigned total_wg, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_x3_grouped_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_x3_grouped_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_x3_grouped_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        attr_set = true;
    }
    const dim3 grid(total_wg), block(256);
#define QARIG_X3G(AK, BK_)                                                                                         \
    hipLaunchKernelGGL((gemm_x3_grouped_kernel<AK, BK_>), grid, block, X3_LDS, st, gp, lda, ldb, ep, M, N, K, tiles_n, \
                       tiles, splitk, slabs, rs_part)
    if (a_kcontig && b_kcontig) QARIG_X3G(true, true);
    else if (a_kcontig) QARIG_X3G(true, false);
    else QARIG_X3G(false, false);                  // (the host refuses the (xc, kc) layout for grouped launches)
#undef QARIG_X3G
    return QARIG_OK;
}

int qarig_gemm_x3_launch(const float* A, int64_t lda, int a_kcontig, const float* B, int64_t ldb, int b_kcontig,
                         const GemmEpilogue& ep, int M, int N, int K, int splitk, float* slabs, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_x3_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_x3_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_x3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_x3_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        attr_set = true;
    }
    const int tiles_n = N / BN;
    const dim3 grid((M / BM) * tiles_n, 1, splitk), block(256);
#define QARIG_X3(AK, BK_)                                                                                          \
    hipLaunchKernelGGL((gemm_x3_kernel<AK, BK_>), grid, block, X3_LDS, st, A, lda, B, ldb, ep, M, N, K, tiles_n, splitk, slabs)
    if (a_kcontig && b_kcontig) QARIG_X3(true, true);
    else if (a_kcontig) QARIG_X3(true, false);
    else if (b_kcontig) QARIG_X3(false, true);
    else QARIG_X3(false, false);
#undef QARIG_X3
    return QARIG_OK;
}
