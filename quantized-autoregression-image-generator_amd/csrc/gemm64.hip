// fp32 MFMA GEMM on 64 x 64 tiles: the Linear products whose 128 x 128 tiling leaves most of the chip idle.
// A per-GPU shard of 8 sequences has 2,048 rows: the 512 -> 512 ResidualLinearLayer products (reference
// models/layers.py:291-304; 93 per step forward, as many input and weight gradients) are 16 x 4 = 64 tiles of
// 128 x 128 -- a quarter of the 256 CUs, which the two-team kernel (gemm_dma_pf2_kernel) + a split-K reduce
// launch ran at 35-40 TF -- and 256 tiles of 64 x 64: every CU, no split, the epilogue in the launch.  The same
// holds for the N x 255-row window evaluations of sliding-window generation (generate_images.py:275-286).
//
// 256 threads = 4 waves in a 2 x 2 arrangement, each wave one 32 x 32 accumulator of v_mfma_f32_32x32x2_f32
// (16 registers: several workgroups per CU cover each other's load latency); operand tiles of 16 k go
// global -> registers -> k-major LDS tiles T[k][x] (row stride 68 floats: conflict-free fragment reads),
// double-buffered, one barrier per k-tile.  Either operand reduction-contiguous ([X][K]) or tile-contiguous
// ([K][X]).  Same fp32 fma chain per output as the 128-tile kernels up to the k-order inside a tile.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qarig_common.h"
#include "gemm_epilogue.h"

namespace qarig {

constexpr int T64 = 64;            // tile edge
constexpr int LD64 = T64 + 4;      // LDS row stride in floats
constexpr int ST64 = BK * LD64;    // floats per staged operand tile

// one thread's float4 of an operand tile: reduction-contiguous source [X][K] -> (row t >> 2, k 4 (t & 3));
// tile-contiguous source [K][X] -> (k t >> 4, x 4 (t & 15))
template <bool KC>
__device__ __forceinline__ float4 t64_load(const float* __restrict__ p, int64_t ld, int x0, int k0, int t) {
    if (KC) return *reinterpret_cast<const float4*>(p + (int64_t)(x0 + (t >> 2)) * ld + k0 + 4 * (t & 3));
    return *reinterpret_cast<const float4*>(p + (int64_t)(k0 + (t >> 4)) * ld + x0 + 4 * (t & 15));
}
template <bool KC>
__device__ __forceinline__ void t64_store(float* __restrict__ T, float4 v, int t) {
    if (KC) {
        float* q = T + (4 * (t & 3)) * LD64 + (t >> 2);
        q[0] = v.x; q[LD64] = v.y; q[2 * LD64] = v.z; q[3 * LD64] = v.w;
    } else {
        *reinterpret_cast<float4*>(T + (t >> 4) * LD64 + 4 * (t & 15)) = v;
    }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm64_kernel(const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb, GemmEpilogue ep,
                                                        int M, int N, int K, int tiles_n, int splitk,
                                                        float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][ST64];     // [stage][A | B]
    __shared__ float rsum[16][T64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * T64, n0 = tn * T64;
    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
        k_begin = blockIdx.z * per;
        k_end = min(K, k_begin + per);
    }
    const int nk = (k_end - k_begin) / BK;
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 31, fh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // sum_k A(m, k) of this split (the bias gradient riding on a weight-gradient product): column tiles 0 only,
    // tile-contiguous A: a thread keeps the sums of its four m over its k rows
    const bool do_rs = !AKC && ep.rowsum != nullptr && tn == 0;
    float4 rs4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // With one workgroup per CU nothing else covers a load's latency (1-2 us from HBM against 0.2 us of MFMAs per
    // k-tile): the operand tiles of the next PD k-tiles are in flight in registers (a ring indexed by the unrolled
    // loop position), each written to its LDS stage one barrier before it is read.
    constexpr int PD = 8;
    float4 ra[PD], rb[PD];
#pragma unroll
    for (int u = 0; u < PD; ++u) {
        const int kk = k_begin + min(u, nk - 1) * BK;      // (tiles past the end re-read the last one; never stored)
        ra[u] = t64_load<AKC>(A, lda, m0, kk, t);
        rb[u] = t64_load<BKC>(B, ldb, n0, kk, t);
    }
    for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int kt = kt0 + u;
            if (kt >= nk) break;                             // workgroup-uniform
            float* TA = lds[kt & 1][0];
            float* TB = lds[kt & 1][1];
            t64_store<AKC>(TA, ra[u], t);
            t64_store<BKC>(TB, rb[u], t);
            if (do_rs) { rs4.x += ra[u].x; rs4.y += ra[u].y; rs4.z += ra[u].z; rs4.w += ra[u].w; }
            __syncthreads();      // tile kt visible; every wave has finished reading the other stage (tile kt - 1)
            if (kt + PD < nk) {
                ra[u] = t64_load<AKC>(A, lda, m0, k_begin + (kt + PD) * BK, t);
                rb[u] = t64_load<BKC>(B, ldb, n0, k_begin + (kt + PD) * BK, t);
            }
            const float* fa = TA + fh * LD64 + wm * 32 + fi;
            const float* fb = TB + fh * LD64 + wn * 32 + fi;
#pragma unroll
            for (int s2 = 0; s2 < BK / 2; ++s2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * s2 * LD64], fb[2 * s2 * LD64], acc, 0, 0, 0);
        }
    }
    if (do_rs) {
        // 16 threads (k rows t >> 4) hold partial sums of the same four m: summed in k-row order
        *reinterpret_cast<float4*>(&rsum[t >> 4][4 * (t & 15)]) = rs4;
        __syncthreads();
        if (t < T64) {
            float s = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += rsum[q][t];
            ep.rowsum[(int64_t)blockIdx.z * M + m0 + t] = s;
        }
    }
    const int col = n0 + wn * 32 + fi;
    if (splitk > 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + acc_row(r, lane);
            slabs[((int64_t)blockIdx.z * M + row) * N + col] = acc[r];
        }
        return;
    }
    const float b = ep.bias ? ep.bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + acc_row(r, lane);
        float v = acc[r] + b;
        if (ep.residual) v += ep.residual[(int64_t)row * ep.ldr + col];
        if (ep.preact) ep.preact[(int64_t)row * ep.ldp + col] = v;
        float y = act_fwd(v, ep.act);
        if (ep.gradz) y *= act_grad(ep.gradz[(int64_t)row * ep.ldz + col], ep.gact);
        ep.C[(int64_t)row * ep.ldc + col] = y;
    }
}

}  // namespace qarig

using namespace qarig;

// Shapes the 64-tile kernel takes (the dispatcher of gemm.hip asks; so does the host when it picks a reduction
// split): whole 64 x 64 tiles and 16-deep k-tiles of 16-B aligned operands, fewer than 192 tiles of 128 x 128,
// and a reduction of at most 1,024 unless the tiles are very few.  Measured (tools/gemm_bench.py, ROWS = 2048 /
// 1024, profiles/r04_gemm64_*.log): 2048 x 512 x 512 26.7 -> 18.0 us, its 512 x 512 weight gradient over 2,048
// rows 28.9 -> 20.7, 1024 x 512 x 512 23.3 -> 14.6, 1024 x 2048 x 512 34.0 -> 29.0; with K = 2048 on 32-64 tiles
// of 128 the split-K ring kernels are as fast or faster (2048 x 512 x 2048: 50.6 against 50.0-60 us): this
// register-staged loop reaches ~70 TF where the LDS-DMA ring reaches ~90 on a quarter of the chip plus a reduce.
extern "C" int qarig_gemm_tile64(int M, int N, int K) {
    if (g_qarig_opt.gemm_tile64 == 0 || M < 64 || N < 64 || K < 16 || M % 64 || N % 64 || K % 16) return 0;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    return g_qarig_opt.gemm_tile64 == 1 || (t128 < 192 && (K <= 1024 || t128 <= 16));
}

int qarig_gemm64_launch(const float* A, int64_t lda, int a_kcontig, const float* B, int64_t ldb, int b_kcontig,
                        const GemmEpilogue& ep, int M, int N, int K, int splitk, float* slabs, hipStream_t st) {
    const int tiles_n = N / T64;
    const dim3 grid((M / T64) * tiles_n, 1, splitk), block(256);
#define QARIG_G64(AK, BK_)                                                                                       \
    hipLaunchKernelGGL((gemm64_kernel<AK, BK_>), grid, block, 0, st, A, lda, B, ldb, ep, M, N, K, tiles_n, splitk, slabs)
    if (a_kcontig && b_kcontig) QARIG_G64(true, true);
    else if (a_kcontig) QARIG_G64(true, false);
    else if (b_kcontig) QARIG_G64(false, true);
    else QARIG_G64(false, false);
#undef QARIG_G64
    return QARIG_OK;
}
