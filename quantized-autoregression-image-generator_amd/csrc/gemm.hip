// fp32 MFMA GEMM with fused epilogues: the contraction behind every Linear layer
// of the Transformer (reference models/layers.py:234-254 LinearLayer, :258-304
// ResidualLinearLayer, :308-366 FeedforwardBlock, :389-418 q/k/v blocks) and its
// backward (dX = dY W, dW = dY^T X).
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )
//
// A and B may each be stored reduction-contiguous ([X][K], "kc") or
// tile-contiguous ([K][X], "xc"); that covers forward (kc,kc), backward-data
// (kc,xc) and backward-weight (xc,xc) without materialising a transpose.
#include <stdlib.h>

#include "qarig_common.h"
#include "gemm_epilogue.h"
#include "ring_common.h"

namespace qarig {

// sum_k A(m0 + tid, k) from the staged A tiles (bias gradient riding on the dW GEMM).
struct RowSumHook {
    float rs;
    bool on;
    int tid;
    __device__ __forceinline__ void operator()(const float* ta, const float*) {
        if (on) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) rs += ta[kk * LDT + tid];
        }
    }
};

// FAST: every tile interior (M,N multiples of 128, every K split a multiple of 16,
// vector-loadable operands) -- branch-free main loop, unguarded epilogue.
template <class SA, class SB, bool FAST>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(SA sa, SB sb, GemmEpilogue ep, int M,
                                                           int N, int K, int tiles_n, int splitk,
                                                           float* slabs, int vec_epi) {
    __shared__ __attribute__((aligned(16))) float lds[GEMM_LDS_FLOATS];
    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
        k_begin = blockIdx.z * per;
        k_end = min(K, k_begin + per);
    }

    Acc acc;
    acc_zero(acc);
    RowSumHook hook{0.0f, ep.rowsum != nullptr && tn == 0 && threadIdx.x < 128, (int)threadIdx.x};
    contract_loop<FAST>(acc, sa, sb, m0, n0, k_begin, k_end, lds, hook);
    if (hook.on && m0 + (int)threadIdx.x < M)
        ep.rowsum[(int64_t)blockIdx.z * M + m0 + threadIdx.x] = hook.rs;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int cl = lane & 31;

    if (FAST && vec_epi) {
        gemm_epilogue_wide<2>(acc, ep, lds, m0, n0, M, N, splitk, slabs);
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + cl;
            if (!FAST && col >= N) continue;
            const float b = (ep.bias && splitk == 1) ? ep.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (!FAST && row >= M) continue;
                float t = acc.t[i][j][r];
                if (splitk > 1) {
                    slabs[((int64_t)blockIdx.z * M + row) * N + col] = t;
                    continue;
                }
                t += b;
                if (ep.residual) t += ep.residual[(int64_t)row * ep.ldr + col];
                if (ep.preact) ep.preact[(int64_t)row * ep.ldp + col] = t;
                float y = act_fwd(t, ep.act);
                if (ep.gradz) y *= act_grad(ep.gradz[(int64_t)row * ep.ldz + col], ep.gact);
                ep.C[(int64_t)row * ep.ldc + col] = y;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Interior GEMM, LDS-DMA staging.  Ablations (profiles/README.md) show the register-staged
// loop above loses ~8 % to its ds_write traffic and ~7 % to ds_read2_b32 fragment reads: LDS
// instruction issue competes with MFMA issue.  Here the tiles never touch VGPRs on the way
// in: global_load_lds_dwordx4 writes them straight into a 2-stage LDS ring (32 KB: 4 blocks
// per CU; ONE barrier per k-tile both publishes tile t and retires the reads of tile t-1,
// whose stage the DMA of tile t+1 then refills under tile t's MFMAs), and reduction-contiguous
// operands are kept k-contiguous in LDS ([x][16] floats, 16-B chunks XOR-swizzled by
// (x>>2)&3 on the SOURCE address so the lane-linear DMA image needs no padding) and read
// back as two conflict-free ds_read_b128 per fragment.  A lane-half h then owns k = 8h..8h+7
// of a tile, so MFMA step s contracts k = {s, 8+s}: a fixed permutation of the summation
// order, nothing else.  Fragment reads are inline asm: hipcc otherwise drains every
// outstanding DMA (vmcnt(0)) in front of any LDS read.
// ---------------------------------------------------------------------------------
// The ring (4 stages) with the fragments of tile t+1 read from LDS into a second register
// set WHILE tile t's 32 MFMAs issue: the LDS round trip (8 x ds_read_b128 + latency behind the
// other waves' LDS traffic) leaves the k-loop's critical path; what remains between two MFMA
// slabs is the barrier and the issue of 4 DMAs and 8 reads.  (Counters on the 2-stage kernel:
// waves spend 19 % of their cycles in s_waitcnt / s_barrier and the matrix pipe idles 24 %.)
// The k-loop of the prefetch kernels: tiles [0, nk) of BK from k_begin, of the 128 x 128 block at
// (m0, n0), through the 4-stage ring at `lds` (64 KB); `tt` = thread index among the ring's 256.
// Contains workgroup barriers: every wave of the workgroup runs it with the same nk.
template <bool AKC, bool BKC>
__device__ __forceinline__ void pf_ring(Acc& acc, float& rs, const bool do_rs, float* lds,
                                        const float* __restrict__ A, int64_t lda,
                                        const float* __restrict__ B, int64_t ldb, int m0, int n0,
                                        int k_begin, int nk, int wave, int lane, int tt) {
    const int wm = wave >> 1, wn = wave & 1;
    const int x = lane & 31, h = lane >> 5;
    const int last = k_begin + (nk - 1) * BK;
    // uniform tile origins; the k advance is k (KC) or k * ld (XC) elements
    const float* a_org = AKC ? A + (int64_t)m0 * lda : A + m0;
    const float* b_org = BKC ? B + (int64_t)n0 * ldb : B + n0;
    const unsigned oa0 = dma_lane_off<AKC>(lda, wave * 2, lane), oa1 = dma_lane_off<AKC>(lda, wave * 2 + 1, lane);
    const unsigned ob0 = dma_lane_off<BKC>(ldb, wave * 2, lane), ob1 = dma_lane_off<BKC>(ldb, wave * 2 + 1, lane);
    const unsigned my_dma_addr = __builtin_amdgcn_readfirstlane(lds_addr(lds + wave * 512));   // this wave's two 1-KB slots
    auto issue = [&](int t, int stage) {             // DMA of tile t (clamped: a harmless re-load past the end)
        const int k = min(k_begin + t * BK, last);
        const char* ak = reinterpret_cast<const char*>(AKC ? a_org + k : a_org + (int64_t)k * lda);
        const char* bk = reinterpret_cast<const char*>(BKC ? b_org + k : b_org + (int64_t)k * ldb);
        const unsigned dst = my_dma_addr + stage * (DMA_STAGE_FLOATS * 4);
        dma16_saddr(ak, oa0, dst);
        dma16_saddr(ak, oa1, dst + 1024);
        dma16_saddr(bk, ob0, dst + DMA_OP_FLOATS * 4);
        dma16_saddr(bk, ob1, dst + DMA_OP_FLOATS * 4 + 1024);
    };
    // sum_k A(m0 + tid, k) of the tile in stage S, k ascending (A is [k][x] here)
    const unsigned rs_base = lds_addr(lds + tt);
#define QARIG_PF_ROWSUM(S)                                                                         \
    {                                                                                              \
        f32x4 r0, r1, r2, r3;                                                                      \
        rd_xc<(S) * DMA_STAGE_FLOATS * 4>(rs_base, r0, r1);                                        \
        rd_xc<(S) * DMA_STAGE_FLOATS * 4 + 8 * 128 * 4>(rs_base, r2, r3);                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) rs += r0[q];                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) rs += r1[q];                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) rs += r2[q];                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) rs += r3[q];                                 \
    }
    if (nk > 0) {
        FragBase fb;
        frag_bases<AKC>(lds, wm, x, h, fb.a0, fb.a1);
        frag_bases<BKC>(lds, wn, x, h, fb.b0, fb.b1);
        issue(0, 0);
        issue(1, 1);
        issue(2, 2);
        issue(3, 3);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // tile 0 landed (1, 2, 3 in flight)
        __builtin_amdgcn_s_barrier();
        Frags8 P, Q;
        frags_read_s<AKC, BKC, 0>(P, fb);
        frags_wait(P);
        __builtin_amdgcn_sched_barrier(0);
        if (do_rs) QARIG_PF_ROWSUM(0)
        int t = 0;
        // body for tile t (stage S = t % 4) with its fragments in CUR; leaves tile t+1's in NXT
#define QARIG_PF_BODY(CUR, NXT, S)                                                                \
        {                                                                                         \
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   /* tile t+1 landed; t+2, t+3 in flight */ \
            __builtin_amdgcn_s_barrier();      /* ... for everyone; all reads of tile t retired */  \
            issue(t + 4, S);                   /* into the stage tile t has just vacated */        \
            frags_read_s<AKC, BKC, (S + 1) % 4>(NXT, fb);                                         \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            frags_mma(acc, CUR);                                                                  \
            frags_wait(NXT);                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            if (do_rs && t + 1 < nk) QARIG_PF_ROWSUM((S + 1) % 4)                                 \
            ++t;                                                                                  \
        }
        while (t + 4 <= nk) {
            QARIG_PF_BODY(P, Q, 0)
            QARIG_PF_BODY(Q, P, 1)
            QARIG_PF_BODY(P, Q, 2)
            QARIG_PF_BODY(Q, P, 3)
        }
        if (t < nk) QARIG_PF_BODY(P, Q, 0)
        if (t < nk) QARIG_PF_BODY(Q, P, 1)
        if (t < nk) QARIG_PF_BODY(P, Q, 2)
#undef QARIG_PF_BODY
#undef QARIG_PF_ROWSUM
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_dma_pf_kernel(const float* __restrict__ A, int64_t lda,
                                                                  const float* __restrict__ B, int64_t ldb,
                                                                  GemmEpilogue ep, int M, int N, int K,
                                                                  int tiles_n, int splitk, float* slabs,
                                                                  int xcd_splits) {
    __shared__ __attribute__((aligned(16))) float lds[PF_STAGES * DMA_STAGE_FLOATS];   // 64 KB: two workgroups per CU
    int tile, z;
    if (xcd_splits) {
        // Split reductions over few tiles (weight gradients: 64 tiles x 8 splits): the grid is flat and a
        // reduction split belongs to ONE XCD -- workgroup b runs on XCD b % 8 (round-robin dispatch), takes
        // split (b % 8) * (splitk / 8) + ... and walks every tile of it -- so that an XCD's L2 streams ONE
        // k-range of both operands instead of every k-range of an eighth of the tiles (fabric reads of the
        // 2048 x 512 gradient over 16,384 rows: 474 MB -> the algorithmic 168 MB).  Placement only changes
        // which L2 serves a line, never a result.
        const int tiles = gridDim.x / splitk;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;        // j-th workgroup of this XCD
        const int per_xcd = splitk >> 3;                             // host: splitk % 8 == 0
        z = xcd * per_xcd + j / tiles;
        tile = j - (j / tiles) * tiles;
    } else {
        tile = xcd_remap(blockIdx.x, gridDim.x);
        z = blockIdx.z;
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
        k_begin = z * per;
        k_end = min(K, k_begin + per);
    }
    const int nk = (k_end - k_begin) / BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    Acc acc;
    acc_zero(acc);
    float rs = 0.0f;
    const bool do_rs = !AKC && ep.rowsum != nullptr && tn == 0 && tid < 128;
    pf_ring<AKC, BKC>(acc, rs, do_rs, lds, A, lda, B, ldb, m0, n0, k_begin, nk, wave, lane, tid);
    if (do_rs) ep.rowsum[(int64_t)z * M + m0 + tid] = rs;
    __syncthreads();                      // ring no longer in use: the epilogue stages through it
    // (flat grid: blockIdx.z == 0, the slab offset is folded into the pointer)
    gemm_epilogue_wide<2>(acc, ep, lds, m0, n0, M, N, splitk, xcd_splits && slabs ? slabs + (int64_t)z * M * N : slabs);
}

// ---------------------------------------------------------------------------------
// Paired form for launches of at most one workgroup per CU (M = a few thousand rows: the
// per-GPU shard of a small global batch, autoregressive decode with many beams).  With 256
// 128x128 tiles or fewer the kernels above leave one wave per SIMD, and a lone wave exposes
// every barrier and DMA wait.  Here a workgroup is 8 waves = two 4-wave teams on the SAME tile:
// team 0 reduces the first half of the block's k-range, team 1 the second, each through its own
// 4-stage ring (128 KB of LDS), so every SIMD holds two waves at different points of the k-loop
// without a second slab in HBM.  At the end the teams swap halves of their accumulators through
// LDS: wave (team h, tile position w) keeps the 32-row half h of its 64x64 tile, adds its
// partner's partial (low-k + high-k, a fixed order) and runs the epilogue on those 32 rows.
template <bool AKC, bool BKC>
__global__ __launch_bounds__(2 * NTHREADS, 1) void gemm_dma_pf2_kernel(const float* __restrict__ A, int64_t lda,
                                                                       const float* __restrict__ B, int64_t ldb,
                                                                       GemmEpilogue ep, int M, int N, int K,
                                                                       int tiles_n, int splitk, float* slabs) {
    constexpr int RING = PF_STAGES * DMA_STAGE_FLOATS;               // 64 KB per team
    extern __shared__ __attribute__((aligned(16))) float lds2[];    // 2 rings = 128 KB
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    int k_begin = 0, k_end = K;
    if (splitk > 1) {
        const int per = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
        k_begin = blockIdx.z * per;
        k_end = min(K, k_begin + per);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave8 >> 2, wave = wave8 & 3;
    const int tt = tid & 255;                                        // thread index inside the team
    const int nk = (k_end - k_begin) / BK / 2;                       // k-tiles per team (host: even split)
    k_begin += team * nk * BK;
    const int wm = wave >> 1, wn = wave & 1;

    Acc acc;
    acc_zero(acc);
    float rs = 0.0f;
    const bool do_rs = !AKC && ep.rowsum != nullptr && tn == 0 && tt < 128;
    // both teams run the same number of k-tiles, so the workgroup-wide barriers inside pair up
    pf_ring<AKC, BKC>(acc, rs, do_rs, lds2 + team * RING, A, lda, B, ldb, m0, n0, k_begin, nk, wave, lane, tt);
    __syncthreads();                      // both rings idle: exchange + epilogue staging reuse them
    // ---- swap accumulator halves: this wave sends its partial of the 32-row half it does NOT keep
    float* xch = lds2;                    // [8 waves][8 x f32x4][64 lanes]: 64 KB
    {
        f32x4* mine = reinterpret_cast<f32x4*>(xch) + (size_t)wave8 * 8 * 64 + lane;
        auto send = [&](const f32x16& s, int j) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                mine[(j * 4 + q) * 64] = f32x4{s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]};
        };
        if (team == 0) { send(acc.t[1][0], 0); send(acc.t[1][1], 1); }      // wave-uniform branch
        else { send(acc.t[0][0], 0); send(acc.t[0][1], 1); }
    }
    float* rsx = lds2 + 8 * 8 * 64 * 4;   // 128 floats behind the exchange buffer
    if (do_rs && team == 1) rsx[tt] = rs;
    __syncthreads();
    {
        const f32x4* theirs = reinterpret_cast<const f32x4*>(xch) + (size_t)(wave8 ^ 4) * 8 * 64 + lane;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = theirs[(j * 4 + q) * 64];
                // low-k partial + high-k partial
                if (team == 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc.t[0][j][4 * q + u] = acc.t[0][j][4 * q + u] + v[u];
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc.t[1][j][4 * q + u] = v[u] + acc.t[1][j][4 * q + u];
                }
            }
    }
    if (do_rs && team == 0) ep.rowsum[(int64_t)blockIdx.z * M + m0 + tt] = rs + rsx[tt];
    __syncthreads();                      // exchange buffer read: the epilogue stages through it
    gemm_epilogue_wave<2, Acc, 2>(acc, ep, lds2 + wave8 * (32 * 64), m0 + wm * 64, n0 + wn * 64, M, N, splitk, slabs,
                                  team, team + 1);
}

// ---------------------------------------------------------------------------------
// Grouped form: `groups` problems of ONE shape (M, N, K, layouts, strides, epilogue kind) with
// their own operand / output pointers, as one launch.  The per-GPU shard of a small global batch
// (2,048 rows) leaves every Linear product of the q/k/v MLPs (reference models/layers.py:389-418)
// at 64 or 256 tiles -- one workgroup per CU or fewer, every launch paying its ramp, its store
// drain and (for the 64-tile shapes) a split-K reduce of its own.  Three (self-attention: q, k, v
// of one layer) or 2 x layers (the cross-attention k / v MLPs of every decoder layer, whose input
// -- the encoder output -- is the same tensor) of them fill the chip as one grid.  The grid is
// flat: workgroup -> (group g, reduction split z, tile), tiles of one (g, z) consecutive so that
// an XCD's L2 sees neighbouring tiles of one problem.  Slab index = g * splitk + z; in "sum" mode
// (sum_g A_g B_g^T: the input gradient of MLPs that share their input) all groups * splits slabs
// reduce into one output.
// (GemmGroupPtrs / GEMM_MAX_GROUPS: gemm_epilogue.h)

template <bool AKC, bool BKC>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_dma_pf_grouped_kernel(GemmGroupPtrs gp, int64_t lda,
                                                                          int64_t ldb, GemmEpilogue ep, int M,
                                                                          int N, int K, int tiles_n, int tiles,
                                                                          int splitk, float* slabs,
                                                                          float* rs_part) {
    __shared__ __attribute__((aligned(16))) float lds[PF_STAGES * DMA_STAGE_FLOATS];   // 64 KB
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int per_g = tiles * splitk;
    const int g = lin / per_g;
    const int r = lin - g * per_g;
    const int z = r / tiles;
    const int tile = r - z * tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int per = K / splitk;                       // host: whole BK tiles per split
    const int k_begin = z * per;
    const int nk = per / BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* A = gp.A[g];
    const float* B = gp.B[g];
    ep.C = gp.C[g];
    ep.bias = gp.bias[g];
    ep.residual = gp.residual[g];
    ep.preact = gp.preact[g];
    ep.gradz = gp.gradz[g];
    const int slab = g * splitk + z;
    float* my_slab = slabs ? slabs + (int64_t)slab * M * N : nullptr;

    Acc acc;
    acc_zero(acc);
    float rs = 0.0f;
    const bool do_rs = !AKC && rs_part != nullptr && tn == 0 && tid < 128;
    pf_ring<AKC, BKC>(acc, rs, do_rs, lds, A, lda, B, ldb, m0, n0, k_begin, nk, wave, lane, tid);
    if (do_rs) rs_part[(int64_t)slab * M + m0 + tid] = rs;
    __syncthreads();                      // ring no longer in use: the epilogue stages through it
    // (1-D grid: blockIdx.z == 0, so the epilogue's slab offset is the one folded into my_slab)
    gemm_epilogue_wide<2>(acc, ep, lds, m0, n0, M, N, my_slab ? 2 : 1, my_slab);
}

// Reduce pass of a grouped launch.  blockIdx.y = output o, whose slabs are [o * nslab, (o + 1) *
// nslab) in ascending (fixed) order; blocks [0, nb) of x: 16-B columns of the M x N output, plain
// (+= when accumulate) or, EPI, through the full epilogue; the blocks behind them: the per-slab
// row sums (bias gradient) of the same output.
template <bool EPI>
__global__ __launch_bounds__(256) void slab_reduce_grouped_kernel(const float* __restrict__ slabs, GemmGroupPtrs gp,
                                                                  GemmEpilogue ep, int M, int N, int nslab,
                                                                  int accumulate, int nb,
                                                                  const float* __restrict__ rs_part) {
    const int o = blockIdx.y;
    if ((int)blockIdx.x >= nb) {
        float* rs_out = gp.rowsum[o];
        const int m = ((int)blockIdx.x - nb) * blockDim.x + threadIdx.x;
        if (m < M && rs_out) {
            const float* p = rs_part + (int64_t)o * nslab * M + m;
            float s = 0.0f;
            for (int zz = 0; zz < nslab; ++zz) s += p[(int64_t)zz * M];
            rs_out[m] = accumulate ? rs_out[m] + s : s;
        }
        return;
    }
    const int64_t total = (int64_t)M * N;
    const float* base = slabs + (int64_t)o * nslab * total;
    float* C = gp.C[o];
    const int n4 = N >> 2;
    const int64_t total4 = total >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)nb * blockDim.x) {
        float4 t = *reinterpret_cast<const float4*>(base + 4 * i);
        for (int zz = 1; zz < nslab; ++zz) {
            const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)zz * total + 4 * i);
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        const int64_t row = i / n4;
        const int col = (int)(i - row * n4) * 4;
        float4* dst = reinterpret_cast<float4*>(C + row * ep.ldc + col);
        if (!EPI) {
            if (accumulate) {
                const float4 c = *dst;
                t.x = c.x + t.x; t.y = c.y + t.y; t.z = c.z + t.z; t.w = c.w + t.w;
            }
            *dst = t;
            continue;
        }
        if (gp.bias[o]) {
            const float4 b = *reinterpret_cast<const float4*>(gp.bias[o] + col);
            t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
        }
        if (gp.residual[o]) {
            const float4 rv = *reinterpret_cast<const float4*>(gp.residual[o] + row * ep.ldr + col);
            t.x += rv.x; t.y += rv.y; t.z += rv.z; t.w += rv.w;
        }
        if (gp.preact[o]) *reinterpret_cast<float4*>(gp.preact[o] + row * ep.ldp + col) = t;
        float4 y = make_float4(act_fwd(t.x, ep.act), act_fwd(t.y, ep.act), act_fwd(t.z, ep.act),
                               act_fwd(t.w, ep.act));
        if (gp.gradz[o]) {
            const float4 zv = *reinterpret_cast<const float4*>(gp.gradz[o] + row * ep.ldz + col);
            y.x *= act_grad(zv.x, ep.gact); y.y *= act_grad(zv.y, ep.gact);
            y.z *= act_grad(zv.z, ep.gact); y.w *= act_grad(zv.w, ep.gact);
        }
        *dst = y;
    }
}

// out[i] (+ld handling) = sum_z slabs[z][i], z ascending: deterministic.
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                   int64_t ldc, int M, int N, int nslab, int accumulate) {
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int z = 0; z < nslab; ++z) s += slabs[(int64_t)z * total + i];
        const int64_t row = i / N, col = i - row * N;
        out[row * ldc + col] = accumulate ? out[row * ldc + col] + s : s;
    }
}

// The weight-gradient case in one launch: blocks [0, nb) reduce the slabs into C as above, the
// blocks behind them the per-split row sums (bias gradient) into rs_out.
__global__ void slab_reduce_rowsum_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                          int64_t ldc, int M, int N, int nslab, int accumulate, int nb,
                                          const float* __restrict__ rs_part, float* __restrict__ rs_out) {
    if ((int)blockIdx.x >= nb) {
        const int m = ((int)blockIdx.x - nb) * blockDim.x + threadIdx.x;
        if (m < M) {
            float s = 0.0f;
            for (int z = 0; z < nslab; ++z) s += rs_part[(int64_t)z * M + m];
            rs_out[m] = accumulate ? rs_out[m] + s : s;
        }
        return;
    }
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)nb * blockDim.x) {
        float s = 0.0f;
        for (int z = 0; z < nslab; ++z) s += slabs[(int64_t)z * total + i];
        const int64_t row = i / N, col = i - row * N;
        out[row * ldc + col] = accumulate ? out[row * ldc + col] + s : s;
    }
}

// Split-K reduce with the full GEMM epilogue: skinny-M GEMMs (autoregressive decode:
// M = a few dozen rows) have too few output tiles to fill the chip, so their reduction
// is split over grid.z and bias / residual / activation are applied here.
__global__ void slab_reduce_epilogue_kernel(const float* __restrict__ slabs, GemmEpilogue ep, int M,
                                            int N, int nslab) {
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        float t = 0.0f;
        for (int z = 0; z < nslab; ++z) t += slabs[(int64_t)z * total + i];
        const int64_t row = i / N, col = i - row * N;
        if (ep.bias) t += ep.bias[col];
        if (ep.residual) t += ep.residual[row * ep.ldr + col];
        if (ep.preact) ep.preact[row * ep.ldp + col] = t;
        float y = act_fwd(t, ep.act);
        if (ep.gradz) y *= act_grad(ep.gradz[row * ep.ldz + col], ep.gact);
        ep.C[row * ep.ldc + col] = y;
    }
}

// Column sums of X[M][N] (bias gradients; LayerNorm gamma/beta gradients).
// Stage 1: one block per (64 columns x COLSUM_ROWS rows), fixed order inside.
// Rows per block: enough blocks to fill the chip at a few thousand rows too (2,048 x 512 -- the
// LayerNorm-affine gradients of an 8-sequence shard -- ran 31 us on 32 blocks of 512 rows).
static inline int colsum_rows(int M) { return M >= 32768 ? 512 : (M >= 8192 ? 128 : 32); }
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X,
                                                             int64_t ldx, int M, int N,
                                                             float* __restrict__ part, int rows) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * rows;
    const int r1 = min(M, r0 + rows);
    float s = 0.0f;
    if (col < N)
        for (int r = r0 + ry; r < r1; r += 4) s += X[(int64_t)r * ldx + col];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < N)
        part[(int64_t)blockIdx.y * N + col] = ((red[0][cx] + red[1][cx]) + red[2][cx]) + red[3][cx];
}


// Skinny GEMM for few rows (single-token decode steps: sequences x beams x one token; 64 rows
// per workgroup pass, up to 512 rows through blockIdx.z; also the forward of the position-table
// projections, a few hundred rows).
// The 128x128 tile kernel leaves 7/8 of its MFMA rows empty there and needs a split-K
// round trip to find any parallelism (measured 16 us + 6 us reduce per call); this one
// is a weight-streaming kernel: a block owns 16 output columns, its 16 waves each own
// 1/16 of K and feed v_mfma_f32_16x16x4_f32 straight from global memory (A rows m on
// MFMA rows, W rows n on MFMA columns; each lane loads 16 B of a row, so lane groups
// cover 64 contiguous bytes), partial tiles meet in LDS and are summed in wave order
// (deterministic), then the usual epilogue runs on 256 threads per 16-row tile.
typedef float floatx4 __attribute__((ext_vector_type(4)));

// Decode-step fusions (the step is a chain of dependent ~5 us launches, so every launch removed is
// time): LN != 0 normalises the A rows on the way in -- LayerNorm over the K columns (K = the model
// width) in nn.LayerNorm's affine form (gamma/beta) or the AdaLN form scale(cond) * LN(x) + shift(cond)
// (csrc/norm.hip, same arithmetic: one wave per row, same summation order) -- so the LayerNorm launch
// in front of a block's first Linear disappears; `mul` multiplies the output elementwise (the
// residual layer's x * scale(cond) of the reference's ResidualLinearLayer, models/layers.py:258-304).
struct SkinnyFuse {
    const float* gamma;   // LN affine form (K)
    const float* beta;
    const float* scale;   // LN AdaLN form (M, K) rows at ldmod
    const float* shift;
    int64_t ldmod;
    float eps;
    const float* mul;     // (M, N) rows at ldmul, or null
    int64_t ldmul;
};

__device__ __forceinline__ float4 ln_apply(float4 a, float mean, float rstd, float4 g, float4 b) {
    return make_float4(((a.x - mean) * rstd) * g.x + b.x, ((a.y - mean) * rstd) * g.y + b.y,
                       ((a.z - mean) * rstd) * g.z + b.z, ((a.w - mean) * rstd) * g.w + b.w);
}

template <int MT, int LN>   // LN: 0 none, 1 gamma/beta, 2 scale/shift rows
__global__ __launch_bounds__(1024) void gemm_skinny_kernel(const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb,
                                                           GemmEpilogue ep, int M, int N, int K,
                                                           int64_t a_gs, int64_t b_gs, int64_t c_gs,
                                                           int64_t bias_gs, SkinnyFuse fz) {
    __shared__ float part[16][MT * 256];
    __shared__ float stat[64][2];
    // grouped form: blockIdx.y selects an independent problem at fixed operand strides
    A += blockIdx.y * a_gs;
    B += blockIdx.y * b_gs;
    ep.C += blockIdx.y * c_gs;
    if (ep.bias) ep.bias += blockIdx.y * bias_gs;
    // more than 64 rows: blockIdx.z walks 64-row slabs (the weights come back from L2)
    const int mz = blockIdx.z * 64;
    A += (int64_t)mz * lda;
    ep.C += (int64_t)mz * ep.ldc;
    if (ep.residual) ep.residual += (int64_t)mz * ep.ldr;
    if (ep.preact) ep.preact += (int64_t)mz * ep.ldp;
    if (ep.gradz) ep.gradz += (int64_t)mz * ep.ldz;
    if (LN == 2) { fz.scale += (int64_t)mz * fz.ldmod; fz.shift += (int64_t)mz * fz.ldmod; }
    if (fz.mul) fz.mul += (int64_t)mz * fz.ldmul;
    M = min(M - mz, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int kslice = K >> 4;
    const int col = lane & 15, kg = lane >> 4;
    const int64_t koff = (int64_t)wave * kslice + kg * 4;
    const float* bp = B + (int64_t)min(n0 + col, N - 1) * ldb + koff;
    const float* ap[MT];
    bool aok[MT];
    floatx4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + col;
        aok[t] = m < M;
        ap[t] = A + (int64_t)(aok[t] ? m : 0) * lda + koff;
        acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    // U k-steps of loads are issued back to back before their MFMAs: the kernel is pure
    // memory latency (a wave's whole K slice is 2..8 such steps), so the loads must overlap
    constexpr int U = LN ? (MT <= 2 ? 2 : 1) : (MT == 1 ? 8 : (MT == 2 ? 4 : 2));
    float mean[MT], rstd[MT];
    float4 bpre[U];       // the first k-steps' weights: in flight while the row statistics are computed
#pragma unroll
    for (int u = 0; u < U; ++u)
        bpre[u] = 16 * u < kslice ? *reinterpret_cast<const float4*>(bp + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (LN) {
        // row statistics, one wave per row as layernorm_fwd_kernel does them (rows of this 64-row
        // slab round-robin over the 16 waves)
        for (int r = wave; r < M; r += 16) {
            const float* xr = A + (int64_t)r * lda;
            float s = 0.0f;
            for (int c = lane * 4; c < K; c += 256) {
                const float4 v = *reinterpret_cast<const float4*>(xr + c);
                s += (v.x + v.y) + (v.z + v.w);
            }
            const float mu = wave_sum(s) / (float)K;
            float q = 0.0f;
            for (int c = lane * 4; c < K; c += 256) {
                float4 v = *reinterpret_cast<const float4*>(xr + c);
                v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
                q = fmaf(v.x, v.x, q); q = fmaf(v.y, v.y, q);
                q = fmaf(v.z, v.z, q); q = fmaf(v.w, v.w, q);
            }
            const float rs = 1.0f / sqrtf(wave_sum(q) / (float)K + fz.eps);
            if (lane == 0) { stat[r][0] = mu; stat[r][1] = rs; }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            mean[t] = aok[t] ? stat[t * 16 + col][0] : 0.0f;
            rstd[t] = aok[t] ? stat[t * 16 + col][1] : 0.0f;
        }
    }
    for (int k0 = 0; k0 < kslice; k0 += 16 * U) {
        float4 b[U], a[MT][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 16 * u;
            const bool in = k < kslice;          // wave-uniform
            b[u] = k0 == 0 ? bpre[u]
                           : (in ? *reinterpret_cast<const float4*>(bp + k) : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
            for (int t = 0; t < MT; ++t)
                a[t][u] = (in && aok[t]) ? *reinterpret_cast<const float4*>(ap[t] + k)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            if (LN == 1 && in) {
                const float4 g = *reinterpret_cast<const float4*>(fz.gamma + koff + k);
                const float4 h = *reinterpret_cast<const float4*>(fz.beta + koff + k);
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    if (aok[t]) a[t][u] = ln_apply(a[t][u], mean[t], rstd[t], g, h);
            }
            if (LN == 2 && in) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    if (aok[t]) {
                        const int64_t mo = (int64_t)(t * 16 + col) * fz.ldmod + koff + k;
                        const float4 g = *reinterpret_cast<const float4*>(fz.scale + mo);
                        const float4 h = *reinterpret_cast<const float4*>(fz.shift + mo);
                        a[t][u] = ln_apply(a[t][u], mean[t], rstd[t], g, h);
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u].x, b[u].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u].y, b[u].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u].z, b[u].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u].w, b[u].w, acc[t], 0, 0, 0);
            }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][t * 256 + (4 * kg + r) * 16 + col] = acc[t][r];
    __syncthreads();
    for (int idx = threadIdx.x; idx < MT * 256; idx += 1024) {
        const int m = (idx >> 8) * 16 + ((idx & 255) >> 4);
        const int n = n0 + (idx & 15);
        if (m >= M || n >= N) continue;
        float v = part[0][idx];
#pragma unroll
        for (int w = 1; w < 16; ++w) v += part[w][idx];
        if (ep.bias) v += ep.bias[n];
        if (ep.residual) v += ep.residual[(int64_t)m * ep.ldr + n];
        if (ep.preact) ep.preact[(int64_t)m * ep.ldp + n] = v;
        float y = act_fwd(v, ep.act);
        if (ep.gradz) y *= act_grad(ep.gradz[(int64_t)m * ep.ldz + n], ep.gact);
        if (fz.mul) y *= fz.mul[(int64_t)m * fz.ldmul + n];
        ep.C[(int64_t)m * ep.ldc + n] = y;
    }
}

}  // namespace qarig

using namespace qarig;

extern "C" size_t qarig_gemm_workspace_bytes(int M, int N, int splitk) {
    if (M < 1 || N < 1 || splitk > (1 << 16)) return 0;
    // split-K slabs + (always) room for the per-split A row sums
    const size_t sk = splitk > 1 ? splitk : 1;
    return (splitk > 1 ? sk * M * N * sizeof(float) : 0) + sk * M * sizeof(float);
}


// decode batches (sequences x beams) up to this many rows stay on the weight-streaming kernel
static constexpr int SKINNY_MAX_ROWS = 512;

static void launch_skinny(const float* A, int64_t lda, const float* B, int64_t ldb, const GemmEpilogue& eps,
                          int M, int N, int K, int64_t a_gs, int64_t b_gs, int64_t c_gs,
                          int64_t bias_gs, int groups, hipStream_t st, const SkinnyFuse& fz = SkinnyFuse{}) {
    dim3 sgrid((N + 15) / 16, groups, (M + 63) / 64), sblock(1024);
    const int mt = M > 64 ? 4 : (M + 15) / 16;
    const int ln = fz.gamma ? 1 : (fz.scale ? 2 : 0);
#define QARIG_SKINNY(MT, LN)                                                                           \
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, LN>), sgrid, sblock, 0, st, A, lda, B, ldb, eps, M, N, K, \
                       a_gs, b_gs, c_gs, bias_gs, fz)
#define QARIG_SKINNY_MT(LN)                                                                            \
    switch (mt) {                                                                                      \
        case 1: QARIG_SKINNY(1, LN); break;                                                            \
        case 2: QARIG_SKINNY(2, LN); break;                                                            \
        case 3: QARIG_SKINNY(3, LN); break;                                                            \
        default: QARIG_SKINNY(4, LN); break;                                                           \
    }
    if (ln == 0) { QARIG_SKINNY_MT(0) } else if (ln == 1) { QARIG_SKINNY_MT(1) } else { QARIG_SKINNY_MT(2) }
#undef QARIG_SKINNY_MT
#undef QARIG_SKINNY
}

// csrc/gemm64.hip
int qarig_gemm64_launch(const float* A, int64_t lda, int a_kcontig, const float* B, int64_t ldb, int b_kcontig,
                        const GemmEpilogue& ep, int M, int N, int K, int splitk, float* slabs, hipStream_t st);

// csrc/gemm_x3.hip
extern "C" int qarig_gemm_x3_ok(int M, int N, int K, int splitk);
int qarig_gemm_x3_launch(const float* A, int64_t lda, int a_kcontig, const float* B, int64_t ldb, int b_kcontig,
                         const GemmEpilogue& ep, int M, int N, int K, int splitk, float* slabs, hipStream_t st);
int qarig_gemm_x3_half_ok(int M, int N, int K, int splitk);
int qarig_gemm_x3_half_launch(const float* A, int64_t lda, int a_kcontig, const float* B, int64_t ldb, int b_kcontig,
                              const GemmEpilogue& ep, int M, int N, int K, int splitk, float* slabs, hipStream_t st);
int qarig_gemm_x3_grouped_launch(const GemmGroupPtrs& gp, int64_t lda, int a_kcontig, int64_t ldb, int b_kcontig,
                                 const GemmEpilogue& ep, int M, int N, int K, int tiles_n, int tiles, int splitk,
                                 float* slabs, float* rs_part, unsigned total_wg, hipStream_t st);

static int gemm_dispatch(const float* A, int64_t lda, int a_kcontig, const float* B,
                         int64_t ldb, int b_kcontig, float* C, int64_t ldc, int M, int N,
                         int K, const float* bias, const float* residual, int64_t ldr,
                         float* preact, int64_t ldp, int act, const float* gradz,
                         int64_t ldz, int gact, int splitk, int accumulate, float* a_rowsum,
                         void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(A && B && C, "gemm: null operand");
    QARIG_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad extents M=%d N=%d K=%d", M, N, K);
    QARIG_CHECK_DIMS("gemm", M, N);
    QARIG_CHECK_DIMS("gemm", M, K);
    QARIG_CHECK_DIMS("gemm", N, K);
    QARIG_CHECK_ARG(splitk <= 4096, "gemm: splitk %d too large", splitk);
    QARIG_CHECK_ARG(act >= 0 && act <= 3 && gact >= 0 && gact <= 3, "gemm: bad activation id");
    if (splitk < 1) splitk = 1;
    if (accumulate) {
        // C += A B^T (gradient accumulation straight into a parameter's .grad)
        QARIG_CHECK_ARG(!bias && !residual && !preact && !gradz && act == ACT_NONE,
                        "gemm: accumulate supports the plain epilogue only");
        if (splitk == 1) { residual = C; ldr = ldc; }   // read-modify-write by the same lane
    }
    const bool plain = !bias && !residual && !preact && !gradz && act == ACT_NONE;
    if (splitk > 1 || a_rowsum) {
        if (ws_bytes < qarig_gemm_workspace_bytes(M, N, splitk) || !workspace) {
            qarig_set_error("gemm: workspace too small (%zu < %zu)", ws_bytes,
                            qarig_gemm_workspace_bytes(M, N, splitk));
            return QARIG_ERR_WORKSPACE;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (M <= SKINNY_MAX_ROWS && a_kcontig && b_kcontig && K % 256 == 0 && !accumulate && !a_rowsum &&
        al16(A) && al16(B) && lda % 4 == 0 && ldb % 4 == 0) {
        if (g_qarig_opt.decode_stream && !preact && !gradz && qarig_decode_linear_supported(M, N, K, 0) && lda >= K &&
            ldb >= K && ldc >= N && (!residual || ldr >= N))
            return qarig_decode_linear_f32(A, lda, 0, 0.0f, nullptr, nullptr, nullptr, nullptr, 0, B, ldb, 0, bias, 0,
                                           residual, ldr, nullptr, 0, C, ldc, 0, 1, M, N, K, act, stream);
        GemmEpilogue eps{C, ldc, bias, residual, ldr, preact, ldp, act, gradz, ldz, gact, nullptr};
        launch_skinny(A, lda, B, ldb, eps, M, N, K, 0, 0, 0, 0, 1, st);
        QARIG_CHECK_LAUNCH("gemm skinny");
        return QARIG_OK;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, splitk), block(NTHREADS);
    float* slabs = (float*)workspace;
    // per-split A row sums live behind the slabs
    float* rs_part = a_rowsum ? slabs + (splitk > 1 ? (size_t)splitk * M * N : 0) : nullptr;
    GemmEpilogue ep{C, ldc, bias, residual, ldr, preact, ldp, act, gradz, ldz, gact, rs_part};
    const bool va = al16(A) && lda % 4 == 0, vb = al16(B) && ldb % 4 == 0;
    int per = K;
    if (splitk > 1) per = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
    const bool fast = va && vb && M % BM == 0 && N % BN == 0 && K % BK == 0 &&
                      (splitk == 1 || K % per == 0);
    auto ok4 = [&](const void* p, int64_t ld) { return !p || (al16(p) && ld % 4 == 0); };
    const int vec_epi = ok4(C, ldc) && ok4(bias, 4) && ok4(residual, ldr) && ok4(preact, ldp) &&
                        ok4(gradz, ldz) && ok4(slabs, 4);
    const bool dma_on = g_qarig_opt.gemm_dma != 0;
    // every interior shape runs the fragment-prefetch ring (gemm_dma_pf_kernel: 4 stages, addresses off
    // the vector ALU); qarig_set_option("gemm_dma", 0) sends them to the register-staged kernel instead
    const bool pf = dma_on;
    // paired form (two 4-wave teams per tile, 128 KB of LDS): launches that would leave one
    // workgroup per CU; option gemm_pair = 0 disables it, 1 forces it wherever it is eligible
    const int pair_env = g_qarig_opt.gemm_pair;
    const int nk_block = per / BK;
    // 64 x 64 tiles (gemm64.hip) where 128 x 128 tiles would leave most CUs without a workgroup: the 512-wide
    // products of a 2,048-row shard, the window evaluations of sliding-window generation
    const bool t64 = g_qarig_opt.gemm_tile64 != 0 && va && vb && M % 64 == 0 && N % 64 == 0 && K % BK == 0 &&
                     (splitk == 1 || (K % per == 0 && per % BK == 0)) && !(a_rowsum && a_kcontig) &&
                     qarig_gemm_tile64(M, N, K);
    const bool pair_ok = dma_on && fast && vec_epi && !(a_rowsum && a_kcontig) && !(!a_kcontig && b_kcontig) &&
                         nk_block % 2 == 0 && nk_block >= 4;
    // opt-in: the products on the bf16 matrix pipe from exact three-way operand splits (gemm_x3.hip)
    // (from 32 output tiles up: below that the two-team and 64-tile kernels are faster -- 512 x 512 over 16,384 rows:
    //  52 against 41 TF-equivalent)
    const bool x3 = g_qarig_opt.gemm_x3 != 0 && fast && vec_epi && !(a_rowsum && a_kcontig) &&
                    (long)tiles_m * tiles_n >= 32 && qarig_gemm_x3_ok(M, N, K, splitk);
    // ... on 64 x 64 tiles where 128 x 128 tiles would leave most of the chip idle
    // (fewer than 192 workgroups of 128 x 128 tiles, reduction splits counted: a weight gradient of 64 tiles x 16
    //  splits fills the chip with the larger tile, whose split work per MFMA is half)
    const bool x3h = g_qarig_opt.gemm_x3 == 1 && va && vb && vec_epi && !(a_rowsum && a_kcontig) &&
                     (long)tiles_m * tiles_n * splitk < 192 && (long)(M / 64) * (N / 64) * splitk >= 32 &&
                     qarig_gemm_x3_half_ok(M, N, K, splitk);
    if (x3h) {
        qarig_gemm_x3_half_launch(A, lda, a_kcontig, B, ldb, b_kcontig, ep, M, N, K, splitk, slabs, st);
    } else if (x3) {
        qarig_gemm_x3_launch(A, lda, a_kcontig, B, ldb, b_kcontig, ep, M, N, K, splitk, slabs, st);
    } else if (t64 && pair_env != 1) {    // (gemm_pair = 1 forces the two-team kernel: the cross-check of this one)
        qarig_gemm64_launch(A, lda, a_kcontig, B, ldb, b_kcontig, ep, M, N, K, splitk, slabs, st);
    } else if (pair_ok && pair_env != 0 && (pair_env == 1 || (long)grid.x * grid.z <= 256)) {
        constexpr int PAIR_LDS = 2 * PF_STAGES * DMA_STAGE_FLOATS * (int)sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)gemm_dma_pf2_kernel<true, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS);
            (void)hipFuncSetAttribute((const void*)gemm_dma_pf2_kernel<true, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS);
            (void)hipFuncSetAttribute((const void*)gemm_dma_pf2_kernel<false, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS);
            attr_set = true;
        }
        dim3 block2(2 * NTHREADS);
        if (a_kcontig && b_kcontig)
            hipLaunchKernelGGL((gemm_dma_pf2_kernel<true, true>), grid, block2, PAIR_LDS, st, A, lda, B, ldb, ep,
                               M, N, K, tiles_n, splitk, slabs);
        else if (a_kcontig)
            hipLaunchKernelGGL((gemm_dma_pf2_kernel<true, false>), grid, block2, PAIR_LDS, st, A, lda, B, ldb, ep,
                               M, N, K, tiles_n, splitk, slabs);
        else
            hipLaunchKernelGGL((gemm_dma_pf2_kernel<false, false>), grid, block2, PAIR_LDS, st, A, lda, B, ldb, ep,
                               M, N, K, tiles_n, splitk, slabs);
    } else if (pf && fast && vec_epi && !(a_rowsum && a_kcontig) && !(!a_kcontig && b_kcontig)) {
        // one reduction split per XCD (see the kernel) where the splits are a multiple of the 8 XCDs
        const int xs = (splitk % 8 == 0 && g_qarig_opt.gemm_xcd_splits != 0) ? 1 : 0;
        const dim3 pgrid = xs ? dim3(grid.x * splitk, 1, 1) : grid;
        if (a_kcontig && b_kcontig)
            hipLaunchKernelGGL((gemm_dma_pf_kernel<true, true>), pgrid, block, 0, st, A, lda, B, ldb, ep, M,
                               N, K, tiles_n, splitk, slabs, xs);
        else if (a_kcontig)
            hipLaunchKernelGGL((gemm_dma_pf_kernel<true, false>), pgrid, block, 0, st, A, lda, B, ldb, ep, M,
                               N, K, tiles_n, splitk, slabs, xs);
        else
            hipLaunchKernelGGL((gemm_dma_pf_kernel<false, false>), pgrid, block, 0, st, A, lda, B, ldb, ep,
                               M, N, K, tiles_n, splitk, slabs, xs);
    } else {
#define QARIG_LAUNCH_GEMM(TA, TB)                                                              \
    do {                                                                                       \
        TA sa{A, lda, M, K, 1.0f, va};                                                         \
        TB sb{B, ldb, N, K, 1.0f, vb};                                                         \
        if (fast)                                                                              \
            hipLaunchKernelGGL((gemm_kernel<TA, TB, true>), grid, block, 0, st, sa, sb, ep, M, \
                               N, K, tiles_n, splitk, slabs, vec_epi);                         \
        else                                                                                   \
            hipLaunchKernelGGL((gemm_kernel<TA, TB, false>), grid, block, 0, st, sa, sb, ep, M,\
                               N, K, tiles_n, splitk, slabs, 0);                               \
    } while (0)
    if (a_kcontig && b_kcontig) QARIG_LAUNCH_GEMM(SrcKContig, SrcKContig);
    else if (a_kcontig && !b_kcontig) QARIG_LAUNCH_GEMM(SrcKContig, SrcXContig);
    else if (!a_kcontig && !b_kcontig) QARIG_LAUNCH_GEMM(SrcXContig, SrcXContig);
    else QARIG_LAUNCH_GEMM(SrcXContig, SrcKContig);
#undef QARIG_LAUNCH_GEMM
    }
    QARIG_CHECK_LAUNCH("gemm");
    if (a_rowsum && splitk > 1 && plain) {   // both reductions of a weight-gradient GEMM in one launch
        const int64_t total = (int64_t)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(slab_reduce_rowsum_kernel, dim3(blocks + (M + 255) / 256), dim3(256), 0, st, slabs, C,
                           ldc, M, N, splitk, accumulate, blocks, rs_part, a_rowsum);
        QARIG_CHECK_LAUNCH("gemm slab + rowsum reduce");
        return QARIG_OK;
    }
    if (a_rowsum) {   // a_rowsum[m] (+)= sum over splits, fixed order
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((M + 255) / 256), dim3(256), 0, st, rs_part,
                           a_rowsum, (int64_t)M, 1, M, splitk, accumulate);
        QARIG_CHECK_LAUNCH("gemm rowsum reduce");
    }
    if (splitk > 1) {
        const int64_t total = (int64_t)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (plain)
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, slabs, C, ldc, M, N,
                               splitk, accumulate);
        else
            hipLaunchKernelGGL(slab_reduce_epilogue_kernel, dim3(blocks), dim3(256), 0, st, slabs, ep,
                               M, N, splitk);
        QARIG_CHECK_LAUNCH("gemm slab reduce");
    }
    return QARIG_OK;
}

// Internal helper shared with conv.hip: out[M][N] (ld ldc) = sum_z slabs[z][M][N].
extern "C" int qarig_slab_reduce_f32(const float* slabs, float* out, int64_t ldc, int M, int N,
                                     int nslab, int accumulate, void* stream) {
    const int64_t total = (int64_t)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, out,
                       ldc, M, N, nslab, accumulate);
    QARIG_CHECK_LAUNCH("slab reduce");
    return QARIG_OK;
}

extern "C" size_t qarig_colsum_workspace_bytes(int M, int N) {
    if (M < 1 || N < 1 || M > (1 << 30)) return 0;
    const int rows = colsum_rows(M);
    const int chunks = (M + rows - 1) / rows;
    return (size_t)chunks * N * sizeof(float);
}

// out[N] = sum over rows of X[M][N]; fixed summation order (bit-reproducible).
extern "C" int qarig_colsum_f32(const float* X, int64_t ldx, int M, int N, float* out,
                                int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(X && out && M > 0 && N > 0, "colsum: bad arguments");
    QARIG_CHECK_DIMS("colsum", M, N);
    if (!workspace || ws_bytes < qarig_colsum_workspace_bytes(M, N)) {
        qarig_set_error("colsum: workspace too small");
        return QARIG_ERR_WORKSPACE;
    }
    const int rows = colsum_rows(M);
    const int chunks = (M + rows - 1) / rows;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, st, X, ldx,
                       M, N, part, rows);
    QARIG_CHECK_LAUNCH("colsum partial");
    int blocks = (N + 255) / 256;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, part, out, (int64_t)N, 1,
                       N, chunks, accumulate);
    QARIG_CHECK_LAUNCH("colsum reduce");
    return QARIG_OK;
}

extern "C" int qarig_gemm_f32(const float* A, int64_t lda, int a_kcontig, const float* B,
                              int64_t ldb, int b_kcontig, float* C, int64_t ldc, int M, int N,
                              int K, const float* bias, const float* residual, int64_t ldr,
                              float* preact, int64_t ldp, int act, const float* gradz,
                              int64_t ldz, int gact, int splitk, int accumulate, float* a_rowsum,
                              void* workspace, size_t ws_bytes, void* stream) {
    return gemm_dispatch(A, lda, a_kcontig, B, ldb, b_kcontig, C, ldc, M, N, K, bias, residual, ldr,
                         preact, ldp, act, gradz, ldz, gact, splitk, accumulate, a_rowsum, workspace,
                         ws_bytes, stream);
}

// ---- grouped interior GEMM (see gemm_dma_pf_grouped_kernel) ------------------------------------
extern "C" int qarig_gemm_grouped_supported(int M, int N, int K, int splitk) {
    if (M < 1 || N < 1 || K < 1 || splitk < 1 || splitk > 64) return 0;
    if (M % BM || N % BN || K % splitk || (K / splitk) % BK || K / splitk < BK) return 0;
    return qarig_dims_ok({M, N}) && qarig_dims_ok({M, K}) && qarig_dims_ok({N, K}) ? 1 : 0;
}

extern "C" size_t qarig_gemm_grouped_workspace_bytes(int groups, int M, int N, int splitk, int sum_groups) {
    if (groups < 1 || groups > GEMM_MAX_GROUPS || M < 1 || N < 1 || splitk < 1 || splitk > 64) return 0;
    const size_t ns = (size_t)groups * splitk;
    const bool slabs = splitk > 1 || sum_groups;
    return (slabs ? ns * M * N * sizeof(float) : 0) + ns * M * sizeof(float);
}

extern "C" int qarig_gemm_f32_grouped(int groups, const float* const* A, int64_t lda, int a_kcontig,
                                      const float* const* B, int64_t ldb, int b_kcontig, float* const* C,
                                      int64_t ldc, int M, int N, int K, const float* const* bias,
                                      const float* const* residual, int64_t ldr, float* const* preact,
                                      int64_t ldp, int act, const float* const* gradz, int64_t ldz, int gact,
                                      int splitk, int accumulate, int sum_groups, float* const* a_rowsum,
                                      void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(groups >= 1 && groups <= GEMM_MAX_GROUPS, "gemm_grouped: 1 <= groups <= %d (got %d)",
                    GEMM_MAX_GROUPS, groups);
    QARIG_CHECK_ARG(A && B && C, "gemm_grouped: null operand table");
    if (splitk < 1) splitk = 1;
    QARIG_CHECK_ARG(qarig_gemm_grouped_supported(M, N, K, splitk),
                    "gemm_grouped: needs M %% 128 == 0, N %% 128 == 0 and whole 16-deep tiles per split "
                    "(M=%d N=%d K=%d splitk=%d)", M, N, K, splitk);
    QARIG_CHECK_ARG(act >= 0 && act <= 3 && gact >= 0 && gact <= 3, "gemm_grouped: bad activation id");
    QARIG_CHECK_ARG(a_kcontig || !b_kcontig, "gemm_grouped: the (xc, kc) layout is not on the hot path");
    QARIG_CHECK_ARG(!a_rowsum || !a_kcontig, "gemm_grouped: a_rowsum needs A stored [K][M]");
    const bool plain = !bias && !residual && !preact && !gradz && act == ACT_NONE;
    QARIG_CHECK_ARG(!(accumulate || sum_groups) || plain,
                    "gemm_grouped: accumulate / sum_groups support the plain epilogue only");
    QARIG_CHECK_ARG(!(sum_groups && a_rowsum), "gemm_grouped: sum_groups has no row sums");
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    QARIG_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && ldr % 4 == 0 && ldp % 4 == 0 && ldz % 4 == 0,
                    "gemm_grouped: row strides must be multiples of 4 elements");
    const int n_out = sum_groups ? 1 : groups;
    GemmGroupPtrs gp;
    for (int g = 0; g < GEMM_MAX_GROUPS; ++g) {
        const int s = g < groups ? g : 0;
        gp.A[g] = A[s];
        gp.B[g] = B[s];
        gp.C[g] = C[sum_groups ? 0 : s];
        gp.bias[g] = bias ? bias[s] : nullptr;
        gp.residual[g] = residual ? residual[s] : nullptr;
        gp.preact[g] = preact ? preact[s] : nullptr;
        gp.gradz[g] = gradz ? gradz[s] : nullptr;
        gp.rowsum[g] = a_rowsum ? a_rowsum[s] : nullptr;
        if (g < groups) {
            QARIG_CHECK_ARG(gp.A[g] && gp.B[g] && (gp.C[g] || g >= n_out), "gemm_grouped: null operand in group %d", g);
            QARIG_CHECK_ARG(al16(gp.A[g]) && al16(gp.B[g]) && al16(gp.C[g]) && al16(gp.bias[g]) &&
                                al16(gp.residual[g]) && al16(gp.preact[g]) && al16(gp.gradz[g]),
                            "gemm_grouped: operands must be 16-B aligned (group %d)", g);
        }
    }
    const bool use_slabs = splitk > 1 || sum_groups;
    const size_t need = qarig_gemm_grouped_workspace_bytes(groups, M, N, splitk, sum_groups);
    if ((use_slabs || a_rowsum) && (!workspace || ws_bytes < need)) {
        qarig_set_error("gemm_grouped: workspace too small (%zu < %zu)", ws_bytes, need);
        return QARIG_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
    const long total_wg = (long)tiles * splitk * groups;
    QARIG_CHECK_ARG(total_wg <= (1L << 30), "gemm_grouped: grid too large");
    float* slabs = use_slabs ? (float*)workspace : nullptr;
    float* rs_part = a_rowsum ? (float*)workspace + (use_slabs ? (size_t)groups * splitk * M * N : 0) : nullptr;
    if (accumulate && !use_slabs)      // C += A B^T: read-modify-write by the same lane
        for (int g = 0; g < GEMM_MAX_GROUPS; ++g) gp.residual[g] = gp.C[g];
    GemmEpilogue ep{nullptr, ldc, nullptr, nullptr, accumulate && !use_slabs ? ldc : ldr, nullptr, ldp, act,
                    nullptr, ldz, gact, nullptr};
    dim3 grid((unsigned)total_wg), block(NTHREADS);
    // opt-in: the grouped products on the bf16 matrix pipe from exact three-way operand splits (gemm_x3.hip)
    if (g_qarig_opt.gemm_x3 != 0 && (K / splitk) % 32 == 0 && total_wg >= 32)
        qarig_gemm_x3_grouped_launch(gp, lda, a_kcontig, ldb, b_kcontig, ep, M, N, K, tiles_n, tiles, splitk, slabs,
                                     rs_part, (unsigned)total_wg, st);
    else if (a_kcontig && b_kcontig)
        hipLaunchKernelGGL((gemm_dma_pf_grouped_kernel<true, true>), grid, block, 0, st, gp, lda, ldb, ep, M, N, K,
                           tiles_n, tiles, splitk, slabs, rs_part);
    else if (a_kcontig)
        hipLaunchKernelGGL((gemm_dma_pf_grouped_kernel<true, false>), grid, block, 0, st, gp, lda, ldb, ep, M, N, K,
                           tiles_n, tiles, splitk, slabs, rs_part);
    else
        hipLaunchKernelGGL((gemm_dma_pf_grouped_kernel<false, false>), grid, block, 0, st, gp, lda, ldb, ep, M, N,
                           K, tiles_n, tiles, splitk, slabs, rs_part);
    QARIG_CHECK_LAUNCH("gemm_grouped");
    if (use_slabs || a_rowsum) {
        const int nslab = sum_groups ? groups * splitk : splitk;
        int nb = 0;
        if (use_slabs) {
            const int64_t total4 = (int64_t)M * N / 4;
            nb = (int)((total4 + 255) / 256);
            const int cap = n_out >= 4 ? 1024 : 2048;
            if (nb > cap) nb = cap;
        }
        const int rb = a_rowsum ? (M + 255) / 256 : 0;
        dim3 rgrid(nb + rb, n_out);
        if (plain)
            hipLaunchKernelGGL((slab_reduce_grouped_kernel<false>), rgrid, dim3(256), 0, st, slabs, gp, ep, M, N,
                               nslab, accumulate, nb, rs_part);
        else
            hipLaunchKernelGGL((slab_reduce_grouped_kernel<true>), rgrid, dim3(256), 0, st, slabs, gp, ep, M, N,
                               nslab, accumulate, nb, rs_part);
        QARIG_CHECK_LAUNCH("gemm_grouped reduce");
    }
    return QARIG_OK;
}

// `groups` independent skinny products in one launch (decode step: the q/k/v MLPs of an
// attention layer, and every projection of the conditioning vector, models/layers.py:100-153,
// 258-304, 389-418):  C_g = act(A_g W_g^T + bias_g),  X_g = X + g * x_gs.  a_gs == 0 shares
// the activations.  M <= 64 rows, K % 256 == 0, reduction-contiguous 16-B aligned operands.
extern "C" int qarig_gemm_grouped_skinny_f32(const float* A, int64_t lda, int64_t a_gs,
                                             const float* W, int64_t ldw, int64_t w_gs, float* C,
                                             int64_t ldc, int64_t c_gs, const float* bias,
                                             int64_t bias_gs, int groups, int M, int N, int K,
                                             int act, void* stream) {
    QARIG_CHECK_ARG(A && W && C, "gemm_grouped_skinny: null operand");
    QARIG_CHECK_ARG(groups > 0 && groups <= 65535 && M > 0 && M <= SKINNY_MAX_ROWS && N > 0 && K > 0 && K % 256 == 0,
                    "gemm_grouped_skinny: needs 0 < M <= 512, K %% 256 == 0 (M=%d N=%d K=%d groups=%d)",
                    M, N, K, groups);
    QARIG_CHECK_ARG(act >= 0 && act <= 3, "gemm_grouped_skinny: bad activation id");
    QARIG_CHECK_ARG((((uintptr_t)A | (uintptr_t)W) & 15) == 0 && lda % 4 == 0 && ldw % 4 == 0 &&
                        a_gs % 4 == 0 && w_gs % 4 == 0,
                    "gemm_grouped_skinny: operands must be 16-B aligned");
    if (g_qarig_opt.decode_stream && qarig_decode_linear_supported(M, N, K, 0) && lda >= K && ldw >= K && ldc >= N)
        return qarig_decode_linear_f32(A, lda, a_gs, 0.0f, nullptr, nullptr, nullptr, nullptr, 0, W, ldw, w_gs, bias,
                                       bias_gs, nullptr, 0, nullptr, 0, C, ldc, c_gs, groups, M, N, K, act, stream);
    GemmEpilogue eps{C, ldc, bias, nullptr, 0, nullptr, 0, act, nullptr, 0, 0, nullptr};
    launch_skinny(A, lda, W, ldw, eps, M, N, K, a_gs, w_gs, c_gs, bias_gs, groups, (hipStream_t)stream);
    QARIG_CHECK_LAUNCH("gemm_grouped_skinny");
    return QARIG_OK;
}

// The decode step's fused form of the same launch: C_g = act(LN(X) W_g^T + bias_g) [* mul], the
// activations X (M, K) shared by the groups and LayerNorm'ed over K on the way in -- affine form
// (gamma, beta) or AdaLN form (scale, shift rows at ldmod), models/layers.py:130-153 / nn.LayerNorm;
// all four null: no normalisation.  mul (M, N) at ldmul, optional: elementwise factor on the output
// (ResidualLinearLayer's x * scale(cond), models/layers.py:258-304), the same for every group.
extern "C" int qarig_gemm_skinny_ln_f32(const float* X, int64_t ldx, float eps, const float* gamma,
                                        const float* beta, const float* scale, const float* shift,
                                        int64_t ldmod, const float* W, int64_t ldw, int64_t w_gs,
                                        float* C, int64_t ldc, int64_t c_gs, const float* bias,
                                        int64_t bias_gs, const float* mul, int64_t ldmul, int groups,
                                        int M, int N, int K, int act, void* stream) {
    QARIG_CHECK_ARG(X && W && C, "gemm_skinny_ln: null operand");
    QARIG_CHECK_ARG(groups > 0 && groups <= 65535 && M > 0 && M <= SKINNY_MAX_ROWS && N > 0 && K > 0 && K % 256 == 0,
                    "gemm_skinny_ln: needs 0 < M <= 512, K %% 256 == 0 (M=%d N=%d K=%d groups=%d)",
                    M, N, K, groups);
    QARIG_CHECK_ARG(act >= 0 && act <= 3, "gemm_skinny_ln: bad activation id");
    QARIG_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "gemm_skinny_ln: gamma/beta pair");
    QARIG_CHECK_ARG((scale == nullptr) == (shift == nullptr), "gemm_skinny_ln: scale/shift pair");
    QARIG_CHECK_ARG(!(gamma && scale), "gemm_skinny_ln: affine and AdaLN forms are exclusive");
    QARIG_CHECK_ARG((((uintptr_t)X | (uintptr_t)W | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)scale |
                      (uintptr_t)shift) & 15) == 0 && ldx % 4 == 0 && ldw % 4 == 0 && w_gs % 4 == 0 &&
                        (!scale || ldmod % 4 == 0),
                    "gemm_skinny_ln: operands must be 16-B aligned");
    QARIG_CHECK_ARG(!(gamma || scale) || eps > 0.0f, "gemm_skinny_ln: eps must be positive");
    if (g_qarig_opt.decode_stream && qarig_decode_linear_supported(M, N, K, (gamma || scale) ? 1 : 0) && ldx >= K &&
        ldw >= K && ldc >= N && (!mul || ldmul == 0 || ldmul >= N) && (!scale || ldmod == 0 || ldmod >= K))
        return qarig_decode_linear_f32(X, ldx, 0, eps, gamma, beta, scale, shift, ldmod, W, ldw, w_gs, bias, bias_gs,
                                       nullptr, 0, mul, ldmul, C, ldc, c_gs, groups, M, N, K, act, stream);
    GemmEpilogue eps_{C, ldc, bias, nullptr, 0, nullptr, 0, act, nullptr, 0, 0, nullptr};
    SkinnyFuse fz{gamma, beta, scale, shift, ldmod, eps, mul, ldmul};
    launch_skinny(X, ldx, W, ldw, eps_, M, N, K, 0, w_gs, c_gs, bias_gs, groups, (hipStream_t)stream, fz);
    QARIG_CHECK_LAUNCH("gemm_skinny_ln");
    return QARIG_OK;
}
