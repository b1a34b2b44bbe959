// BMU (best-matching-unit) search: Codebook.get_patches_bmu of the reference
// (models/Codebook.py:77-99) = patchify (models/layers.py:8-34) -> torch.cdist
// (matmul form for >25 rows) -> argmin(first index) -> int64.
//
// One fused kernel: the latent is gathered in patch order while staging (no
// patchified copy in HBM), the -2 x.w contraction runs on v_mfma_f32_32x32x2_f32
// with the accumulator pre-loaded with |w|^2, and each lane keeps a running
// (sqrt-distance, first index) so the (rows x K) distance matrix never exists.
//
// Arithmetic definition (what oracle/bmu_oracle.c restates; all fp32):
//   w2[k] = fma-chain_e  w[k][e]^2            (e ascending, from 0)
//   x2[r] = fma-chain_e  x[r][e]^2
//   acc   = 0; for e ascending: acc = fmaf(-2*w[k][e], x[r][e], acc)     (the MFMA chain)
//   d     = sqrtf(max((acc + w2[k]) + x2[r], 0));  index = first k with minimal d.
// Both norm chains are computed from the operand tiles while they sit in LDS for the
// MFMAs (a hook of the contraction loop): the codebook and the latent are read once.
// The scan itself is branch-free on t = acc + w2[k] (x2[r] is the same for every candidate
// of a row): each lane keeps (min t, its first index, second-smallest t) at 5 VALU ops per
// candidate (add, v_med3, cmp, cndmask, min).  d = f(t) = sqrtf(max(t + x2, 0)) is monotone
// non-decreasing, so argmin-first over d equals argmin-first over t unless another candidate
// maps to the same d as the minimum (the add's rounding, the clamp or sqrt collapsing
// neighbouring floats: ~1 row in 65k).  Exactly those rows -- f(second) == f(min) -- are
// re-scanned by the whole block with the literal definition above, so the result is
// bit-identical to it on every input.
#include <limits.h>
#include <type_traits>

#include "qarig_common.h"

namespace qarig {

struct PatchGeom {
    const float* x;
    int N, C, H, W, pH, pW, gh, gw;  // gh x gw patch grid
    int D;                           // C*pH*pW
    int R;                           // N*gh*gw patch rows
};

__device__ __forceinline__ int64_t patch_row_base(const PatchGeom& g, int row) {
    const int per = g.gh * g.gw;
    const int n = row / per;
    const int rem = row - n * per;
    const int ph = rem / g.gw, pw = rem - ph * g.gw;
    return ((int64_t)n * g.C * g.H + (int64_t)ph * g.pH) * g.W + (int64_t)pw * g.pW;
}

// B-side loader: x index = patch row, k index = element (c, i, j) of the patch,
// channel-major then row then column, exactly patchify's order.
struct SrcPatch {
    PatchGeom g;
    int64_t rowbase;
    bool valid;

    __device__ __forceinline__ void init(int x0, int tid) {
        const int row = x0 + (tid & 127);
        valid = row < g.R;
        rowbase = valid ? patch_row_base(g, row) : 0;
    }
    __device__ __forceinline__ bool interior(int, int, int) const { return false; }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        load(r, x0, k0, tid);
    }
    __device__ __forceinline__ void load(float (&r)[STAGE], int /*x0*/, int k0, int tid) const {
        int e = k0 + (tid >> 7) * 8;
        int j = e % g.pW;
        int t = e / g.pW;
        int i = t % g.pH;
        int c = t / g.pH;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            r[q] = (valid && e + q < g.D) ? g.x[rowbase + ((int64_t)c * g.H + i) * g.W + j] : 0.0f;
            if (++j == g.pW) {
                j = 0;
                if (++i == g.pH) { i = 0; ++c; }
            }
        }
    }
    __device__ __forceinline__ void store(const float (&r)[STAGE], float* T, int tid) const {
        const int x = tid & 127;
        const int k = (tid >> 7) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) T[(k + q) * LDT + x] = r[q];
    }
};

// |w|^2 (threads 0..127: one code each) and |x|^2 (threads 128..255: one patch row each)
// as sequential fma chains over the staged tiles: TA holds -2*w, TB holds x.
struct NormHook {
    float acc;
    int tid;
    __device__ __forceinline__ void operator()(const float* ta, const float* tb) {
        if (tid < 128) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) {
                const float wv = ta[kk * LDT + tid] * -0.5f;   // exact
                acc = fmaf(wv, wv, acc);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) {
                const float xv = tb[kk * LDT + tid - 128];
                acc = fmaf(xv, xv, acc);
            }
        }
    }
};

struct BmuState {
    float d2;   // smallest t = acc + |w|^2
    int idx;    // its first index
    float sec;  // smallest t among all OTHER candidates
};

__device__ __forceinline__ float bmu_dist(float t, float x2) { return sqrtf(fmaxf(t + x2, 0.0f)); }

// One candidate of the branch-free scan.
__device__ __forceinline__ void bmu_scan(float t, int code, float& best, int& idx, float& sec) {
    sec = __builtin_amdgcn_fmed3f(best, t, sec);   // sec >= best always: the median is the new second
    idx = t < best ? code : idx;
    best = fminf(best, t);
}

__device__ __forceinline__ BmuState bmu_merge(BmuState a, BmuState b) {
    if (b.d2 < a.d2 || (b.d2 == a.d2 && b.idx < a.idx)) {
        const BmuState t = a; a = b; b = t;
    }
    a.sec = fminf(a.sec, b.d2);   // b.sec >= b.d2
    return a;
}

// Another candidate maps to the minimum's distance: the first-index rule over d must decide.
__device__ __forceinline__ int bmu_needs_exact(const BmuState& s, float x2) {
    return s.sec < INFINITY && bmu_dist(s.sec, x2) == bmu_dist(s.d2, x2);
}

// Literal re-scan of the flagged rows of a 128-row block, all 256 threads per row:
// thread t takes codes t, t+256, ... (ascending), the block reduces (d, index) with the
// first-index rule.  Same fp32 chains as the MFMA path and the oracle.
__device__ __forceinline__ void bmu_exact_rows(const PatchGeom& g, const float* __restrict__ w, int K, int p0,
                               const int* flags, float* lds, int64_t* __restrict__ out) {
    float* rs = lds;                                  // [256]
    int* ri = reinterpret_cast<int*>(lds + 256);      // [256]
    const int tid = threadIdx.x;
    // flags[128] holds the two 64-bit ballots of the block's rows (block-uniform reads)
    const unsigned long long* masks = reinterpret_cast<const unsigned long long*>(flags);
    for (int half = 0; half < 2; ++half)
    for (unsigned long long m = masks[half]; m; m &= m - 1) {
        const int r = half * 64 + __ffsll((long long)m) - 1;
        const int row = p0 + r;
        const int64_t base = patch_row_base(g, row);
        float x2 = 0.0f;
        for (int c = 0; c < g.C; ++c)
            for (int i = 0; i < g.pH; ++i) {
                const float* p = g.x + base + ((int64_t)c * g.H + i) * g.W;
                for (int j = 0; j < g.pW; ++j) x2 = fmaf(p[j], p[j], x2);
            }
        float best = INFINITY;
        int bidx = INT_MAX;
        for (int k = tid; k < K; k += 256) {
            const float* wk = w + (int64_t)k * g.D;
            float w2 = 0.0f;
            for (int e = 0; e < g.D; ++e) w2 = fmaf(wk[e], wk[e], w2);
            float acc = 0.0f;
            int e = 0;
            for (int c = 0; c < g.C; ++c)
                for (int i = 0; i < g.pH; ++i) {
                    const float* p = g.x + base + ((int64_t)c * g.H + i) * g.W;
                    for (int j = 0; j < g.pW; ++j, ++e) acc = fmaf(-2.0f * wk[e], p[j], acc);
                }
            const float d = sqrtf(fmaxf((acc + w2) + x2, 0.0f));
            if (d < best) { best = d; bidx = k; }
        }
        __syncthreads();
        rs[tid] = best;
        ri[tid] = bidx;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const float s2 = rs[tid + o];
                const int i2 = ri[tid + o];
                if (s2 < rs[tid] || (s2 == rs[tid] && i2 < ri[tid])) { rs[tid] = s2; ri[tid] = i2; }
            }
            __syncthreads();
        }
        if (tid == 0) out[row] = ri[0] == INT_MAX ? 0 : (int64_t)ri[0];
    }
}

// Block epilogue shared by the BMU kernels: combine the 4 holders of each patch column
// (lane halves x waves along the code axis), write the index (or the per-split partial
// state), then re-scan exactly the rows whose minimum shares its sqrt with another candidate.
// `lds` needs 13*128 + 512 floats and must no longer be in use as operand tiles.
__device__ __forceinline__ void bmu_block_finish(const PatchGeom& g, const float* __restrict__ w,
                                                 int K, int p0, const float (&best_d2)[2],
                                                 const int (&best_i)[2], const float (&sec_d2)[2],
                                                 float* lds, const float* x2rows,
                                                 float* __restrict__ part_d,
                                                 int* __restrict__ part_i,
                                                 float* __restrict__ part_s,
                                                 float* __restrict__ part_x2,
                                                 int64_t* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, cl = lane & 31;
    // Combine the 4 holders of each patch column: lane halves (h) x waves (wm).
    float* cd = lds;                                   // [4][128] min d2
    int* ci = reinterpret_cast<int*>(lds + 4 * 128);   // [4][128] its first index
    float* cs = lds + 8 * 128;                         // [4][128] second-smallest d2
    const int slot = wm * 2 + (lane >> 5);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + cl;
        cd[slot * 128 + col] = best_d2[j];
        ci[slot * 128 + col] = best_i[j];
        cs[slot * 128 + col] = sec_d2[j];
    }
    __syncthreads();
    int* flags = reinterpret_cast<int*>(lds + 12 * 128);   // [128] rows needing the exact scan
    if (tid < 128) {
        const int prow = p0 + tid;
        int flag = 0;
        if (prow < g.R) {
            BmuState st{cd[tid], ci[tid], cs[tid]};
#pragma unroll
            for (int q = 1; q < 4; ++q) st = bmu_merge(st, BmuState{cd[q * 128 + tid], ci[q * 128 + tid],
                                                                   cs[q * 128 + tid]});
            if (out) {
                flag = bmu_needs_exact(st, x2rows[tid]);
                out[prow] = st.idx == INT_MAX ? 0 : (int64_t)st.idx;
            } else {
                const int64_t o = (int64_t)blockIdx.y * g.R + prow;
                part_d[o] = st.d2;
                part_i[o] = st.idx;
                part_s[o] = st.sec;
                if (blockIdx.y == 0) part_x2[prow] = x2rows[tid];
            }
        }
        const unsigned long long m = __ballot(flag);
        if (lane == 0) reinterpret_cast<unsigned long long*>(flags)[wave] = m;
    }
    __syncthreads();
    if (out) bmu_exact_rows(g, w, K, p0, flags, lds, out);
}

__global__ __launch_bounds__(NTHREADS, 2) void bmu_mma_kernel(PatchGeom g,
                                                              const float* __restrict__ w, int K,
                                                              int tiles_per_split,
                                                              float* __restrict__ part_d,
                                                              int* __restrict__ part_i,
                                                              float* __restrict__ part_s,
                                                              float* __restrict__ part_x2,
                                                              int64_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[GEMM_LDS_FLOATS];
    __shared__ float norms[256];   // [0,128): |w|^2 of the code tile, [128,256): |x|^2 of the rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1;
    const int p0 = blockIdx.x * BN;  // first patch row of this block

    SrcKContig sa{w, (int64_t)g.D, K, g.D, -2.0f,
                  (((uintptr_t)w & 15) == 0) && (g.D % 4 == 0)};
    SrcPatch sb;
    sb.g = g;
    sb.init(p0, tid);

    float best_d2[2] = {INFINITY, INFINITY};
    float sec_d2[2] = {INFINITY, INFINITY};
    int best_i[2] = {INT_MAX, INT_MAX};

    const int code_tiles = (K + BM - 1) / BM;
    const int ct0 = blockIdx.y * tiles_per_split;
    const int ct1 = min(code_tiles, ct0 + tiles_per_split);
    for (int ct = ct0; ct < ct1; ++ct) {
        const int c0 = ct * BM;
        Acc acc;
        acc_zero(acc);
        NormHook hook{0.0f, tid};
        contract_loop<false>(acc, sa, sb, c0, p0, 0, g.D, lds, hook);
        norms[tid] = hook.acc;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lc = wm * 64 + i * 32 + acc_row(r, lane);
                const int code = c0 + lc;
                const float w2 = code < K ? norms[lc] : INFINITY;   // padding codes never win
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bmu_scan(acc.t[i][j][r] + w2, code, best_d2[j], best_i[j], sec_d2[j]);
            }
        __syncthreads();   // norms[0..127] is rewritten by the next code tile
    }

    bmu_block_finish(g, w, K, p0, best_d2, best_i, sec_d2, lds, norms + 128, part_d, part_i, part_s,
                     part_x2, out);
}

// Small patch widths (D <= 16*NKT <= 64: every hierarchical patch size of the reference's
// cascade except "whole latent").  The block's 128 patch rows are gathered into LDS ONCE
// (with their |x|^2 chains); code tiles stream through a double-buffered LDS slot whose
// next global loads are in flight under the current tile's MFMAs; the branch-free
// (min, first index, second) scan follows each tile.  Nothing is re-read from HBM.
template <int NKT, int KS>
__global__ __launch_bounds__(NTHREADS, 2) void bmu_small_kernel(PatchGeom g,
                                                                const float* __restrict__ w, int K,
                                                                int tiles_per_split,
                                                                float* __restrict__ part_d,
                                                                int* __restrict__ part_i,
                                                                float* __restrict__ part_s,
                                                                float* __restrict__ part_x2,
                                                                int64_t* __restrict__ out) {
    // KS = 2-deep MFMA steps that carry data in the last k-tile (D <= 4 -> 2 ... D > 8 -> 8)
    // TB[NKT] | TA[2][NKT]; the combine / exact-rescan scratch reuses TA after the loop
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float norms[256];
    float* TB = lds;
    float* TA = lds + NKT * TILE_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int p0 = blockIdx.x * BN;

    SrcKContig sa{w, (int64_t)g.D, K, g.D, -2.0f,
                  (((uintptr_t)w & 15) == 0) && (g.D % 4 == 0)};
    SrcPatch sb;
    sb.g = g;
    sb.init(p0, tid);

    const int code_tiles = (K + BM - 1) / BM;
    const int ct0 = blockIdx.y * tiles_per_split;
    const int ct1 = min(code_tiles, ct0 + tiles_per_split);

    float ra[NKT][STAGE];
    // patches: staged once
    {
        float rb[STAGE];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            sb.load(rb, p0, kt * BK, tid);
            sb.store(rb, TB + kt * TILE_FLOATS, tid);
        }
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) sa.load(ra[kt], ct0 * BM, kt * BK, tid);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) sa.store(ra[kt], TA + kt * TILE_FLOATS, tid);
    __syncthreads();
    if (tid >= 128) {   // |x|^2 chains, e ascending
        float acc = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) {
                const float xv = TB[kt * TILE_FLOATS + kk * LDT + tid - 128];
                acc = fmaf(xv, xv, acc);
            }
        norms[tid] = acc;
    }

    float best_d2[2] = {INFINITY, INFINITY};
    float sec_d2[2] = {INFINITY, INFINITY};
    int best_i[2] = {INT_MAX, INT_MAX};

    for (int ct = ct0; ct < ct1; ++ct) {
        const int c0 = ct * BM;
        float* ta = TA + ((ct - ct0) & 1) * NKT * TILE_FLOATS;
        float* na = TA + (((ct - ct0) & 1) ^ 1) * NKT * TILE_FLOATS;
        const bool more = ct + 1 < ct1;
        if (more) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) sa.load(ra[kt], c0 + BM, kt * BK, tid);
        }
        if (tid < 128) {   // |w|^2 chain of this code tile (TA holds -2w)
            float acc = 0.0f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) {
                    const float wv = ta[kt * TILE_FLOATS + kk * LDT + tid] * -0.5f;
                    acc = fmaf(wv, wv, acc);
                }
            norms[tid] = acc;
        }
        Acc acc;
        acc_zero(acc);
#pragma unroll
        for (int kt = 0; kt < NKT - 1; ++kt)
            mma_tile<BK / 2>(acc, ta + kt * TILE_FLOATS, TB + kt * TILE_FLOATS, wm, wn, lane);
        mma_tile<KS>(acc, ta + (NKT - 1) * TILE_FLOATS, TB + (NKT - 1) * TILE_FLOATS, wm, wn, lane);
        if (more) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) sa.store(ra[kt], na + kt * TILE_FLOATS, tid);
        }
        __syncthreads();   // norms[] visible; next code tile staged
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lc = wm * 64 + i * 32 + acc_row(r, lane);
                const int code = c0 + lc;
                const float w2 = code < K ? norms[lc] : INFINITY;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bmu_scan(acc.t[i][j][r] + w2, code, best_d2[j], best_i[j], sec_d2[j]);
            }
        __syncthreads();   // norms[0..127] is rewritten by the next tile
    }
    bmu_block_finish(g, w, K, p0, best_d2, best_i, sec_d2, TA, norms + 128, part_d, part_i, part_s,
                     part_x2, out);
}

// ---------------------------------------------------------------------------------
// Narrow patches (D <= 16: the HR stages of the cascade), codebook resident in LDS.
// The block stages its code range ONCE as MFMA A-fragments (-2w split into even / odd
// elements, |w|^2 beside them), every wave keeps its 32 patch rows as B-fragments in
// registers for the whole launch, and the main loop has no barrier: per 32-code tile
// 2 x ds_read_b128 + KS+1 MFMAs + a 4-op scan per candidate.  |w|^2 enters through one
// extra MFMA step (A = |w|^2, B = 1): the chain's last operation is then
// fma(|w|^2, 1, acc) = round(acc + |w|^2), the `acc + w2` of the definition above.
// CS waves share a row tile and split the block's codes among them (launches with few rows
// still fill the chip: 32 * 4/CS rows per block).  grid.y = code chunks when the codebook
// exceeds the LDS budget; their partial states go through bmu_finalize_kernel.
// The resident kernel's inner loop is written as a sequence of volatile asm statements, which
// the compiler keeps in source order: one MFMA of the NEXT tile pair, then the scan of two
// candidates of the CURRENT pair (8 VALU ops, in the shadow of that MFMA's 64 cycles), and so
// on.  Left to the compiler the loop becomes 18 MFMAs followed by 128 VALU ops, and the
// co-resident waves -- which run in lock-step -- then queue for the matrix pipe together and
// for the vector ALU together.
// Scan of one candidate: 4 VALU ops (v_med3, v_cmp, v_cndmask, v_min) with the candidate's
// position inside the pair as an inline constant; the compiler's own fminf() canonicalises
// both operands first, and v_cndmask cannot take an SGPR position next to VCC on gfx9.
template <int R>
__device__ __forceinline__ void res_scan2(float t0, float t1, float& best, int& rnew, float& sec) {
    static_assert(R >= 0 && R + 1 <= 64, "positions must stay inline constants");
    asm volatile(
        "v_med3_f32 %2, %0, %3, %2\n"
        "v_cmp_nlt_f32 vcc, %3, %0\n"
        "v_cndmask_b32 %1, %5, %1, vcc\n"
        "v_min_f32 %0, %0, %3\n"
        "v_med3_f32 %2, %0, %4, %2\n"
        "v_cmp_nlt_f32 vcc, %4, %0\n"
        "v_cndmask_b32 %1, %6, %1, vcc\n"
        "v_min_f32 %0, %0, %4\n"
        : "+v"(best), "+v"(rnew), "+v"(sec)
        : "v"(t0), "v"(t1), "n"(R), "n"(R + 1)
        : "vcc");
}
// Group form of the scan (GROUPS launches: code chunks of >= 8 tile pairs per wave): the minimum of 8
// candidates by v_min3 (4 ops), then ONE 4-op scan step for the group -- 1 VALU op per candidate
// instead of 4.  What the scan tracks is then (smallest group minimum, its group, second smallest
// group minimum); the winning group's 8 candidates are re-evaluated exactly after the loop (the
// same fma chain on the vector ALU gives the MFMA's bits), which recovers the index inside the
// group and the second smallest value inside it.
template <int R>
__device__ __forceinline__ void res_scan_grp(float a0, float a1, float a2, float a3, float a4, float a5,
                                             float a6, float a7, float& best, int& rnew, float& sec) {
    static_assert(R >= 0 && R <= 64, "group positions must stay inline constants");
    float tmp;
    asm volatile(
        "v_min3_f32 %3, %4, %5, %6\n"
        "v_min3_f32 %3, %3, %7, %8\n"
        "v_min3_f32 %3, %3, %9, %10\n"
        "v_min_f32 %3, %3, %11\n"
        "v_med3_f32 %2, %0, %3, %2\n"
        "v_cmp_nlt_f32 vcc, %3, %0\n"
        "v_cndmask_b32 %1, %12, %1, vcc\n"
        "v_min_f32 %0, %0, %3\n"
        : "+v"(best), "+v"(rnew), "+v"(sec), "=&v"(tmp)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "n"(R)
        : "vcc");
}
template <bool FIRST>
__device__ __forceinline__ void res_mfma(f32x16& acc, float a, float b) {
    // s_nop 1: an MFMA may read a VGPR no sooner than 2 wait states after a vector-ALU write
    // of it; the compiler inserts those for its own MFMAs, not for one inside inline asm (the
    // |w|^2 operand is a v_cndmask result the scheduler likes to place right in front)
    if constexpr (FIRST)
        asm volatile("s_nop 1\nv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else
        asm volatile("s_nop 1\nv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct PatchOffsets {
    int off[16];   // element e of a patch, relative to the patch's first element
    // floor(n / d) = (n * m) >> sh for 0 <= n < 2^31 (Granlund-Montgomery: m = floor(2^sh / d) + 1,
    // sh = 31 + ceil(log2 d)): the two divisions of patch_row_base without ~35-op integer divides
    unsigned m_per, m_gw;
    int sh_per, sh_gw;
};
__device__ __forceinline__ int fast_div(int n, unsigned m, int sh) {
    return (int)(((unsigned long long)(unsigned)n * m) >> sh);
}
__device__ __forceinline__ int64_t patch_row_base_fast(const PatchGeom& g, const PatchOffsets& po, int row) {
    const int per = g.gh * g.gw;
    const int n = fast_div(row, po.m_per, po.sh_per);
    const int rem = row - n * per;
    const int ph = fast_div(rem, po.m_gw, po.sh_gw), pw = rem - ph * g.gw;
    return ((int64_t)n * g.C * g.H + (int64_t)ph * g.pH) * g.W + (int64_t)pw * g.pW;
}

constexpr int RES_MAX_LDS = 59 * 1024;   // staged codes; + 4.5 KB of combine scratch < 64 KB

template <int KS>
__device__ __forceinline__ void res_frag(const float* AE, int c, int hi, int chunk, float (&a)[KS]) {
    // half `hi` of code c: KS floats, 16-B chunks swapped on alternate groups of 8 codes so that
    // the 16 lanes of a b128 phase cover all banks
    const float* p = AE + ((size_t)hi * chunk + c) * KS;
    if constexpr (KS == 8) {
        const int sw = (c >> 3) & 1;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p + 4 * sw);
        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(p + 4 * (sw ^ 1));
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = lo[q]; a[4 + q] = hi4[q]; }
    } else if constexpr (KS == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = v[q];
    } else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 v = *reinterpret_cast<const f32x2*>(p);
        a[0] = v[0]; a[1] = v[1];
    }
}

// Literal re-scan of the flagged rows (bmu_exact_rows) for a single-chunk launch of the
// resident kernel: the codes are read back from their LDS image (-2w, so the product chain
// takes them as they are; w = -0.5 * (-2w) is exact) and the patch row is fetched with all
// its loads in flight at once -- a flagged row costs about one memory round trip instead of
// ~50 dependent ones, which matters because the launch ends with its slowest workgroup.
template <int KS>
__device__ __forceinline__ void res_exact_rows(const PatchGeom& g, const PatchOffsets& po, const float* AE,
                                            const float* W2, int chunk, int K, int p0, const int* flags,
                                            float* lds, int64_t* __restrict__ out) {
    float* rs = lds;                                  // [4] per-wave minima
    int* ri = reinterpret_cast<int*>(lds + 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long* masks = reinterpret_cast<const unsigned long long*>(flags);
    for (int half = 0; half < 2; ++half)
    for (unsigned long long m = masks[half]; m; m &= m - 1) {
        const int r = half * 64 + __ffsll((long long)m) - 1;
        const int row = p0 + r;
        const float* px = g.x + patch_row_base(g, row);
        float xs[2 * KS];
#pragma unroll
        for (int e = 0; e < 2 * KS; ++e) {
            const float v = px[po.off[e]];
            xs[e] = e < g.D ? v : 0.0f;
        }
        float x2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 2 * KS; ++e) x2 = fmaf(xs[e], xs[e], x2);
        float best = INFINITY;
        int bidx = INT_MAX;
        for (int k = tid; k < K; k += NTHREADS) {      // K <= chunk: one chunk
            float ev[KS], od[KS];
            res_frag<KS>(AE, k, 0, chunk, ev);
            res_frag<KS>(AE, k, 1, chunk, od);
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                acc = fmaf(ev[q], xs[2 * q], acc);
                acc = fmaf(od[q], xs[2 * q + 1], acc);
            }
            const float d = sqrtf(fmaxf((acc + W2[k]) + x2, 0.0f));
            if (d < best) { best = d; bidx = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float s2 = __shfl_xor(best, o);
            const int i2 = __shfl_xor(bidx, o);
            if (s2 < best || (s2 == best && i2 < bidx)) { best = s2; bidx = i2; }
        }
        __syncthreads();
        if (lane == 0) { rs[wave] = best; ri[wave] = bidx; }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (rs[q] < best || (rs[q] == best && ri[q] < bidx)) { best = rs[q]; bidx = ri[q]; }
            out[row] = bidx == INT_MAX ? 0 : (int64_t)bidx;
        }
    }
}

template <int KS, int CS, bool GROUPS, int NT>
__global__ __launch_bounds__(NTHREADS, 2) void bmu_resident_kernel(PatchGeom g, PatchOffsets po,
                                                                   const float* __restrict__ w, int K,
                                                                   int chunk,
                                                                   float* __restrict__ part_d,
                                                                   int* __restrict__ part_i,
                                                                   float* __restrict__ part_s,
                                                                   float* __restrict__ part_x2,
                                                                   int64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RPB = 32 * (4 / CS);          // patch rows per block
    float* AE = lds;                            // [2][chunk][KS]: even, then odd elements of -2w
    float* W2 = lds + (size_t)2 * chunk * KS;   // [chunk]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, cl = lane & 31;
    const int cbase = blockIdx.y * chunk;
    const int p0 = blockIdx.x * RPB;

    // ---- global loads first, all in flight together: this lane's patch row, then two codes
    // per thread per staging pass (a launch of a few hundred workgroups is a chain of memory
    // round trips before anything else: every dependent trip removed is ~0.7 us of its ~6)
    const int rtile = wave / CS, cpart = wave % CS;
    const int row = p0 + rtile * 32 + cl;
    float xv[2 * KS];
    {   // rows past the end read row 0 and are never written; po.off[e >= D] = off[0]
        const float* px = g.x + patch_row_base_fast(g, po, row < g.R ? row : 0);
#pragma unroll
        for (int e = 0; e < 2 * KS; ++e) xv[e] = px[po.off[e]];
    }
    const bool vec = (g.D % 4 == 0) && (((uintptr_t)w & 15) == 0);
    for (int c = tid; c < chunk; c += 2 * NTHREADS) {
        float we[2][2 * KS];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int code = cbase + c + u * NTHREADS;
            const float* wk = w + (int64_t)(code < K ? code : 0) * g.D;   // clamped: selected away below
            if (vec) {
#pragma unroll
                for (int q = 0; q < KS / 2; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wk + (4 * q < g.D ? 4 * q : 0));
#pragma unroll
                    for (int t = 0; t < 4; ++t) we[u][4 * q + t] = v[t];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 2 * KS; ++e) we[u][e] = wk[e < g.D ? e : 0];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = c + u * NTHREADS;
            const bool live = cbase + cc < K;
#pragma unroll
            for (int e = 0; e < 2 * KS; ++e) we[u][e] = (live && e < g.D) ? we[u][e] : 0.0f;
            float w2 = 0.0f;
#pragma unroll
            for (int e = 0; e < 2 * KS; ++e) w2 = fmaf(we[u][e], we[u][e], w2);
            if (cc < chunk) {
                float* pe = AE + (size_t)cc * KS;
                float* po_ = AE + ((size_t)chunk + cc) * KS;
                if constexpr (KS == 8) {
                    const int sw = (cc >> 3) & 1;
                    *reinterpret_cast<f32x4*>(pe + 4 * sw) = f32x4{-2.0f * we[u][0], -2.0f * we[u][2], -2.0f * we[u][4], -2.0f * we[u][6]};
                    *reinterpret_cast<f32x4*>(pe + 4 * (sw ^ 1)) = f32x4{-2.0f * we[u][8], -2.0f * we[u][10], -2.0f * we[u][12], -2.0f * we[u][14]};
                    *reinterpret_cast<f32x4*>(po_ + 4 * sw) = f32x4{-2.0f * we[u][1], -2.0f * we[u][3], -2.0f * we[u][5], -2.0f * we[u][7]};
                    *reinterpret_cast<f32x4*>(po_ + 4 * (sw ^ 1)) = f32x4{-2.0f * we[u][9], -2.0f * we[u][11], -2.0f * we[u][13], -2.0f * we[u][15]};
                } else {
#pragma unroll
                    for (int q = 0; q < KS; ++q) { pe[q] = -2.0f * we[u][2 * q]; po_[q] = -2.0f * we[u][2 * q + 1]; }
                }
                W2[cc] = live ? w2 : INFINITY;
            }
        }
    }

    // ---- the patch row: |x|^2 chain over its elements in order, then the half this lane feeds
    // to the MFMA (k = 2 s + hi).  (A select between two slots of one array would become an
    // indexed access, i.e. a trip through scratch memory: pairs of scalars instead.)
    float x2 = 0.0f;
    float b[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const float e0 = 2 * s < g.D ? xv[2 * s] : 0.0f, e1 = 2 * s + 1 < g.D ? xv[2 * s + 1] : 0.0f;
        x2 = fmaf(e0, e0, x2);
        x2 = fmaf(e1, e1, x2);
        b[s] = hi ? e1 : e0;
    }
    const float b_ext = hi ? 0.0f : 1.0f;
    __syncthreads();

    // ---- main loop: this wave's code tiles, two at a time (independent accumulators); the
    // MFMAs of the next pair are issued before the current pair is scanned, so the scan's VALU
    // work runs under them
    const int tiles_w = chunk / (32 * CS);            // a multiple of NT (host)
    const int t0 = cpart * tiles_w;
    float best = INFINITY, sec = INFINITY;
    int seq_best = INT_MAX;
    // step<ISSUE, SCAN>: the NT (KS + 1) MFMAs of the NT tiles starting at tile tpn into n[], NT
    // independent accumulate chains taken round robin (with NT = 2 every MFMA waits for the one but
    // last: measured 15 % of the launch), interleaved with the scan of the NT tiles starting at tpc
    // held in c[].  Candidates ascend within the lane: seq = 16 * tile + r (2 * tile + half for groups).
    struct Tiles { f32x16 t[NT]; };
    struct Frags { float af[NT][KS]; float wf[NT]; };     // A fragments + |w|^2 of NT code tiles
    auto fetch = [&](int tp, Frags& f) {
        const int cc = (t0 + tp) * 32 + cl;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            res_frag<KS>(AE, cc + 32 * u, hi, chunk, f.af[u]);
            const float wv = W2[cc + 32 * u];
            f.wf[u] = hi ? 0.0f : wv;
        }
    };
    // fn: the fragments of the tiles being issued (fetched one step earlier, so their LDS latency
    // is behind a step of MFMAs); ff: receives the fragments of the tiles at tpf (< 0: none)
    auto step = [&](auto issue_c, auto scan_c, int tpn, Tiles& n, Frags& fn, int tpf, Frags& ff, int tpc, Tiles& c) {
        constexpr bool ISSUE = decltype(issue_c)::value, SCAN = decltype(scan_c)::value;
        constexpr int SLOTS = NT * (KS + 1);
        int rnew = -1;
        static_for<0, SLOTS>([&](auto mc) {
            constexpr int M = decltype(mc)::value;
            if constexpr (ISSUE) {
                constexpr int S = M / NT, U = M % NT;
                if constexpr (S < KS) res_mfma<S == 0>(n.t[U], fn.af[U][S], b[S]);
                else res_mfma<false>(n.t[U], fn.wf[U], b_ext);
                if constexpr (M == NT - 1)
                    if (tpf >= 0) fetch(tpf, ff);       // behind the first round of MFMAs
            }
            if constexpr (SCAN && !GROUPS) {
                // scan2 chunk j (8 per tile) follows MFMA floor(j * SLOTS / (8 NT))
                static_for<0, 8 * NT>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    if constexpr (J * SLOTS / (8 * NT) == M)
                        res_scan2<2 * J>(c.t[J / 8][2 * (J % 8)], c.t[J / 8][2 * (J % 8) + 1], best, rnew, sec);
                });
            }
            if constexpr (SCAN && GROUPS) {
                // group j (2 per tile: halves of an accumulator tile, ascending codes within the
                // lane) follows MFMA floor(j * SLOTS / (2 NT))
                static_for<0, 2 * NT>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    if constexpr (J * SLOTS / (2 * NT) == M) {
                        const f32x16& ct = c.t[J / 2];
                        constexpr int o = 8 * (J & 1);
                        res_scan_grp<J>(ct[o], ct[o + 1], ct[o + 2], ct[o + 3], ct[o + 4], ct[o + 5], ct[o + 6],
                                        ct[o + 7], best, rnew, sec);
                    }
                });
            }
        });
        // The MFMAs sit in inline asm, so the compiler inserts none of the wait states it owes
        // between a matrix-core write and a vector-ALU read of the same registers -- and it
        // does read them: register copies of an accumulator at the loop edges.  Every issuing
        // step therefore ends with the 16-pass latency in nops (24 cycles per NT (KS + 1) MFMAs).
        if constexpr (ISSUE) {
            if constexpr (NT == 2) asm volatile("s_nop 15\ns_nop 7" : "+v"(n.t[0]), "+v"(n.t[1]));
            else asm volatile("s_nop 15\ns_nop 7" : "+v"(n.t[0]), "+v"(n.t[1]), "+v"(n.t[2]), "+v"(n.t[3]));
        }
        // position of the winner: candidate (16 per tile) or group (2 per tile) index within the wave
        if constexpr (SCAN) seq_best = rnew >= 0 ? tpc * (GROUPS ? 2 : 16) + rnew : seq_best;
    };
    {
        Tiles A, B;
        Frags FA, FB;
        constexpr std::true_type yes{};
        constexpr std::false_type no{};
        fetch(0, FA);
        step(yes, no, 0, A, FA, NT < tiles_w ? NT : -1, FB, 0, B);
        int tp = 0;
        while (true) {
            if (tp + NT < tiles_w) step(yes, yes, tp + NT, B, FB, tp + 2 * NT < tiles_w ? tp + 2 * NT : -1, FA, tp, A);
            else step(no, yes, 0, B, FB, -1, FA, tp, A);
            tp += NT;
            if (tp >= tiles_w) break;
            if (tp + NT < tiles_w) step(yes, yes, tp + NT, A, FA, tp + 2 * NT < tiles_w ? tp + 2 * NT : -1, FB, tp, B);
            else step(no, yes, 0, A, FA, -1, FB, tp, B);
            tp += NT;
            if (tp >= tiles_w) break;
        }
    }
    // seq -> code: tile t0 + seq/16, accumulator row acc_row(seq % 16, lane)
    BmuState st{best, INT_MAX, sec};
    if constexpr (!GROUPS) {
        if (seq_best != INT_MAX) {
            const int code = cbase + (t0 + (seq_best >> 4)) * 32 + acc_row(seq_best & 15, lane);
            st.idx = code < K ? code : INT_MAX;
        }
    } else if (seq_best != INT_MAX) {
        // the winning group, candidate by candidate: t = fma chain over the elements in order, then
        // + |w|^2 -- the bits the MFMA produced -- through the plain 5-op scan; its minimum is `best`
        // again, its first index is the lane's index, its second smallest joins `sec`
        const int tile = t0 + (seq_best >> 1), o = 8 * (seq_best & 1);
        float gb = INFINITY, gs = INFINITY;
        int gi = INT_MAX;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int cl_ = tile * 32 + acc_row(o + r, lane);       // code inside the chunk
            float ev[KS], od[KS];
            res_frag<KS>(AE, cl_, 0, chunk, ev);
            res_frag<KS>(AE, cl_, 1, chunk, od);
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                acc = fmaf(ev[q], 2 * q < g.D ? xv[2 * q] : 0.0f, acc);
                acc = fmaf(od[q], 2 * q + 1 < g.D ? xv[2 * q + 1] : 0.0f, acc);
            }
            const int code = cbase + cl_;
            bmu_scan(acc + W2[cl_], code < K ? code : INT_MAX, gb, gi, gs);
        }
        st.idx = gi;
        st.sec = fminf(sec, gs);
    }
    {   // the two lane halves hold disjoint codes of the same row
        BmuState o{__shfl_xor(st.d2, 32), __shfl_xor(st.idx, 32), __shfl_xor(st.sec, 32)};
        st = bmu_merge(st, o);
    }
    // combine scratch behind the staged codes (they stay readable for the exact re-scan)
    float* sc = W2 + chunk;
    float* cd = sc;                                    // [CS][RPB]
    int* ci = reinterpret_cast<int*>(sc + CS * RPB);
    float* cs = sc + 2 * CS * RPB;
    float* cx = sc + 3 * CS * RPB;                     // [RPB] |x|^2
    int* flags = reinterpret_cast<int*>(sc + 3 * CS * RPB + RPB);   // 128 ints (two 64-bit ballots)
    float* scratch = sc + 3 * CS * RPB + RPB + 128;    // 512 floats for the exact re-scan
    if (hi == 0) {
        const int col = rtile * 32 + cl;
        cd[cpart * RPB + col] = st.d2;
        ci[cpart * RPB + col] = st.idx;
        cs[cpart * RPB + col] = st.sec;
        if (cpart == 0) cx[col] = x2;
    }
    __syncthreads();
    if (tid < 128) {
        const int prow = p0 + tid;
        int flag = 0;
        if (tid < RPB && prow < g.R) {
            BmuState m{cd[tid], ci[tid], cs[tid]};
#pragma unroll
            for (int q = 1; q < CS; ++q) m = bmu_merge(m, BmuState{cd[q * RPB + tid], ci[q * RPB + tid], cs[q * RPB + tid]});
            if (out) {
                flag = bmu_needs_exact(m, cx[tid]);
                out[prow] = m.idx == INT_MAX ? 0 : (int64_t)m.idx;
            } else {
                const int64_t o = (int64_t)blockIdx.y * g.R + prow;
                part_d[o] = m.d2;
                part_i[o] = m.idx;
                part_s[o] = m.sec;
                if (blockIdx.y == 0) part_x2[prow] = cx[tid];
            }
        }
        const unsigned long long mk = __ballot(flag);
        if (lane == 0) reinterpret_cast<unsigned long long*>(flags)[wave] = mk;
    }
    __syncthreads();
    if (out) res_exact_rows<KS>(g, po, AE, W2, chunk, K, p0, flags, scratch, out);
}

// Merge of the per-split partial states (splits cover ascending code ranges), then the
// exact re-scan of flagged rows.  One block per 128 rows.
__global__ __launch_bounds__(256) void bmu_finalize_kernel(PatchGeom g, const float* __restrict__ w,
                                                           int K, const float* __restrict__ part_d,
                                                           const int* __restrict__ part_i,
                                                           const float* __restrict__ part_s,
                                                           const float* __restrict__ part_x2,
                                                           int nsplit, int64_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    __shared__ int flags[128];
    const int p0 = blockIdx.x * 128;
    const int tid = threadIdx.x;
    if (tid < 128) {
        const int row = p0 + tid;
        int flag = 0;
        if (row < g.R) {
            BmuState st{part_d[row], part_i[row], part_s[row]};
            for (int z = 1; z < nsplit; ++z) {
                const int64_t o = (int64_t)z * g.R + row;
                st = bmu_merge(st, BmuState{part_d[o], part_i[o], part_s[o]});
            }
            flag = bmu_needs_exact(st, part_x2[row]);
            out[row] = st.idx == INT_MAX ? 0 : (int64_t)st.idx;
        }
        const unsigned long long m = __ballot(flag);
        if ((tid & 63) == 0) reinterpret_cast<unsigned long long*>(flags)[tid >> 6] = m;
    }
    __syncthreads();
    bmu_exact_rows(g, w, K, p0, flags, lds, out);
}

// torch.cdist's small-input branch (both operands <= 25 rows): direct
// sqrt(sum (x-w)^2), sequential chain.
__global__ void bmu_direct_kernel(PatchGeom g, const float* __restrict__ w, int K,
                                  int64_t* __restrict__ out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= g.R) return;
    const int64_t base = patch_row_base(g, row);
    float best = INFINITY;
    int idx = 0;
    for (int k = 0; k < K; ++k) {
        const float* wk = w + (int64_t)k * g.D;
        float acc = 0.0f;
        int e = 0;
        for (int c = 0; c < g.C; ++c)
            for (int i = 0; i < g.pH; ++i)
                for (int j = 0; j < g.pW; ++j, ++e) {
                    const float d = g.x[base + ((int64_t)c * g.H + i) * g.W + j] - wk[e];
                    acc = fmaf(d, d, acc);
                }
        const float s = sqrtf(acc);
        if (s < best) { best = s; idx = k; }
    }
    out[row] = idx;
}


// ---------------------------------------------------------------------------------
// Few patch rows, long patches (the conditional codebook: one 4096-element patch per latent,
// 64 rows per batch).  The 128x128 MFMA tiling finds 4 workgroups there, each walking a
// 4096-deep reduction alone (0.59 ms).  The work is tiny (rows*K*D = 134 M fma), so it runs
// on the vector ALU instead, one (row, code) pair per lane with the SAME k-ordered chains as
// the MFMA form (an MFMA accumulates in k order with one rounding per fma, so
// fmaf(-2 w_e, x_e, acc) over ascending e is bit-identical).  acc (rows x K), w2 (K) and x2 (rows) go to
// the workspace; the second kernel takes sqrt(max((acc + w2) + x2, 0)) and the first minimum per row.
//
// Two kernels.  bmu_fewrows_dot_kernel (further down): any geometry; 8 x 8 pairs per 64-lane workgroup,
// operands through LDS in 128-element chunks, every lane carrying the product chain and its own |w|^2 and
// |x|^2 chains.  bmu_fewrows_roles_kernel (here): the aligned case -- the conditional codebook itself,
// whole latents as patches (pW == W, pH == H, D % 256 == 0, 16-B aligned x and codebook).  Counters of the
// general kernel at 64 x 512 x 4096 (111 us): its per-element guards make the compiler wait for each of a
// lane's 256 loads before it issues the next (SQ_WAIT_ANY 177 k of a wave's 315 k cycles), and a lone wave
// per SIMD issues the three chains' fmacs, their LDS reads and the patch index arithmetic one instruction
// at a time (110 k cycles of issue).  Here the loads are straight-line, one contiguous KB of one row per
// instruction (gathering 128 B of each of 8 rows per instruction: 60 us instead of 45), the next chunk's
// loads are in flight under the current chunk's chain, and the three chains are split over workgroup ROLES
// so that every wave carries one (5 k instead of 21 k vector instructions per wave): blockIdx.y < row
// tiles: the (row, code) products; next y: |w|^2 of code tile blockIdx.x; last y: |x|^2 of row tile
// blockIdx.x.  Same e-ascending fma chains: bit-identical values.
// Tile: RT_T x RT_T pairs per workgroup of RT_T * RT_T / 64 waves; wave v stages tile rows RT_NQ v ... +
// RT_NQ - 1 of each operand (a row's chunk is 256 elements = one 1-KB load instruction of the wave).
// 16 x 16 pairs read every row / code chunk once per 16 partners: half the L2 traffic of 8 x 8 at the same
// 45 us -- a wave still waits ~2.4 us per chunk whatever the tile or the chunk size.
constexpr int RT_T = 16;
constexpr int RT_THREADS = RT_T * RT_T;
constexpr int RT_CH = 256;
constexpr int RT_LD = RT_CH + 4;
constexpr int RT_NQ = RT_T / (RT_THREADS / 64);        // tile rows staged per wave
template <int ROLE>
__device__ __forceinline__ void fewrows_role(const PatchGeom& g, const float* __restrict__ w, int K, int code0,
                                             int row0, float (*xs)[RT_LD], float (*ws)[RT_LD],
                                             float* __restrict__ acc_out, float* __restrict__ w2_out,
                                             float* __restrict__ x2_out) {
    constexpr bool NX = ROLE != 1, NW = ROLE != 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = tid % RT_T, r = tid / RT_T;
    // load q of a chunk: tile row RT_NQ * wave + q (x row row0 + .. / code code0 + ..), elements 4 * lane ... + 3:
    // 1 KB contiguous per instruction (patches are whole latents here: element e of a row lies at x_row + e).
    // Rows / codes past the end read row / code 0 instead; their results are never stored.
    const float* wq[RT_NQ];
    const float* xq[RT_NQ];
#pragma unroll
    for (int q = 0; q < RT_NQ; ++q) {
        const int t = RT_NQ * wave + q;
        wq[q] = w + (int64_t)(code0 + t < K ? code0 + t : 0) * g.D + 4 * lane;
        xq[q] = g.x + (row0 + t < g.R ? patch_row_base(g, row0 + t) : 0) + 4 * lane;
    }
    // (two chunks of loads in flight measured slower: 66 against 46 us on the 8 x 8 tile)
    float4 xr[RT_NQ], wr[RT_NQ];
    auto fetch = [&](int e0) {
#pragma unroll
        for (int q = 0; q < RT_NQ; ++q) {
            if constexpr (NX) xr[q] = *reinterpret_cast<const float4*>(xq[q] + e0);
            if constexpr (NW) wr[q] = *reinterpret_cast<const float4*>(wq[q] + e0);
        }
    };
    float acc = 0.0f;
    fetch(0);
    for (int e0 = 0; e0 < g.D; e0 += RT_CH) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RT_NQ; ++q) {
            const int t = RT_NQ * wave + q;
            if constexpr (ROLE == 0)            // -2 x: (-2 w) x == w (-2 x) exactly
                *reinterpret_cast<float4*>(&xs[t][4 * lane]) =
                    make_float4(-2.0f * xr[q].x, -2.0f * xr[q].y, -2.0f * xr[q].z, -2.0f * xr[q].w);
            if constexpr (ROLE == 2) *reinterpret_cast<float4*>(&xs[t][4 * lane]) = xr[q];
            if constexpr (NW) *reinterpret_cast<float4*>(&ws[t][4 * lane]) = wr[q];
        }
        __syncthreads();
        if (e0 + RT_CH < g.D) fetch(e0 + RT_CH);
#define QARIG_FR_FMAC(ACC, A, B) asm("v_fmac_f32 %0, %1, %2" : "+v"(ACC) : "v"(A), "v"(B))
#pragma unroll 8
        for (int e = 0; e < RT_CH; e += 4) {
            float4 A, B;
            if constexpr (ROLE == 0) { A = *reinterpret_cast<const float4*>(&ws[c][e]); B = *reinterpret_cast<const float4*>(&xs[r][e]); }
            if constexpr (ROLE == 1) { A = *reinterpret_cast<const float4*>(&ws[c][e]); B = A; }
            if constexpr (ROLE == 2) { A = *reinterpret_cast<const float4*>(&xs[r][e]); B = A; }
            QARIG_FR_FMAC(acc, A.x, B.x);
            QARIG_FR_FMAC(acc, A.y, B.y);
            QARIG_FR_FMAC(acc, A.z, B.z);
            QARIG_FR_FMAC(acc, A.w, B.w);
        }
#undef QARIG_FR_FMAC
    }
    const int row = row0 + r, code = code0 + c;
    if constexpr (ROLE == 0) { if (row < g.R && code < K) acc_out[(int64_t)row * K + code] = acc; }
    if constexpr (ROLE == 1) { if (r == 0 && code < K) w2_out[code] = acc; }
    if constexpr (ROLE == 2) { if (c == 0 && row < g.R) x2_out[row] = acc; }
}
__global__ __launch_bounds__(RT_THREADS) void bmu_fewrows_roles_kernel(PatchGeom g, const float* __restrict__ w,
                                                                       int K, int row_tiles,
                                                                       float* __restrict__ acc_out,
                                                                       float* __restrict__ w2_out,
                                                                       float* __restrict__ x2_out) {
    __shared__ __attribute__((aligned(16))) float xs[RT_T][RT_LD];
    __shared__ __attribute__((aligned(16))) float ws[RT_T][RT_LD];
    const int by = blockIdx.y;                  // workgroup-uniform
    if (by < row_tiles)
        fewrows_role<0>(g, w, K, blockIdx.x * RT_T, by * RT_T, xs, ws, acc_out, w2_out, x2_out);
    else if (by == row_tiles)
        fewrows_role<1>(g, w, K, blockIdx.x * RT_T, 0, xs, ws, acc_out, w2_out, x2_out);
    else if ((int)blockIdx.x < row_tiles)
        fewrows_role<2>(g, w, K, 0, blockIdx.x * RT_T, xs, ws, acc_out, w2_out, x2_out);
}

// The general form: any D, any patch shape, ragged rows / codes; every lane carries the three chains
// (128-element chunks, each waited for between two barriers).
constexpr int FR_CH = 128;    // elements per staged chunk
constexpr int FR_T = 8;       // 8 rows x 8 codes per 64-lane workgroup: 512 workgroups at 64 x 512
constexpr int FR_LD = FR_CH + 4;   // 16-B aligned rows, lanes of different codes on different banks
__global__ __launch_bounds__(64) void bmu_fewrows_dot_kernel(PatchGeom g, const float* __restrict__ w,
                                                             int K, float* __restrict__ acc_out,
                                                             float* __restrict__ w2_out,
                                                             float* __restrict__ x2_out) {
    __shared__ __attribute__((aligned(16))) float xs[FR_T][FR_LD];
    __shared__ __attribute__((aligned(16))) float ws[FR_T][FR_LD];
    const int tid = threadIdx.x;
    const int c = tid & 7, r = tid >> 3;
    const int code0 = blockIdx.x * FR_T, row0 = blockIdx.y * FR_T;
    // staging: lane -> (tile row tid >> 3, 16 consecutive elements at (tid & 7) * 16)
    const int sr = tid >> 3, se = (tid & 7) * 16;
    const int srow = row0 + sr, scode = code0 + sr;
    const int64_t rbase = srow < g.R ? patch_row_base(g, srow) : 0;
    const bool xvec = (g.pW & 3) == 0 && (g.W & 3) == 0 && (((uintptr_t)g.x) & 15) == 0;
    const bool wvec = (g.D & 3) == 0 && (((uintptr_t)w) & 15) == 0;
    float acc = 0.0f, w2 = 0.0f, x2 = 0.0f;
    for (int e0 = 0; e0 < g.D; e0 += FR_CH) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
            const int e = e0 + se + q;
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), wv = xv;
            if (srow < g.R) {
                if (xvec && e + 3 < g.D) {
                    const int j = e % g.pW, t = e / g.pW;
                    const int i = t % g.pH, ch = t / g.pH;
                    xv = *reinterpret_cast<const float4*>(g.x + rbase + ((int64_t)ch * g.H + i) * g.W + j);
                } else {
                    float t4[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int u = 0; u < 4; ++u)
                        if (e + u < g.D) {
                            const int j = (e + u) % g.pW, t = (e + u) / g.pW;
                            const int i = t % g.pH, ch = t / g.pH;
                            t4[u] = g.x[rbase + ((int64_t)ch * g.H + i) * g.W + j];
                        }
                    xv = make_float4(t4[0], t4[1], t4[2], t4[3]);
                }
            }
            if (scode < K) {
                const float* wp = w + (int64_t)scode * g.D + e;
                if (wvec && e + 3 < g.D) wv = *reinterpret_cast<const float4*>(wp);
                else
                    wv = make_float4(e < g.D ? wp[0] : 0.f, e + 1 < g.D ? wp[1] : 0.f,
                                     e + 2 < g.D ? wp[2] : 0.f, e + 3 < g.D ? wp[3] : 0.f);
            }
            *reinterpret_cast<float4*>(&xs[sr][se + q]) = xv;
            *reinterpret_cast<float4*>(&ws[sr][se + q]) = wv;
        }
        __syncthreads();
        // zero padding past D contributes fmaf(-0, 0, acc) = acc and fmaf(0, 0, n2) = n2: exact
#pragma unroll 8
        for (int e = 0; e < FR_CH; e += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(&xs[r][e]);
            const float4 wv = *reinterpret_cast<const float4*>(&ws[c][e]);
            acc = fmaf(-2.0f * wv.x, xv.x, acc);
            acc = fmaf(-2.0f * wv.y, xv.y, acc);
            acc = fmaf(-2.0f * wv.z, xv.z, acc);
            acc = fmaf(-2.0f * wv.w, xv.w, acc);
            w2 = fmaf(wv.x, wv.x, w2); w2 = fmaf(wv.y, wv.y, w2);
            w2 = fmaf(wv.z, wv.z, w2); w2 = fmaf(wv.w, wv.w, w2);
            x2 = fmaf(xv.x, xv.x, x2); x2 = fmaf(xv.y, xv.y, x2);
            x2 = fmaf(xv.z, xv.z, x2); x2 = fmaf(xv.w, xv.w, x2);
        }
    }
    const int row = row0 + r, code = code0 + c;
    if (row < g.R && code < K) acc_out[(int64_t)row * K + code] = acc;
    if (r == 0 && blockIdx.y == 0 && code < K) w2_out[code] = w2;
    if (c == 0 && blockIdx.x == 0 && row < g.R) x2_out[row] = x2;
}

__global__ __launch_bounds__(256) void bmu_fewrows_argmin_kernel(const float* __restrict__ acc, int K,
                                                                 const float* __restrict__ w2,
                                                                 const float* __restrict__ x2,
                                                                 int64_t* __restrict__ out) {
    __shared__ float bd[256];
    __shared__ int bi[256];
    const int row = blockIdx.x;
    const float xx = x2[row];
    float best = INFINITY;
    int idx = INT_MAX;
    for (int k = threadIdx.x; k < K; k += 256) {          // ascending k per lane: strict < keeps the first
        const float d = sqrtf(fmaxf((acc[(int64_t)row * K + k] + w2[k]) + xx, 0.0f));
        if (d < best) { best = d; idx = k; }
    }
    bd[threadIdx.x] = best;
    bi[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float od = bd[threadIdx.x + o];
            const int oi = bi[threadIdx.x + o];
            if (od < bd[threadIdx.x] || (od == bd[threadIdx.x] && oi < bi[threadIdx.x])) {
                bd[threadIdx.x] = od;
                bi[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[row] = bi[0] == INT_MAX ? 0 : (int64_t)bi[0];
}

}  // namespace qarig

using namespace qarig;

// ---------------------------------------------------------------------------------------------
// Coarse pass on the bf16 MFMA + certificate + exact re-scan (D <= 16, K <= 1024, K % 32 == 0).
//
// The exact kernels above run at the SUM of their fp32-MFMA and vector-ALU cycles (the two do not
// overlap on this chip: DESIGN 10) -- 9 MFMAs of 64 cycles and a 16-candidate scan per 32 x 32
// tile.  The bf16 MFMA is 16x faster per product AND co-executes with the vector ALU, so here the
// candidates' t = -2 x.w + |w|^2 come from the bf16 pipe at (almost) full fp32 accuracy and the
// scan is all that is left on the vector ALU:
//   * every operand is split into three bf16 pieces, v = h + m + l EXACTLY (3 x 8 = 24 mantissa
//     bits, successive round-to-nearest remainders), and the six products hh, hm, mh, hl, lh, mm are
//     accumulated in fp32 by six v_mfma_f32_32x32x16_bf16 (K = 16 = the whole patch width in one
//     instruction); |w|^2 (its exact fp32 chain) is the accumulator's initial value, read from an LDS
//     copy in accumulator order.  6 x 32 = 192 matrix cycles per tile, beside the scan's ~200.
//   * the scan is three vector instructions per candidate: the candidate's register number (which of the
//     16 codes of its tile and lane half) replaces the low four bits of its value (v_and_or_b32), then
//     second = median(min, t, second), min = median(min, t, -inf): no compare / select pair, and the
//     minimum's code comes out of its own low bits.
//   * what the coarse value can be off by: the three dropped products (<= 2^-26 |x||2w| per element),
//     the rounding of the 6 x 16 fp32 additions inside the matrix core, the four replaced bits
//     (15 ulp), against the definition's own 17-step chain:
//     |t~ - t| <= eps = 1e-5 (|x|^2 + 2 max|w|^2)  (sum|terms| <= 2|x||w| + |w|^2; BMU_COARSE_EPS below).
//   * certificate: the scan keeps (min, its first index, second-smallest) of t~ per row.  If
//     second - min > 3 eps, the exact minimum is the same candidate and no other candidate can tie
//     with it after the definition's `+ |x|^2`, clamp and sqrt (those collapse values closer than
//     ~2^-22 (t + |x|^2) < eps): the index is final.  Every other row (near-ties, exact ties,
//     non-finite data) is re-scanned with the literal definition -- codes rebuilt EXACTLY from their
//     LDS image (h + m + l; a code whose split is not exact, e.g. a denormal piece, sends the block
//     to the fp32 codebook in memory instead), the row's fp32 values from an LDS copy -- so the
//     result is bit-identical to oracle/bmu_oracle.c on every input, as before.
typedef __bf16 bmu_bf16x8 __attribute__((ext_vector_type(8)));
// v128 .. v224: the registers the scan loop's asm statement names (tools/gen_bmu_scan.py)
#define BMU_V8(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define BMU_SCAN_CLOBBERS "v128", "v129", BMU_V8(13), BMU_V8(14), BMU_V8(15), BMU_V8(16), BMU_V8(17), BMU_V8(18), \
    BMU_V8(19), BMU_V8(20), BMU_V8(21), "v220", "v221", "v222", "v223", "v224"
typedef unsigned int bmu_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bmu_pack_bf16(float a, float b) {   // a in the low half
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
}
// (a, b) = H + M + L piecewise; returns false if the three pieces do not add up to the values
__device__ __forceinline__ bool bmu_split3(float a, float b, uint32_t& H, uint32_t& M, uint32_t& L) {
    H = bmu_pack_bf16(a, b);
    const float ha = __uint_as_float(H << 16), hb = __uint_as_float(H & 0xffff0000u);
    const float ra = a - ha, rb = b - hb;
    M = bmu_pack_bf16(ra, rb);
    const float ma = __uint_as_float(M << 16), mb = __uint_as_float(M & 0xffff0000u);
    const float sa = ra - ma, sb = rb - mb;
    L = bmu_pack_bf16(sa, sb);
    const float la = __uint_as_float(L << 16), lb = __uint_as_float(L & 0xffff0000u);
    return (ha + ma) + la == a && (hb + mb) + lb == b;
}
// where code k's |w|^2 sits in the accumulator-ordered copy: the accumulator register r of lane half h of tile T
// holds code 32 T + 4 h + (r & 3) + 8 (r >> 2)
__device__ __forceinline__ int bmu_w2_at(int k) {
    const int c = k & 31;
    return ((k >> 5) * 2 + ((c >> 2) & 1)) * 16 + (c & 3) + 4 * (c >> 3);
}
__device__ __forceinline__ float bmu_join3(uint32_t H, uint32_t M, uint32_t L, int hi) {
    const uint32_t mask = hi ? 0xffff0000u : 0u;
    const float h = __uint_as_float(hi ? (H & mask) : (H << 16));
    const float m = __uint_as_float(hi ? (M & mask) : (M << 16));
    const float l = __uint_as_float(hi ? (L & mask) : (L << 16));
    return (h + m) + l;
}

// Bound of |coarse value - the definition's fp32 chain| in units of S = |x|^2 + 2 max|w|^2 (u = 2^-24):
//   * the 6 x 16 additions inside the matrix core, each charged a rounding of u of the running magnitude
//     (<= sum|terms| <= 2|x||w| + |w|^2 <= S):                                     gamma_96  = 5.7e-6
//   * the three dropped piece products, 2^-26 |x||2w| per element:                              0.05e-6
//   * the candidate's register number in its low four bits: <= 15 ulp = 30 u:                   1.8e-6
//   * the 17-step rounding of the definition's own chain:                                       1.0e-6
//   in all 8.6e-6 <= eps = 1e-5.  What v_mfma_f32_32x32x16_bf16 really does was measured
// (tools/mfma_rounding_probe.hip, profiles/r04_mfma_rounding.log): the 16 products and C are aligned to the
// largest exponent among them with ONE guard bit, bits below it are dropped (a product of 0.75 ulp beside 1.0
// contributes 0.5 ulp; sixteen products of 1/16 ulp beside C = 1 vanish), the sum is then rounded to nearest even
// (C = 1 plus 0.75 ulp gives 1 + 1 ulp, ties go to even).  Worst case per instruction: 16 x 2^-25 of the largest
// addend from the alignment + 2^-24 from the final rounding = 9 u of it, i.e. 54 u = 3.2e-6 over the six
// instructions -- inside the 96 u the derivation charges.  With the certificate's factor 3 the remaining true
// gap exceeds 1e-5 of the scale, far above what clamp + sqrt can collapse (2^-22).  (2e-5 was tried: the 65,536 x
// 512 x 16 launch re-scans 148 rows instead of 73 and takes 18.4 instead of 16.3 us.)
// tests/test_gpu_core.py::test_bmu_coarse_pass_on_constructed_near_ties sweeps code pairs whose gap runs from
// far below eps to above 4 eps.
constexpr float BMU_COARSE_EPS = 1e-5f;

// PREP: stages the image of a codebook (what every block of the search kernel builds in LDS) into
// `image` instead: [3 planes: K x 96 B][|w|^2 in accumulator order: K x 4 B][max |w|^2, inexact flag: 8 B].  A frozen
// codebook (tokenising a dataset, the Transformer training loop) is prepared once and its image
// copied into LDS by every later launch (qarig_bmu_prepare).
template <bool PREP>
__global__ __launch_bounds__(NTHREADS, 2) void bmu_coarse_kernel(PatchGeom g, PatchOffsets po,
                                                                 const float* __restrict__ w, int K,
                                                                 int64_t* __restrict__ out,
                                                                 unsigned* __restrict__ stats,
                                                                 const unsigned char* __restrict__ image_in,
                                                                 unsigned char* __restrict__ image_out) {
    // planes [hi | mid | lo] of fragments: [plane][tile][half][code] x 16 B
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    bmu_u32x4* frag = reinterpret_cast<bmu_u32x4*>(lds_raw);
    const int NT = K >> 5;
    const int plane = NT * 64;                                  // fragments per plane
    // exact |w|^2 chains in the accumulator's order: [tile][lane half][register] (bmu_w2_at)
    float* W2 = reinterpret_cast<float*>(frag + 3 * plane);
    float* XS = W2 + K;                                         // [128][16] the block's patch rows, fp32
    float* red = XS + 128 * 16;                                 // [8] reductions, [8..12) flag masks, [12] inexact
    unsigned* flagmask = reinterpret_cast<unsigned*>(red + 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, cl = lane & 31;
    const int p0 = blockIdx.x * 128;
    const int D = g.D;
    const long long tc0 = stats ? clock64() : 0;      // phase clocks (tests / tools only)

    // ---- this lane's patch row: 8 of its 16 elements (e = 8h .. 8h+7); the loads are issued first
    // so that their latency runs under the codebook staging
    const int prow = p0 + wave * 32 + cl;
    const bool live = prow < g.R;
    float xv[8];
    if constexpr (!PREP) {
        const float* px = g.x + patch_row_base_fast(g, po, live ? prow : g.R - 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int off = h ? po.off[8 + q] : po.off[q];        // (po.off[e >= D] = po.off[0]: a valid address)
            xv[q] = px[off];
        }
    }
    // ---- stage the codebook: -2w split three ways, |w|^2 chain, its three pieces; two codes per
    // pass with all eight 16-B loads in flight
    float w2max = 0.0f;
    int inexact = 0;
    const size_t image_bytes = (size_t)K * 100;               // planes + |w|^2 (a multiple of 16: K % 32 == 0)
    if (!PREP && image_in) {
        // a prepared image: global -> LDS without a register round trip (1 KiB per wave instruction, every piece
        // of the image in flight at once); the last partial KiB through registers
        const int n16 = (int)(image_bytes >> 4);
        const int nblk = n16 >> 6;
        for (int b = wave; b < nblk; b += NTHREADS / 64)
            __builtin_amdgcn_global_load_lds(
                (__attribute__((address_space(1))) const void*)(image_in + ((size_t)b * 64 + lane) * 16),
                (__attribute__((address_space(3))) void*)(lds_raw + (size_t)b * 1024), 16, 0, 0);
        const uint4* src = reinterpret_cast<const uint4*>(image_in);
        uint4* dst = reinterpret_cast<uint4*>(lds_raw);
        if (nblk * 64 + tid < n16) dst[nblk * 64 + tid] = src[nblk * 64 + tid];
        const float* hdr = reinterpret_cast<const float*>(image_in + image_bytes);
        w2max = hdr[0];
        inexact = hdr[1] != 0.0f;
    } else
    for (int k0 = tid; k0 < K; k0 += 2 * NTHREADS) {
        float4 raw[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = min(k0 + u * NTHREADS, K - 1);
            const float4* wk4 = reinterpret_cast<const float4*>(w + (int64_t)k * D);   // host: D % 4 == 0, 16-B aligned
#pragma unroll
            for (int q = 0; q < 4; ++q) raw[u][q] = 4 * q < D ? wk4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + u * NTHREADS;
            if (k >= K) break;
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[4 * q] = raw[u][q].x; v[4 * q + 1] = raw[u][q].y; v[4 * q + 2] = raw[u][q].z; v[4 * q + 3] = raw[u][q].w;
            }
            float w2 = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) w2 = e < D ? fmaf(v[e], v[e], w2) : w2;
            W2[bmu_w2_at(k)] = w2;
            w2max = fmaxf(w2max, w2);
            if (!(w2 <= 3.0e38f)) inexact = 1;          // a NaN / Inf code: the definition's clamp decides, exactly
            const int T = k >> 5, c = k & 31;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                bmu_u32x4 H, M, L;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t a, b, cc;
                    if (!bmu_split3(-2.0f * v[8 * hh + 2 * q], -2.0f * v[8 * hh + 2 * q + 1], a, b, cc)) inexact = 1;
                    H[q] = a; M[q] = b; L[q] = cc;
                }
                const int at = (T * 2 + hh) * 32 + c;
                frag[at] = H;
                frag[plane + at] = M;
                frag[2 * plane + at] = L;
            }
        }
    }
    float x2a = 0.0f;
    bmu_u32x4 XH = {0u, 0u, 0u, 0u}, XM = XH, XL = XH;
    if constexpr (!PREP) {
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[q] = 8 * h + q < D ? xv[q] : 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t a, b, cc;
            if (!bmu_split3(xv[2 * q], xv[2 * q + 1], a, b, cc)) inexact = 1;   // (the coarse value is then off; the
            XH[q] = a; XM[q] = b; XL[q] = cc;                                   //  block re-scans every row exactly)
            x2a += xv[2 * q] * xv[2 * q] + xv[2 * q + 1] * xv[2 * q + 1];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) XS[(wave * 32 + cl) * 16 + 8 * h + q] = xv[q];
        x2a += __shfl_xor(x2a, 32);
    }
    // the image's DMA has run under the split of the patch rows: landed before the barrier publishes it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // block-wide max |w|^2 and the inexact flag
    w2max = wave_max(w2max);
    inexact = __any(inexact);
    if (lane == 0) { red[wave] = w2max; red[4 + wave] = inexact ? 1.0f : 0.0f; }
    __syncthreads();
    w2max = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const bool any_inexact = (red[4] + red[5] + red[6] + red[7]) != 0.0f;   // then every row is re-scanned
    if constexpr (PREP) {
        const uint4* src = reinterpret_cast<const uint4*>(lds_raw);
        uint4* dst = reinterpret_cast<uint4*>(image_out);
        for (int i = tid; i < (int)(image_bytes >> 4); i += NTHREADS) dst[i] = src[i];
        if (tid == 0) {
            float* hdr = reinterpret_cast<float*>(image_out + image_bytes);
            hdr[0] = w2max;
            hdr[1] = any_inexact ? 1.0f : 0.0f;
        }
        return;
    }

    const long long tc1 = stats ? clock64() : 0;
    // ---- the coarse scan, software-pipelined over three accumulator sets in rotation: while tile T is scanned,
    // the six MFMAs of tile T+1 (a dependent chain of 32-cycle instructions on the matrix pipe, started from the
    // tile's |w|^2) are issued with seven scan instructions behind each, and the fragments and |w|^2 of tile T+2
    // come in from LDS -- no LDS read is waited for in the step that issued it, no register copies, no idle wait
    // states behind a chain.  The loop is one inline-asm statement (tools/gen_bmu_scan.py writes it and documents
    // its registers): hipcc's scheduler did not keep this shape under any sched_group_barrier pipeline.
    // scan state: (min, second-smallest) of t~, the minimum carrying its register number in its low four bits
    // (3 vector instructions per candidate: v_and_or_b32, 2 x v_med3_f32), and the tile in which the minimum fell
    float best = INFINITY, sec = INFINITY;
    int tidx = -1;
    if constexpr (!PREP) {
        const bmu_bf16x8 bXH = __builtin_bit_cast(bmu_bf16x8, XH), bXM = __builtin_bit_cast(bmu_bf16x8, XM),
                         bXL = __builtin_bit_cast(bmu_bf16x8, XL);
        const unsigned f0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)(frag + h * 32 + cl);
        const unsigned wa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)(W2 + h * 16);
        const int plane_bytes = __builtin_amdgcn_readfirstlane(plane * 16), nt = __builtin_amdgcn_readfirstlane(NT);
        int st_, su_;
        asm volatile(
#include "bmu_scan_asm.inc"
            : [best] "=&v"(best), [sec] "=&v"(sec), [tidx] "=&v"(tidx), [t] "=&s"(st_), [u] "=&s"(su_)
            : [xh] "v"(bXH), [xm] "v"(bXM), [xl] "v"(bXL), [f0] "v"(f0), [pl] "s"(plane_bytes), [w] "v"(wa), [nt] "s"(nt)
            : "vcc", "scc", "memory", BMU_SCAN_CLOBBERS);
    }
    const int ridx = (int)(__float_as_uint(best) & 15u);
    const long long tc2 = stats ? clock64() : 0;
    // ---- merge the two lane halves of a row, certify
    const int idx = tidx < 0 ? INT_MAX : tidx * 32 + 4 * h + (ridx & 3) + 8 * (ridx >> 2);
    BmuState st = bmu_merge(BmuState{best, idx, sec},
                            BmuState{__shfl_xor(best, 32), __shfl_xor(idx, 32), __shfl_xor(sec, 32)});
    const float eps = BMU_COARSE_EPS * (1.001f * x2a + 2.0f * w2max);
    const bool certified = !any_inexact && (st.sec - st.d2 > 3.0f * eps);
    const int flag = live && h == 0 && !certified;
    if (live && h == 0 && certified) out[prow] = (int64_t)st.idx;
    const unsigned long long fm = __ballot(flag);
    if (lane == 0) flagmask[wave] = (unsigned)fm;
    __syncthreads();
    if (stats && tid == 0) {
        const unsigned n = __popc(flagmask[0]) + __popc(flagmask[1]) + __popc(flagmask[2]) + __popc(flagmask[3]);
        if (n) atomicAdd(stats, n);
    }
    // ---- literal re-scan of the rows without a certificate
    const bool from_memory = any_inexact;
    float* rs = red;                                   // [4] per-wave minima (red[0..3] no longer needed)
    int* ri = reinterpret_cast<int*>(red + 4);
    for (int wv = 0; wv < 4; ++wv)
    for (unsigned m = flagmask[wv]; m; m &= m - 1) {
        const int r = wv * 32 + __ffs((int)m) - 1;
        float xs[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) xs[e] = XS[r * 16 + e];
        float x2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) x2 = e < D ? fmaf(xs[e], xs[e], x2) : x2;
        float bd = INFINITY;
        int bi = INT_MAX;
        for (int k = tid; k < K; k += NTHREADS) {
            float m2[16];
            if (from_memory) {
                const float* wk = w + (int64_t)k * D;
#pragma unroll
                for (int e = 0; e < 16; ++e) m2[e] = e < D ? -2.0f * wk[e] : 0.0f;
            } else {
                const int T = k >> 5, c = k & 31;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int at = (T * 2 + hh) * 32 + c;
                    const bmu_u32x4 H = frag[at], M = frag[plane + at], L = frag[2 * plane + at];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        m2[8 * hh + 2 * q] = bmu_join3(H[q], M[q], L[q], 0);
                        m2[8 * hh + 2 * q + 1] = bmu_join3(H[q], M[q], L[q], 1);
                    }
                }
            }
            float acc = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc = e < D ? fmaf(m2[e], xs[e], acc) : acc;
            const float d = sqrtf(fmaxf((acc + W2[bmu_w2_at(k)]) + x2, 0.0f));
            if (d < bd) { bd = d; bi = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float s2 = __shfl_xor(bd, o);
            const int i2 = __shfl_xor(bi, o);
            if (s2 < bd || (s2 == bd && i2 < bi)) { bd = s2; bi = i2; }
        }
        __syncthreads();
        if (lane == 0) { rs[wave] = bd; ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (rs[q] < bd || (rs[q] == bd && ri[q] < bi)) { bd = rs[q]; bi = ri[q]; }
            out[p0 + r] = bi == INT_MAX ? 0 : (int64_t)bi;
        }
    }
    if (stats && tid == 0) {
        const long long tc3 = clock64();
        atomicAdd(stats + 1, (unsigned)(tc1 - tc0));
        atomicAdd(stats + 2, (unsigned)(tc2 - tc1));
        atomicAdd(stats + 3, (unsigned)(tc3 - tc2));
        atomicAdd(stats + 4, 1u);
    }
}

static int bmu_code_tiles(int K) { return (K + BM - 1) / BM; }

static constexpr int FEWROWS_MAX = 128, FEWROWS_MIN_D = 512;

extern "C" size_t qarig_bmu_workspace_bytes(int64_t rows, int K) {
    if (rows < 0 || rows > (1LL << 31) || K < 1 || K > (1 << 24)) return 0;     // refused by qarig_bmu_fwd
    // per-split (min, idx, second) partials for up to code_tiles splits + |x|^2 per row
    size_t need = (size_t)bmu_code_tiles(K) * (size_t)rows * 12 + (size_t)rows * 4 + 64;
    // few-rows form: rows x K dot products + |w|^2 + |x|^2
    if (rows <= FEWROWS_MAX) need = need > ((size_t)rows * K + K + rows) * 4 ? need : ((size_t)rows * K + K + rows) * 4;
    return need;
}

static bool bmu_coarse_ok(const PatchGeom& g, int K, const float* codebook) {
    return g.D <= 16 && g.D % 4 == 0 && K % 32 == 0 && K >= 32 && K <= 1024 && g.R > 25 &&
           (((uintptr_t)codebook) & 15) == 0;
}

static void bmu_patch_offsets(const PatchGeom& g, PatchOffsets& po) {
    for (int e = 0; e < 16; ++e) {
        const int ee = e < g.D ? e : 0;
        const int j = ee % g.pW, i = (ee / g.pW) % g.pH, c = ee / (g.pW * g.pH);
        po.off[e] = (c * g.H + i) * g.W + j;
    }
    auto magic = [](int d, unsigned& m, int& sh) {
        int L = 0;
        while ((1LL << L) < d) ++L;
        sh = 31 + L;
        m = (unsigned)((1ULL << sh) / (unsigned long long)d + 1ULL);
    };
    magic(g.gh * g.gw, po.m_per, po.sh_per);
    magic(g.gw, po.m_gw, po.sh_gw);
}

static size_t bmu_coarse_lds(int K) { return (size_t)K * 96 + (size_t)K * 4 + 128 * 16 * 4 + 16 * 4; }
static void bmu_coarse_attr(size_t shm) {
    static size_t attr_shm = 0;
    if (shm > attr_shm) {
        (void)hipFuncSetAttribute((const void*)bmu_coarse_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        (void)hipFuncSetAttribute((const void*)bmu_coarse_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr_shm = shm;
    }
}

static int bmu_coarse_launch(const PatchGeom& g, const float* codebook, int K, int64_t* out_idx,
                             unsigned* stats, const void* image, hipStream_t st) {
    PatchOffsets po;
    bmu_patch_offsets(g, po);
    const size_t shm = bmu_coarse_lds(K);
    bmu_coarse_attr(shm);
    hipLaunchKernelGGL(bmu_coarse_kernel<false>, dim3((g.R + 127) / 128), dim3(NTHREADS), shm, st, g, po, codebook, K,
                       out_idx, stats, (const unsigned char*)image, (unsigned char*)nullptr);
    QARIG_CHECK_LAUNCH("bmu coarse");
    return QARIG_OK;
}

// Image of a codebook for qarig_bmu_fwd_coarse (its `prepared` argument): bytes, and the one-block kernel
// that writes it.  0 bytes = the coarse form does not take this codebook.
extern "C" size_t qarig_bmu_prepare_bytes(int K, int D) {
    if (D < 1 || D > 16 || D % 4 || K < 32 || K > 1024 || K % 32) return 0;
    return (size_t)K * 100 + 16;
}
extern "C" int qarig_bmu_prepare(const float* codebook, int K, int D, void* image, void* stream) {
    QARIG_CHECK_ARG(codebook && image, "bmu_prepare: null pointer");
    QARIG_CHECK_ARG(qarig_bmu_prepare_bytes(K, D) != 0 && (((uintptr_t)codebook | (uintptr_t)image) & 15) == 0,
                    "bmu_prepare: needs D <= 16, D %% 4 == 0, K %% 32 == 0, 32 <= K <= 1024, 16-B aligned pointers");
    PatchGeom g{nullptr, 1, 1, 1, 1, 1, 1, 1, 1, D, 1};
    PatchOffsets po{};
    const size_t shm = bmu_coarse_lds(K);
    bmu_coarse_attr(shm);
    hipLaunchKernelGGL(bmu_coarse_kernel<true>, dim3(1), dim3(NTHREADS), shm, (hipStream_t)stream, g, po, codebook, K,
                       (int64_t*)nullptr, (unsigned*)nullptr, (const unsigned char*)nullptr, (unsigned char*)image);
    QARIG_CHECK_LAUNCH("bmu prepare");
    return QARIG_OK;
}

// The coarse-pass form on its own (tests, benchmarks): fails where qarig_bmu_fwd would fall back to
// the exact kernels; *uncertified (device, caller-zeroed, may be NULL) += rows that took the exact
// re-scan.
extern "C" int qarig_bmu_fwd_coarse(const float* x, int N, int C, int H, int W, int pH, int pW,
                                    const float* codebook, int K, int D, int64_t* out_idx,
                                    unsigned* uncertified, const void* prepared, void* stream) {
    QARIG_CHECK_ARG(x && codebook && out_idx, "bmu_coarse: null pointer");
    QARIG_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && pH > 0 && pW > 0 && K > 0 && pH <= H && pW <= W,
                    "bmu_coarse: bad extents");
    QARIG_CHECK_DIMS("bmu_coarse", N, C, H, W);
    QARIG_CHECK_DIMS("bmu_coarse", K, C, pH, pW);
    QARIG_CHECK_ARG(D == C * pH * pW, "bmu_coarse: codebook width %d != C*pH*pW = %d", D, C * pH * pW);
    PatchGeom g{x, N, C, H, W, pH, pW, H / pH, W / pW, D, 0};
    const int64_t rows = (int64_t)N * g.gh * g.gw;
    QARIG_CHECK_ARG(rows < INT_MAX, "bmu_coarse: too many patch rows");
    g.R = (int)rows;
    QARIG_CHECK_ARG(bmu_coarse_ok(g, K, codebook),
                    "bmu_coarse: needs D <= 16, D %% 4 == 0, K %% 32 == 0, 32 <= K <= 1024, more than 25 rows");
    QARIG_CHECK_ARG(!prepared || (((uintptr_t)prepared) & 15) == 0, "bmu_coarse: prepared image must be 16-B aligned");
    return bmu_coarse_launch(g, codebook, K, out_idx, uncertified, prepared, (hipStream_t)stream);
}

extern "C" int qarig_bmu_fwd_prepared(const float* x, int N, int C, int H, int W, int pH, int pW,
                                      const float* codebook, int K, int D, int64_t* out_idx,
                                      void* workspace, size_t ws_bytes, const void* prepared, void* stream);
extern "C" int qarig_bmu_fwd(const float* x, int N, int C, int H, int W, int pH, int pW,
                             const float* codebook, int K, int D, int64_t* out_idx,
                             void* workspace, size_t ws_bytes, void* stream) {
    return qarig_bmu_fwd_prepared(x, N, C, H, W, pH, pW, codebook, K, D, out_idx, workspace, ws_bytes, nullptr, stream);
}

// qarig_bmu_fwd with the codebook's prepared image (qarig_bmu_prepare; NULL = none): where the dispatch takes the
// coarse-pass kernel its workgroups copy the image into LDS instead of converting the codebook themselves.
extern "C" int qarig_bmu_fwd_prepared(const float* x, int N, int C, int H, int W, int pH, int pW,
                                      const float* codebook, int K, int D, int64_t* out_idx,
                                      void* workspace, size_t ws_bytes, const void* prepared, void* stream) {
    QARIG_CHECK_ARG(x && codebook && out_idx, "bmu: null pointer");
    QARIG_CHECK_ARG(!prepared || (((uintptr_t)prepared) & 15) == 0, "bmu: prepared image must be 16-B aligned");
    QARIG_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && pH > 0 && pW > 0 && K > 0,
                    "bmu: bad extents");
    QARIG_CHECK_ARG(pH <= H && pW <= W, "bmu: patch larger than the latent");
    QARIG_CHECK_DIMS("bmu", N, C, H, W);
    QARIG_CHECK_DIMS("bmu", K, C, pH, pW);
    QARIG_CHECK_ARG(D == C * pH * pW, "bmu: codebook width %d != C*pH*pW = %d", D, C * pH * pW);
    PatchGeom g{x, N, C, H, W, pH, pW, H / pH, W / pW, D, 0};
    const int64_t rows = (int64_t)N * g.gh * g.gw;
    QARIG_CHECK_ARG(rows < INT_MAX, "bmu: too many patch rows");
    g.R = (int)rows;
    hipStream_t st = (hipStream_t)stream;

    if (g.R <= 25 && K <= 25) {
        hipLaunchKernelGGL(bmu_direct_kernel, dim3((g.R + 63) / 64), dim3(64), 0, st, g, codebook, K,
                           out_idx);
        QARIG_CHECK_LAUNCH("bmu direct");
        return QARIG_OK;
    }
    if (!workspace || ws_bytes < qarig_bmu_workspace_bytes(rows, K)) {
        qarig_set_error("bmu: workspace too small (%zu < %zu)", ws_bytes,
                        qarig_bmu_workspace_bytes(rows, K));
        return QARIG_ERR_WORKSPACE;
    }
    if (g.R <= FEWROWS_MAX && D >= FEWROWS_MIN_D) {
        float* acc = (float*)workspace;
        float* w2 = acc + (size_t)g.R * K;
        float* x2 = w2 + K;
        const bool fast = pW == W && pH == H && (D % RT_CH) == 0 &&
                          ((((uintptr_t)x) | ((uintptr_t)codebook)) & 15) == 0;
        const int rt = (g.R + RT_T - 1) / RT_T, ct = (K + RT_T - 1) / RT_T;
        // (the |x|^2 role takes the first rt blocks of the last grid row)
        if (fast && rt <= ct)
            hipLaunchKernelGGL(bmu_fewrows_roles_kernel, dim3(ct, rt + 2), dim3(RT_THREADS), 0, st, g, codebook, K, rt,
                               acc, w2, x2);
        else
            hipLaunchKernelGGL(bmu_fewrows_dot_kernel, dim3((K + 7) / 8, (g.R + 7) / 8), dim3(64), 0, st, g,
                               codebook, K, acc, w2, x2);
        QARIG_CHECK_LAUNCH("bmu fewrows dot");
        hipLaunchKernelGGL(bmu_fewrows_argmin_kernel, dim3(g.R), dim3(256), 0, st, acc, K, w2, x2, out_idx);
        QARIG_CHECK_LAUNCH("bmu fewrows argmin");
        return QARIG_OK;
    }
    float* part_d = (float*)workspace;
    int* part_i = (int*)(part_d + (size_t)bmu_code_tiles(K) * g.R);
    float* part_s = (float*)(part_i + (size_t)bmu_code_tiles(K) * g.R);
    float* part_x2 = part_s + (size_t)bmu_code_tiles(K) * g.R;

    const int ptiles = (g.R + BN - 1) / BN;
    const int ctiles = bmu_code_tiles(K);
    int nsplit = (512 + ptiles - 1) / ptiles;
    if (nsplit > ctiles) nsplit = ctiles;
    if (nsplit < 1) nsplit = 1;
    const int per = (ctiles + nsplit - 1) / nsplit;
    nsplit = (ctiles + per - 1) / per;
    int64_t* direct = nsplit == 1 ? out_idx : (int64_t*)nullptr;
    // coarse bf16 pass + certificate + exact re-scan where it beats the exact kernels although every
    // workgroup stages the codebook itself (callers with a frozen codebook pass a prepared image to
    // qarig_bmu_fwd_coarse instead: qarig.ops.bmu)
    const int coarse_env = g_qarig_opt.bmu_coarse;
    if (bmu_coarse_ok(g, K, codebook) && coarse_env != 0 && (coarse_env == 1 || g.R >= 24576))
        return bmu_coarse_launch(g, codebook, K, out_idx, nullptr, prepared, st);
    if (D <= 16) {
        const int ks = D <= 4 ? 2 : (D <= 8 ? 4 : 8);
        const int bpc = (2 * ks + 1) * 4;                         // LDS bytes per code
        const int chunk_max = RES_MAX_LDS / bpc / 256 * 256;
        const int nchunks = (K + chunk_max - 1) / chunk_max;
        // waves sharing a row tile: as few as still give the chip two workgroups per CU
        int cs = 1;
        while (cs < 4 && (int64_t)((g.R + 128 / cs - 1) / (128 / cs)) * nchunks < 512) cs *= 2;
        if (const int v = g_qarig_opt.bmu_cs; v == 1 || v == 2 || v == 4) cs = v;
        const int unit = 64 * cs;                                 // two 32-code tiles per wave
        const int chunk = ((K + nchunks - 1) / nchunks + unit - 1) / unit * unit;
        const size_t shm = (size_t)chunk * bpc + (size_t)(3 * 128 + 128 + 128 + 512) * sizeof(float);
        PatchOffsets po;
        for (int e = 0; e < 16; ++e) {
            const int ee = e < D ? e : 0;
            const int j = ee % pW, i = (ee / pW) % pH, c = ee / (pW * pH);
            po.off[e] = (c * H + i) * W + j;
        }
        auto magic = [](int d, unsigned& m, int& sh) {
            int L = 0;
            while ((1LL << L) < d) ++L;
            sh = 31 + L;
            m = (unsigned)((1ULL << sh) / (unsigned long long)d + 1ULL);
        };
        magic(g.gh * g.gw, po.m_per, po.sh_per);
        magic(g.gw, po.m_gw, po.sh_gw);
        int64_t* direct_r = nchunks == 1 ? out_idx : (int64_t*)nullptr;
        dim3 grid((g.R + 128 / cs - 1) / (128 / cs), nchunks), block(NTHREADS);
        // group scan where a wave walks at least 4 tile pairs (its one-off re-evaluation of the winning
        // group costs about what it saves on 2); QARIG_BMU_GROUPS=0/1 overrides
        const int groups_env = g_qarig_opt.bmu_groups;
        // (D <= 8 only: with D = 16 the nine MFMAs per tile dominate and the group form measured no gain)
        const bool groups = groups_env >= 0 ? groups_env != 0 : (ks <= 4 && chunk / (32 * cs) >= 8);
        // (a four-tiles-per-step form -- four independent accumulate chains -- measured no gain and was removed)
#define QARIG_BMU_RES(KS_, CS_)                                                                    \
        do {                                                                                       \
            if (groups)                                                                            \
                hipLaunchKernelGGL((bmu_resident_kernel<KS_, CS_, true, 2>), grid, block, shm, st, g, po,     \
                                   codebook, K, chunk, part_d, part_i, part_s, part_x2, direct_r); \
            else                                                                                   \
                hipLaunchKernelGGL((bmu_resident_kernel<KS_, CS_, false, 2>), grid, block, shm, st, g, po,    \
                                   codebook, K, chunk, part_d, part_i, part_s, part_x2, direct_r); \
        } while (0)
#define QARIG_BMU_RES_CS(KS_)                                                                      \
        do {                                                                                       \
            if (cs == 1) QARIG_BMU_RES(KS_, 1);                                                    \
            else if (cs == 2) QARIG_BMU_RES(KS_, 2);                                               \
            else QARIG_BMU_RES(KS_, 4);                                                            \
        } while (0)
        if (ks == 2) QARIG_BMU_RES_CS(2);
        else if (ks == 4) QARIG_BMU_RES_CS(4);
        else QARIG_BMU_RES_CS(8);
#undef QARIG_BMU_RES_CS
#undef QARIG_BMU_RES
        QARIG_CHECK_LAUNCH("bmu resident");
        if (nchunks > 1) {
            hipLaunchKernelGGL(bmu_finalize_kernel, dim3((g.R + 127) / 128), dim3(256), 0, st, g, codebook,
                               K, part_d, part_i, part_s, part_x2, nchunks, out_idx);
            QARIG_CHECK_LAUNCH("bmu finalize");
        }
        return QARIG_OK;
    }
    if (D <= 64) {
        const int nkt = D <= 32 ? 2 : 4;
        const size_t shm = (size_t)3 * nkt * TILE_FLOATS * sizeof(float);
        dim3 grid(ptiles, nsplit), block(NTHREADS);
#define QARIG_BMU_SMALL(NKT_, KS_)                                                              \
        hipLaunchKernelGGL((bmu_small_kernel<NKT_, KS_>), grid, block, shm, st, g, codebook, K, per, \
                           part_d, part_i, part_s, part_x2, direct)
        if (D <= 32) QARIG_BMU_SMALL(2, 8);      // (D <= 16 never gets here: the resident / coarse kernels take it)
        else {
            static bool attr_set = false;   // > 64 KB of dynamic LDS needs the opt-in once
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)bmu_small_kernel<4, 8>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
                attr_set = true;
            }
            QARIG_BMU_SMALL(4, 8);
        }
#undef QARIG_BMU_SMALL
    } else {
        hipLaunchKernelGGL(bmu_mma_kernel, dim3(ptiles, nsplit), dim3(NTHREADS), 0, st, g, codebook, K,
                           per, part_d, part_i, part_s, part_x2, direct);
    }
    QARIG_CHECK_LAUNCH("bmu mma");
    if (nsplit > 1) {
        hipLaunchKernelGGL(bmu_finalize_kernel, dim3((g.R + 127) / 128), dim3(256), 0, st, g, codebook,
                           K, part_d, part_i, part_s, part_x2, nsplit, out_idx);
        QARIG_CHECK_LAUNCH("bmu finalize");
    }
    return QARIG_OK;
}
