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
// fmaf(-2 w_e, x_e, acc) over ascending e is bit-identical): 8 x 8 pairs per 64-lane workgroup,
// operands through LDS in 128-element chunks, every lane carrying its own |w|^2 and |x|^2
// chains (the first pair row / column publishes them).  acc (rows x K), w2 (K) and x2 (rows) go to the workspace; the
// second kernel takes sqrt(max((acc + w2) + x2, 0)) and the first minimum per row.
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

extern "C" int qarig_bmu_fwd(const float* x, int N, int C, int H, int W, int pH, int pW,
                             const float* codebook, int K, int D, int64_t* out_idx,
                             void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(x && codebook && out_idx, "bmu: null pointer");
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
    static const bool resident_on = []() { const char* e = getenv("QARIG_BMU_RESIDENT"); return !(e && e[0] == '0'); }();
    if (D <= 16 && resident_on) {
        const int ks = D <= 4 ? 2 : (D <= 8 ? 4 : 8);
        const int bpc = (2 * ks + 1) * 4;                         // LDS bytes per code
        const int chunk_max = RES_MAX_LDS / bpc / 256 * 256;
        const int nchunks = (K + chunk_max - 1) / chunk_max;
        // waves sharing a row tile: as few as still give the chip two workgroups per CU
        int cs = 1;
        while (cs < 4 && (int64_t)((g.R + 128 / cs - 1) / (128 / cs)) * nchunks < 512) cs *= 2;
        if (const char* e = getenv("QARIG_BMU_CS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) cs = v; }
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
        static const int groups_env = []() { const char* e = getenv("QARIG_BMU_GROUPS"); return e ? atoi(e) : -1; }();
        // (D <= 8 only: with D = 16 the nine MFMAs per tile dominate and the group form measured no gain)
        const bool groups = groups_env >= 0 ? groups_env != 0 : (ks <= 4 && chunk / (32 * cs) >= 8);
        // four tiles per step (four independent accumulate chains) where the wave's tile count allows
        static const int quad_env = []() { const char* e = getenv("QARIG_BMU_QUADS"); return e ? atoi(e) : -1; }();
        const bool quads = groups && (chunk / (32 * cs)) % 4 == 0 && quad_env == 1;   // measured: no gain; opt-in
#define QARIG_BMU_RES(KS_, CS_)                                                                    \
        do {                                                                                       \
            if (quads)                                                                             \
                hipLaunchKernelGGL((bmu_resident_kernel<KS_, CS_, true, 4>), grid, block, shm, st, g, po,     \
                                   codebook, K, chunk, part_d, part_i, part_s, part_x2, direct_r); \
            else if (groups)                                                                       \
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
        const int nkt = D <= 16 ? 1 : (D <= 32 ? 2 : 4);
        const size_t shm = (size_t)3 * nkt * TILE_FLOATS * sizeof(float);
        dim3 grid(ptiles, nsplit), block(NTHREADS);
#define QARIG_BMU_SMALL(NKT_, KS_)                                                              \
        hipLaunchKernelGGL((bmu_small_kernel<NKT_, KS_>), grid, block, shm, st, g, codebook, K, per, \
                           part_d, part_i, part_s, part_x2, direct)
        if (D <= 4) QARIG_BMU_SMALL(1, 2);
        else if (D <= 8) QARIG_BMU_SMALL(1, 4);
        else if (D <= 16) QARIG_BMU_SMALL(1, 8);
        else if (D <= 32) QARIG_BMU_SMALL(2, 8);
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
