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
//   acc   = w2[k]; for e ascending: acc = fmaf(-2*w[k][e], x[r][e], acc)
//   d     = sqrtf(max(acc + x2[r], 0));  index = first k with minimal d.
// The sqrt is only evaluated when a candidate beats the running squared
// distance; since sqrtf is monotone that gives the same first-minimum of d.
#include <limits.h>

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

// |w_k|^2 per codeword, sequential fma chain.
__global__ void bmu_code_norm_kernel(const float* __restrict__ w, int K, int D,
                                     float* __restrict__ w2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float* p = w + (int64_t)k * D;
    float acc = 0.0f;
    for (int e = 0; e < D; ++e) acc = fmaf(p[e], p[e], acc);
    w2[k] = acc;
}

// |x_r|^2 per patch row, sequential fma chain over the patch order.
__global__ void bmu_patch_norm_kernel(PatchGeom g, float* __restrict__ x2) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= g.R) return;
    const int64_t base = patch_row_base(g, row);
    float acc = 0.0f;
    for (int c = 0; c < g.C; ++c)
        for (int i = 0; i < g.pH; ++i) {
            const float* p = g.x + base + ((int64_t)c * g.H + i) * g.W;
            for (int j = 0; j < g.pW; ++j) acc = fmaf(p[j], p[j], acc);
        }
    x2[row] = acc;
}

__global__ __launch_bounds__(NTHREADS, 2) void bmu_mma_kernel(PatchGeom g,
                                                              const float* __restrict__ w, int K,
                                                              const float* __restrict__ w2,
                                                              const float* __restrict__ x2,
                                                              int fused_norms, int tiles_per_split,
                                                              float* __restrict__ part_s,
                                                              int* __restrict__ part_i,
                                                              int64_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[GEMM_LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, cl = lane & 31;
    const int p0 = blockIdx.x * BN;  // first patch row of this block

    SrcKContig sa{w, (int64_t)g.D, K, g.D, -2.0f,
                  (((uintptr_t)w & 15) == 0) && (g.D % 4 == 0)};
    SrcPatch sb;
    sb.g = g;
    sb.init(p0, tid);

    float best_d2[2] = {INFINITY, INFINITY};
    float best_s[2] = {INFINITY, INFINITY};
    int best_i[2] = {INT_MAX, INT_MAX};
    float x2v[2];
    __shared__ float norms[128];   // |x|^2 of the block's rows, then |w|^2 of each code tile
    if (fused_norms) {
        // small D: the norm chains (same sequential fma order as the stand-alone kernels)
        // are computed here, saving two launches and their boundaries
        if (tid < 128) {
            float acc = 0.0f;
            if (sb.valid)
                for (int c = 0; c < g.C; ++c)
                    for (int i = 0; i < g.pH; ++i) {
                        const float* p = g.x + sb.rowbase + ((int64_t)c * g.H + i) * g.W;
                        for (int j = 0; j < g.pW; ++j) acc = fmaf(p[j], p[j], acc);
                    }
            norms[tid] = acc;
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int prow = p0 + wn * 64 + j * 32 + cl;
        x2v[j] = fused_norms ? norms[wn * 64 + j * 32 + cl] : (prow < g.R ? x2[prow] : 0.0f);
    }

    const int code_tiles = (K + BM - 1) / BM;
    const int ct0 = blockIdx.y * tiles_per_split;
    const int ct1 = min(code_tiles, ct0 + tiles_per_split);
    for (int ct = ct0; ct < ct1; ++ct) {
        const int c0 = ct * BM;
        if (fused_norms) {
            __syncthreads();           // previous users of norms[] are done
            if (tid < 128) {
                float a2 = 0.0f;
                if (c0 + tid < K) {
                    const float* p = w + (int64_t)(c0 + tid) * g.D;
                    for (int e = 0; e < g.D; ++e) a2 = fmaf(p[e], p[e], a2);
                }
                norms[tid] = a2;
            }
            __syncthreads();
        }
        Acc acc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lc = wm * 64 + i * 32 + acc_row(r, lane);
                const float v = fused_norms ? norms[lc] : (c0 + lc < K ? w2[c0 + lc] : 0.0f);
                acc.t[i][0][r] = v;
                acc.t[i][1][r] = v;
            }
        contract(acc, sa, sb, c0, p0, 0, g.D, lds);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int code = c0 + wm * 64 + i * 32 + acc_row(r, lane);
                    const float d2 = fmaxf(acc.t[i][j][r] + x2v[j], 0.0f);
                    if (code < K && d2 < best_d2[j]) {
                        const float s = sqrtf(d2);
                        if (s < best_s[j]) { best_s[j] = s; best_i[j] = code; }
                        best_d2[j] = d2;
                    }
                }
    }

    // Combine the 4 holders of each patch column: lane halves (h) x waves (wm).
    float* cs = lds;                                   // [4][128]
    int* ci = reinterpret_cast<int*>(lds + 4 * 128);   // [4][128]
    const int slot = wm * 2 + (lane >> 5);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + cl;
        cs[slot * 128 + col] = best_s[j];
        ci[slot * 128 + col] = best_i[j];
    }
    __syncthreads();
    if (tid < 128) {
        const int prow = p0 + tid;
        if (prow < g.R) {
            float s = cs[tid];
            int idx = ci[tid];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const float s2 = cs[q * 128 + tid];
                const int i2 = ci[q * 128 + tid];
                if (s2 < s || (s2 == s && i2 < idx)) { s = s2; idx = i2; }
            }
            if (out) {
                out[prow] = idx == INT_MAX ? 0 : (int64_t)idx;
            } else {
                part_s[(int64_t)blockIdx.y * g.R + prow] = s;
                part_i[(int64_t)blockIdx.y * g.R + prow] = idx;
            }
        }
    }
}

__global__ void bmu_finalize_kernel(const float* __restrict__ part_s, const int* __restrict__ part_i,
                                    int R, int nsplit, int64_t* __restrict__ out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    float s = part_s[row];
    int idx = part_i[row];
    for (int z = 1; z < nsplit; ++z) {
        const float s2 = part_s[(int64_t)z * R + row];
        const int i2 = part_i[(int64_t)z * R + row];
        if (s2 < s || (s2 == s && i2 < idx)) { s = s2; idx = i2; }
    }
    out[row] = idx == INT_MAX ? 0 : (int64_t)idx;
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

}  // namespace qarig

using namespace qarig;

static int bmu_code_tiles(int K) { return (K + BM - 1) / BM; }

extern "C" size_t qarig_bmu_workspace_bytes(int64_t rows, int K) {
    // w2[K] + x2[R] + per-split (float, int) partials for up to code_tiles splits.
    const size_t R = (size_t)rows;
    return sizeof(float) * ((size_t)K + R) + (size_t)bmu_code_tiles(K) * R * 8 + 64;
}

extern "C" int qarig_bmu_fwd(const float* x, int N, int C, int H, int W, int pH, int pW,
                             const float* codebook, int K, int D, int64_t* out_idx,
                             void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(x && codebook && out_idx, "bmu: null pointer");
    QARIG_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && pH > 0 && pW > 0 && K > 0,
                    "bmu: bad extents");
    QARIG_CHECK_ARG(pH <= H && pW <= W, "bmu: patch larger than the latent");
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
    float* w2 = (float*)workspace;
    float* x2 = w2 + K;
    float* part_s = x2 + g.R;
    int* part_i = (int*)(part_s + (size_t)bmu_code_tiles(K) * g.R);

    const int fused_norms = D <= 256;
    if (!fused_norms) {
        hipLaunchKernelGGL(bmu_code_norm_kernel, dim3((K + 63) / 64), dim3(64), 0, st, codebook, K, D,
                           w2);
        QARIG_CHECK_LAUNCH("bmu code norm");
        hipLaunchKernelGGL(bmu_patch_norm_kernel, dim3((g.R + 63) / 64), dim3(64), 0, st, g, x2);
        QARIG_CHECK_LAUNCH("bmu patch norm");
    }

    const int ptiles = (g.R + BN - 1) / BN;
    const int ctiles = bmu_code_tiles(K);
    int nsplit = (512 + ptiles - 1) / ptiles;
    if (nsplit > ctiles) nsplit = ctiles;
    if (nsplit < 1) nsplit = 1;
    const int per = (ctiles + nsplit - 1) / nsplit;
    nsplit = (ctiles + per - 1) / per;
    hipLaunchKernelGGL(bmu_mma_kernel, dim3(ptiles, nsplit), dim3(NTHREADS), 0, st, g, codebook, K,
                       w2, x2, fused_norms, per, part_s, part_i,
                       nsplit == 1 ? out_idx : (int64_t*)nullptr);
    QARIG_CHECK_LAUNCH("bmu mma");
    if (nsplit > 1) {
        hipLaunchKernelGGL(bmu_finalize_kernel, dim3((g.R + 255) / 256), dim3(256), 0, st, part_s,
                           part_i, g.R, nsplit, out_idx);
        QARIG_CHECK_LAUNCH("bmu finalize");
    }
    return QARIG_OK;
}
