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
};

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
        // Wide epilogue: the wave's 64x64 tile goes through its private 8 KB of LDS in two
        // 32-row halves, so that every global access of the epilogue (C, preact, residual,
        // gradz, slabs) is a 16-B-per-lane, 256-B-per-row dwordx4 instead of 4x as many
        // 4-B accesses (the epilogue is store-issue bound otherwise).
        // 4 waves x 32 x 64 floats = 32 KB <= GEMM_LDS_FLOATS; unpadded rows are conflict-free
        // for both the b32 writes (half-waves hit different rows) and the b128 reads
        constexpr int EL = 64;
        float* stage = lds + wave * (32 * EL);
        const int er = lane >> 4, ec = (lane & 15) * 4;
        const int gc = n0 + wn * 64 + ec;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ep.bias && splitk == 1) bv = *reinterpret_cast<const float4*>(ep.bias + gc);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[acc_row(r, lane) * EL + j * 32 + cl] = acc.t[i][j][r];
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int lr = it * 4 + er;
                const int64_t row = m0 + wm * 64 + i * 32 + lr;
                float4 t = *reinterpret_cast<const float4*>(stage + lr * EL + ec);
                if (splitk > 1) {
                    *reinterpret_cast<float4*>(slabs + ((int64_t)blockIdx.z * M + row) * N + gc) = t;
                    continue;
                }
                t.x += bv.x; t.y += bv.y; t.z += bv.z; t.w += bv.w;
                if (ep.residual) {
                    const float4 rv = *reinterpret_cast<const float4*>(ep.residual + row * ep.ldr + gc);
                    t.x += rv.x; t.y += rv.y; t.z += rv.z; t.w += rv.w;
                }
                if (ep.preact) *reinterpret_cast<float4*>(ep.preact + row * ep.ldp + gc) = t;
                float4 y = make_float4(act_fwd(t.x, ep.act), act_fwd(t.y, ep.act),
                                       act_fwd(t.z, ep.act), act_fwd(t.w, ep.act));
                if (ep.gradz) {
                    const float4 z = *reinterpret_cast<const float4*>(ep.gradz + row * ep.ldz + gc);
                    y.x *= act_grad(z.x, ep.gact); y.y *= act_grad(z.y, ep.gact);
                    y.z *= act_grad(z.z, ep.gact); y.w *= act_grad(z.w, ep.gact);
                }
                *reinterpret_cast<float4*>(ep.C + row * ep.ldc + gc) = y;
            }
            __syncthreads();
        }
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
constexpr int COLSUM_ROWS = 512;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X,
                                                             int64_t ldx, int M, int N,
                                                             float* __restrict__ part) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * COLSUM_ROWS;
    const int r1 = min(M, r0 + COLSUM_ROWS);
    float s = 0.0f;
    if (col < N)
        for (int r = r0 + ry; r < r1; r += 4) s += X[(int64_t)r * ldx + col];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < N)
        part[(int64_t)blockIdx.y * N + col] = ((red[0][cx] + red[1][cx]) + red[2][cx]) + red[3][cx];
}

}  // namespace qarig

using namespace qarig;

extern "C" size_t qarig_gemm_workspace_bytes(int M, int N, int splitk) {
    // split-K slabs + (always) room for the per-split A row sums
    const size_t sk = splitk > 1 ? splitk : 1;
    return (splitk > 1 ? sk * M * N * sizeof(float) : 0) + sk * M * sizeof(float);
}

extern "C" int qarig_gemm_f32(const float* A, int64_t lda, int a_kcontig, const float* B,
                              int64_t ldb, int b_kcontig, float* C, int64_t ldc, int M, int N,
                              int K, const float* bias, const float* residual, int64_t ldr,
                              float* preact, int64_t ldp, int act, const float* gradz,
                              int64_t ldz, int gact, int splitk, int accumulate, float* a_rowsum,
                              void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(A && B && C, "gemm: null operand");
    QARIG_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad extents M=%d N=%d K=%d", M, N, K);
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
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, splitk), block(NTHREADS);
    hipStream_t st = (hipStream_t)stream;
    float* slabs = (float*)workspace;
    // per-split A row sums live behind the slabs
    float* rs_part = a_rowsum ? slabs + (splitk > 1 ? (size_t)splitk * M * N : 0) : nullptr;
    GemmEpilogue ep{C, ldc, bias, residual, ldr, preact, ldp, act, gradz, ldz, gact, rs_part};
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool va = al16(A) && lda % 4 == 0, vb = al16(B) && ldb % 4 == 0;
    int per = K;
    if (splitk > 1) per = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
    const bool fast = va && vb && M % BM == 0 && N % BN == 0 && K % BK == 0 &&
                      (splitk == 1 || K % per == 0);
    auto ok4 = [&](const void* p, int64_t ld) { return !p || (al16(p) && ld % 4 == 0); };
    const int vec_epi = ok4(C, ldc) && ok4(bias, 4) && ok4(residual, ldr) && ok4(preact, ldp) &&
                        ok4(gradz, ldz) && ok4(slabs, 4);
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
    QARIG_CHECK_LAUNCH("gemm");
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
    const int chunks = (M + COLSUM_ROWS - 1) / COLSUM_ROWS;
    return (size_t)chunks * N * sizeof(float);
}

// out[N] = sum over rows of X[M][N]; fixed summation order (bit-reproducible).
extern "C" int qarig_colsum_f32(const float* X, int64_t ldx, int M, int N, float* out,
                                int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(X && out && M > 0 && N > 0, "colsum: bad arguments");
    if (!workspace || ws_bytes < qarig_colsum_workspace_bytes(M, N)) {
        qarig_set_error("colsum: workspace too small");
        return QARIG_ERR_WORKSPACE;
    }
    const int chunks = (M + COLSUM_ROWS - 1) / COLSUM_ROWS;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, st, X, ldx,
                       M, N, part);
    QARIG_CHECK_LAUNCH("colsum partial");
    int blocks = (N + 255) / 256;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, part, out, (int64_t)N, 1,
                       N, chunks, accumulate);
    QARIG_CHECK_LAUNCH("colsum reduce");
    return QARIG_OK;
}
