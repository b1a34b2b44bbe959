// Token embedding + sinusoidal positions (reference models/Transformer.py:127-167,
// models/layers.py:83-96) and the embedding-weight gradient.
#include "qarig_common.h"

namespace qarig {

// out[r][c] = c < half ? sin(pos[r]*freq[c]) : cos(pos[r]*freq[c-half]).
// freq (half floats) is computed by the host exactly as the reference does
// (torch.exp(arange(half) * -ln(1e4)/(half-1)) in fp32) so that the angle is
// bit-identical; sinf/cosf are the full-range ocml versions.
__global__ void posemb_kernel(const float* __restrict__ pos, int R, int D,
                              const float* __restrict__ freq, float* __restrict__ out) {
    const int half = D >> 1;
    const int64_t total = (int64_t)R * D;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / D), c = (int)(idx - (int64_t)r * D);
        const float p = pos[r];
        out[idx] = c < half ? sinf(p * freq[c]) : cosf(p * freq[c - half]);
    }
}

// out[m][:] = table[ids[m]][:] + pe[m % S][:]   (pe may be null)
__global__ void embedding_fwd_kernel(const int64_t* __restrict__ ids, int M, int S, int D, int V,
                                     const float* __restrict__ table,
                                     const float* __restrict__ pe, float* __restrict__ out,
                                     int* __restrict__ bad) {
    const int64_t total = (int64_t)M * D;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / D), c = (int)(idx - (int64_t)m * D);
        const int64_t id = ids[m];
        if (id < 0 || id >= V) {
            if (c == 0) atomicExch(bad, 1);
            out[idx] = 0.0f;
            continue;
        }
        float t = table[id * D + c];
        if (pe) t += pe[(int64_t)(m % S) * D + c];
        out[idx] = t;
    }
}

// Token assembly of the training loop (reference train_quantized_transformer.py:423-484) in one
// launch: the decoder input / target sequences
//   base   : input = [lr tokens | hr + k_lr],  target = [hr | <end> = k_hr]
//   enc-dec: input = [<start> = k_hr | hr],    target = [hr | <end>]
// are never materialised; window w of sample n starts at offs[n] and the kernel writes
// hr_in (N,W), hr_tg (N,W) and the absolute positions pos (N,W) = offs[n] + w directly.
// window == 0: no sliding window, the whole sequences (W = S_in) and no positions.
__global__ void assemble_tokens_kernel(const int64_t* __restrict__ lr, int S_lr,
                                       const int64_t* __restrict__ hr, int S_hr, int N, int base,
                                       int k_lr, int k_hr, const int64_t* __restrict__ offs, int W,
                                       int64_t* __restrict__ hr_in, int64_t* __restrict__ hr_tg,
                                       int64_t* __restrict__ pos, int* __restrict__ bad_flag) {
    const int lead = base ? S_lr : 1;          // tokens in front of the HR tokens of the input
    const int64_t total = (int64_t)N * W;
    const int64_t o_max = (int64_t)lead + S_hr - W;   // last valid window start
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / W), w = (int)(idx - (int64_t)n * W);
        int64_t o = offs ? offs[n] : 0;
        if (o < 0 || o > o_max) {                                  // a window outside the sequence: flag it
            if (bad_flag) *bad_flag = 1;                           // (the host raises IndexError), read a
            o = o < 0 ? 0 : o_max;                                 // clamped window instead of foreign memory
        }
        const int64_t j = o + w;                                   // index in the full sequences
        int64_t vin;
        if (j < lead) vin = base ? lr[(int64_t)n * S_lr + j] : (int64_t)k_hr;
        else vin = hr[(int64_t)n * S_hr + (j - lead)] + (base ? k_lr : 0);
        hr_in[idx] = vin;
        hr_tg[idx] = j < S_hr ? hr[(int64_t)n * S_hr + j] : (int64_t)k_hr;
        if (pos) pos[idx] = j;
    }
}

// dtable[v][:] = sum over m with ids[m]==v of dy[m][:], m ascending (deterministic, no
// atomics).  One block per vocabulary row; every wave scans the id stream 64 ids at a
// time (coalesced) and walks only the set bits of the match ballot.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, int M,
                                                            int D, const float* __restrict__ dy,
                                                            float* __restrict__ dtable) {
    const int v = blockIdx.x;
    const int lane = threadIdx.x & 63;
    for (int c0 = 0; c0 < D; c0 += 256) {
        const int c = c0 + threadIdx.x;
        float acc = 0.0f;
        for (int base = 0; base < M; base += 64) {
            const int m = base + lane;
            unsigned long long hit = __ballot(m < M && ids[m] == v);
            while (hit) {
                const int b = __ffsll((long long)hit) - 1;
                hit &= hit - 1;
                if (c < D) acc += dy[(int64_t)(base + b) * D + c];
            }
        }
        if (c < D) dtable[(int64_t)v * D + c] = acc;
    }
}

// Narrow tables (D <= 64: codebook rows): one workgroup per CPW consecutive table rows, its four
// waves scan one quarter of the id stream each -- 8 x 64 ids in flight per wave, every 64 ids
// answer CPW ballots -- and the four partial sums are added in stream order (fixed: deterministic).
template <int CPW>
__global__ __launch_bounds__(256) void embedding_bwd_narrow_kernel(const int64_t* __restrict__ ids, int M,
                                                                   int D, int V,
                                                                   const float* __restrict__ dy,
                                                                   float* __restrict__ dtable) {
    __shared__ float part[4][CPW][64];
    const int v0 = blockIdx.x * CPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // quarters in units of 64 ids
    const int groups = (M + 63) / 64;
    const int g_per = (groups + 3) / 4;
    const int g_begin = wave * g_per, g_end = min(groups, g_begin + g_per);
    float acc[CPW];
#pragma unroll
    for (int s = 0; s < CPW; ++s) acc[s] = 0.0f;
    constexpr int U = 8;
    for (int g0 = g_begin; g0 < g_end; g0 += U) {
        int64_t id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = (g0 + u) * 64 + lane;
            id[u] = (g0 + u < g_end && m < M) ? ids[m] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rel = (int)(id[u] - v0);      // ids outside the table never match (v0 + s < V)
            const bool mine = id[u] >= v0 && id[u] < v0 + CPW;
            if (!__any(mine)) continue;
#pragma unroll
            for (int s = 0; s < CPW; ++s) {
                unsigned long long hit = __ballot(mine && rel == s);
                while (hit) {
                    const int b = __ffsll((long long)hit) - 1;
                    hit &= hit - 1;
                    if (lane < D) acc[s] += dy[(int64_t)((g0 + u) * 64 + b) * D + lane];
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < CPW; ++s) part[wave][s][lane] = acc[s];
    __syncthreads();
    for (int i = threadIdx.x; i < CPW * 64; i += 256) {
        const int s = i >> 6, c = i & 63;
        if (c < D && v0 + s < V)
            dtable[(int64_t)(v0 + s) * D + c] = ((part[0][s][c] + part[1][s][c]) + part[2][s][c]) + part[3][s][c];
    }
}

}  // namespace qarig

using namespace qarig;

static int ew_blocks(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// get_positional_embeddings, models/layers.py:83-96.  pos: fp32 (R,), freq: (D/2,).
extern "C" int qarig_posemb_fwd(const float* pos, int R, int D, const float* freq, float* out,
                                void* stream) {
    QARIG_CHECK_ARG(pos && freq && out && R > 0 && D > 0 && (D % 2) == 0, "posemb: bad arguments");
    QARIG_CHECK_DIMS("posemb", R, D);
    hipLaunchKernelGGL(posemb_kernel, dim3(ew_blocks((int64_t)R * D)), dim3(256), 0,
                       (hipStream_t)stream, pos, R, D, freq, out);
    QARIG_CHECK_LAUNCH("posemb");
    return QARIG_OK;
}

// nn.Embedding lookup fused with the additive position table (Transformer.py:127-139,
// 154-167).  ids: int64 (M,), M = N*S; pe: (S,D) or NULL.  *bad_flag (device int,
// caller-zeroed) is set to 1 if an id is out of range (the host raises IndexError).
extern "C" int qarig_embedding_fwd(const int64_t* ids, int M, int S, int D, int V,
                                   const float* table, const float* pe, float* out, int* bad_flag,
                                   void* stream) {
    QARIG_CHECK_ARG(ids && table && out && bad_flag && M > 0 && S > 0 && D > 0 && V > 0,
                    "embedding_fwd: bad arguments");
    QARIG_CHECK_DIMS("embedding_fwd", M, D);
    QARIG_CHECK_DIMS("embedding_fwd", V, D);
    QARIG_CHECK_ARG(S <= (1 << 24), "embedding_fwd: bad arguments");
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ew_blocks((int64_t)M * D)), dim3(256), 0,
                       (hipStream_t)stream, ids, M, S, D, V, table, pe, out, bad_flag);
    QARIG_CHECK_LAUNCH("embedding_fwd");
    return QARIG_OK;
}

extern "C" int qarig_embedding_bwd(const int64_t* ids, int M, int D, int V, const float* dy,
                                   float* dtable, void* stream) {
    QARIG_CHECK_ARG(ids && dy && dtable && M > 0 && D > 0 && V > 0, "embedding_bwd: bad arguments");
    QARIG_CHECK_DIMS("embedding_bwd", M, D);
    QARIG_CHECK_DIMS("embedding_bwd", V, D);
    if (D <= 64 && V >= 4096)
        hipLaunchKernelGGL(embedding_bwd_narrow_kernel<8>, dim3((V + 7) / 8), dim3(256), 0, (hipStream_t)stream,
                           ids, M, D, V, dy, dtable);
    else if (D <= 64)
        hipLaunchKernelGGL(embedding_bwd_narrow_kernel<1>, dim3(V), dim3(256), 0, (hipStream_t)stream, ids, M,
                           D, V, dy, dtable);
    else
        hipLaunchKernelGGL(embedding_bwd_kernel, dim3(V), dim3(256), 0, (hipStream_t)stream, ids, M, D,
                           dy, dtable);
    QARIG_CHECK_LAUNCH("embedding_bwd");
    return QARIG_OK;
}

// hr_in / hr_tg / pos (N,W) int64 from the BMU indices (see assemble_tokens_kernel).  lr (N,S_lr)
// is read by the base model only; offs (N) int64 window starts or NULL (then W must be the
// input length S_lr + S_hr (base) / 1 + S_hr and pos may be NULL).  An offset outside [0, S_in - W]
// sets *bad_flag (device int, may be NULL) and is clamped, as the other index-consuming kernels do.
extern "C" int qarig_assemble_tokens(const int64_t* lr, int S_lr, const int64_t* hr, int S_hr, int N,
                                     int base, int k_lr, int k_hr, const int64_t* offs, int W,
                                     int64_t* hr_in, int64_t* hr_tg, int64_t* pos, int* bad_flag,
                                     void* stream) {
    QARIG_CHECK_ARG(hr && hr_in && hr_tg && (lr || !base), "assemble_tokens: null pointer");
    QARIG_CHECK_ARG(N > 0 && S_hr > 0 && W > 0 && (!base || S_lr > 0), "assemble_tokens: bad extents");
    QARIG_CHECK_DIMS("assemble_tokens", N, S_hr);
    QARIG_CHECK_DIMS("assemble_tokens", N, W);
    QARIG_CHECK_ARG(base ? (S_lr <= (1 << 24)) : 1, "assemble_tokens: bad extents");
    const int s_in = (base ? S_lr : 1) + S_hr;
    QARIG_CHECK_ARG(W <= s_in && (offs || W == s_in), "assemble_tokens: window %d vs sequence %d", W, s_in);
    // the target has S_hr + 1 entries; with the base model's S_lr leading tokens logits and
    // targets line up only for S_lr == 1 -- as in the reference, longer LR
    // sequences simply run out of target (guarded: <end> is written past it)
    const int64_t total = (int64_t)N * W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, lr, S_lr,
                       hr, S_hr, N, base, k_lr, k_hr, offs, W, hr_in, hr_tg, pos, bad_flag);
    QARIG_CHECK_LAUNCH("assemble_tokens");
    return QARIG_OK;
}
