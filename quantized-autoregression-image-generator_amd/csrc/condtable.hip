// Position-table form of the conditioning path.
//
// In the reference every decoder block projects the conditioning tensor `cond` (N,S,D)
// through its ScaleLayer/ShiftLayer Linear layers for EVERY token (models/layers.py:100-153,
// 258-304): 6 (base) or 9 (encoder-decoder) 512x512 GEMMs per layer over N*S rows, 20 % of the
// model's FLOPs.  But `cond` is a function of the token's integer position alone
// (Transformer.py:154-167: pos_cond -> sinusoid -> pos_cond_layer), and a training batch holds
// a few hundred distinct positions among its tens of thousands of tokens.  The host therefore
// evaluates pos_cond_layer and every projection ONCE PER DISTINCT POSITION (a (P,D) table, P =
// position bound) and the per-token consumers read their row of the table through an index:
//   * AdaLN modulate: qarig_layernorm_fwd/bwd with mod_idx (norm.hip);
//   * ResidualLinearLayer's x * scale(cond): qarig_mul_rows_fwd/bwd (here).
// Row r of a GEMM depends on row r of its input only, so the forward values are those of the
// per-token evaluation; in the backward pass the per-token gradients of a table row are summed
// in a fixed order (qarig_segment_sum over the row map built by qarig_rowmap_build):
// deterministic, no atomics.
#include "qarig_common.h"

namespace qarig {

// cnt[p] = number of tokens with idx == p; bad flag for indices outside [0, P).
__global__ __launch_bounds__(256) void rowmap_count_kernel(const int* __restrict__ idx, int M, int P,
                                                           int* __restrict__ cnt,
                                                           int* __restrict__ bad) {
    const int p = blockIdx.x;
    int c = 0;
    for (int r = threadIdx.x; r < M; r += 256) {
        const int v = idx[r];
        c += v == p;
        if (p == 0 && (v < 0 || v >= P)) atomicExch(bad, 1);
    }
    __shared__ int part[4];
    c = (int)wave_sum((float)c);     // counts <= 2^24: exact in fp32
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[p] = part[0] + part[1] + part[2] + part[3];
}

// offsets[0..P] = exclusive prefix sum of cnt[0..P-1]; one block.
__global__ __launch_bounds__(256) void rowmap_scan_kernel(const int* __restrict__ cnt, int P,
                                                          int* __restrict__ offsets) {
    __shared__ int carry;
    __shared__ int buf[256];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < P; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < P ? cnt[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {          // Hillis-Steele inclusive scan
            const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < P) offsets[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += buf[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[P] = carry;
}

// rows[offsets[p] ...] = the tokens with idx == p, ascending.  One wave per table row.
__global__ __launch_bounds__(64) void rowmap_fill_kernel(const int* __restrict__ idx, int M,
                                                         const int* __restrict__ offsets,
                                                         int* __restrict__ rows) {
    const int p = blockIdx.x, lane = threadIdx.x;
    int run = offsets[p];
    for (int base = 0; base < M; base += 64) {
        const int r = base + lane;
        const bool hit = r < M && idx[r] == p;
        const unsigned long long m = __ballot(hit);
        if (hit) rows[run + __popcll(m & ((1ull << lane) - 1ull))] = r;
        run += __popcll(m);
    }
}

// out[p][c] = sum over the tokens t of table row p of src[t][c].  The block's SG sub-groups
// take the row's tokens round-robin (sub-group s: entries s, s+SG, ...; two loads in flight
// each) and their partial sums are added in sub-group order: a fixed order, so the result is
// run-to-run identical, while 2*SG row reads are in flight instead of one.
constexpr int SEG_GROUPS = 4;
__global__ __launch_bounds__(512) void segment_sum_kernel(const float* __restrict__ src,
                                                          const int* __restrict__ offsets,
                                                          const int* __restrict__ rows, int D,
                                                          float* __restrict__ out) {
    __shared__ float4 part[SEG_GROUPS][128];
    const int p = blockIdx.x;
    const int beg = offsets[p], end = offsets[p + 1];
    const int sg = threadIdx.x >> 7, t = threadIdx.x & 127;
    for (int c0 = blockIdx.y * 512; c0 < D; c0 += gridDim.y * 512) {
        const int c = c0 + t * 4;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        if (c < D) {
            int e = beg + sg;
            for (; e + SEG_GROUPS < end; e += 2 * SEG_GROUPS) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)rows[e] * D + c);
                const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)rows[e + SEG_GROUPS] * D + c);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            }
            if (e < end) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)rows[e] * D + c);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            }
        }
        __syncthreads();
        part[sg][t] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
        __syncthreads();
        if (sg == 0 && c < D) {
            float4 r = part[0][t];
#pragma unroll
            for (int g = 1; g < SEG_GROUPS; ++g) {
                r.x += part[g][t].x; r.y += part[g][t].y; r.z += part[g][t].z; r.w += part[g][t].w;
            }
            *reinterpret_cast<float4*>(out + (int64_t)p * D + c) = r;
        }
    }
}

// y[r][c] = a[r][c] * tab[idx[r]][c]
__global__ void mul_rows_kernel(const float* __restrict__ a, const float* __restrict__ tab,
                                const int* __restrict__ idx, float* __restrict__ y, int M, int D4) {
    const int64_t total = (int64_t)M * D4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / D4), c4 = (int)(i - (int64_t)r * D4);
        const float4 av = reinterpret_cast<const float4*>(a)[i];
        const float4 bv = reinterpret_cast<const float4*>(tab)[(int64_t)idx[r] * D4 + c4];
        reinterpret_cast<float4*>(y)[i] = make_float4(av.x * bv.x, av.y * bv.y, av.z * bv.z, av.w * bv.w);
    }
}

// da = dy * tab[idx[r]],  db_tok = dy * a  (per token; the caller segment-sums it)
__global__ void mul_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                    const float* __restrict__ tab, const int* __restrict__ idx,
                                    float* __restrict__ da, float* __restrict__ db_tok, int M,
                                    int D4) {
    const int64_t total = (int64_t)M * D4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / D4), c4 = (int)(i - (int64_t)r * D4);
        const float4 d = reinterpret_cast<const float4*>(dy)[i];
        const float4 av = reinterpret_cast<const float4*>(a)[i];
        const float4 bv = reinterpret_cast<const float4*>(tab)[(int64_t)idx[r] * D4 + c4];
        reinterpret_cast<float4*>(da)[i] = make_float4(d.x * bv.x, d.y * bv.y, d.z * bv.z, d.w * bv.w);
        reinterpret_cast<float4*>(db_tok)[i] = make_float4(d.x * av.x, d.y * av.y, d.z * av.z, d.w * av.w);
    }
}

}  // namespace qarig

using namespace qarig;

static dim3 ct_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    return dim3((unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)));
}

// Row map of an index vector: offsets (P+1 ints) and rows (M ints) such that
// rows[offsets[p] .. offsets[p+1]) lists, ascending, the tokens r with idx[r] == p.
// counts: P ints of scratch.  *bad_flag (device int, caller-zeroed) is set if an index lies
// outside [0, P) (the host raises IndexError, as nn.Embedding would).
extern "C" int qarig_rowmap_build(const int* idx, int M, int P, int* counts, int* offsets, int* rows,
                                  int* bad_flag, void* stream) {
    QARIG_CHECK_ARG(idx && counts && offsets && rows && bad_flag && M > 0 && P > 0,
                    "rowmap_build: bad arguments");
    QARIG_CHECK_DIMS("rowmap_build", M);
    QARIG_CHECK_DIMS("rowmap_build", P);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rowmap_count_kernel, dim3(P), dim3(256), 0, st, idx, M, P, counts, bad_flag);
    QARIG_CHECK_LAUNCH("rowmap count");
    hipLaunchKernelGGL(rowmap_scan_kernel, dim3(1), dim3(256), 0, st, counts, P, offsets);
    QARIG_CHECK_LAUNCH("rowmap scan");
    hipLaunchKernelGGL(rowmap_fill_kernel, dim3(P), dim3(64), 0, st, idx, M, offsets, rows);
    QARIG_CHECK_LAUNCH("rowmap fill");
    return QARIG_OK;
}

// out (P,D) = per-table-row sums of src (M,D) in ascending token order.  D % 4 == 0, 16-B aligned.
extern "C" int qarig_segment_sum(const float* src, const int* offsets, const int* rows, int P, int D,
                                 float* out, void* stream) {
    QARIG_CHECK_ARG(src && offsets && rows && out && P > 0 && D > 0 && D % 4 == 0 &&
                        ((((uintptr_t)src | (uintptr_t)out)) & 15) == 0,
                    "segment_sum: bad arguments (D %% 4 == 0, 16-B aligned)");
    QARIG_CHECK_DIMS("segment_sum", P, D);
    hipLaunchKernelGGL(segment_sum_kernel, dim3(P, (D + 511) / 512), dim3(512), 0,
                       (hipStream_t)stream, src, offsets, rows, D, out);
    QARIG_CHECK_LAUNCH("segment_sum");
    return QARIG_OK;
}

// ResidualLinearLayer's x * scale(cond) with scale(cond) given as a position table
// (models/layers.py:293-295): y[r] = a[r] * tab[idx[r]].
extern "C" int qarig_mul_rows_fwd(const float* a, const float* tab, const int* idx, float* y, int M,
                                  int D, void* stream) {
    QARIG_CHECK_ARG(a && tab && idx && y && M > 0 && D > 0 && D % 4 == 0 &&
                        ((((uintptr_t)a | (uintptr_t)tab | (uintptr_t)y)) & 15) == 0,
                    "mul_rows_fwd: bad arguments (D %% 4 == 0, 16-B aligned)");
    QARIG_CHECK_DIMS("mul_rows_fwd", M, D);
    hipLaunchKernelGGL(mul_rows_kernel, ct_grid((int64_t)M * (D / 4)), dim3(256), 0,
                       (hipStream_t)stream, a, tab, idx, y, M, D / 4);
    QARIG_CHECK_LAUNCH("mul_rows_fwd");
    return QARIG_OK;
}

extern "C" int qarig_mul_rows_bwd(const float* dy, const float* a, const float* tab, const int* idx,
                                  float* da, float* db_tok, int M, int D, void* stream) {
    QARIG_CHECK_ARG(dy && a && tab && idx && da && db_tok && M > 0 && D > 0 && D % 4 == 0 &&
                        ((((uintptr_t)dy | (uintptr_t)a | (uintptr_t)tab | (uintptr_t)da |
                           (uintptr_t)db_tok)) & 15) == 0,
                    "mul_rows_bwd: bad arguments (D %% 4 == 0, 16-B aligned)");
    QARIG_CHECK_DIMS("mul_rows_bwd", M, D);
    hipLaunchKernelGGL(mul_rows_bwd_kernel, ct_grid((int64_t)M * (D / 4)), dim3(256), 0,
                       (hipStream_t)stream, dy, a, tab, idx, da, db_tok, M, D / 4);
    QARIG_CHECK_LAUNCH("mul_rows_bwd");
    return QARIG_OK;
}
