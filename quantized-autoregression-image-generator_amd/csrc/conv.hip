// Convolutions of the autoencoder (reference models/layers.py:157-230 ConvLayer,
// DownsampleConvLayer, UpsampleConvLayer; models/FC_Encoder.py, FC_Decoder.py) as
// implicit GEMM on the shared fp32-MFMA contraction core, NCHW in and out.
//
//   out[n][co][oy][ox] = act(bias[co] + sum_{ci,ty,tx} W'[co][(ci,ty,tx)] *
//                            x[n][ci][oy*stride + offy[ty]][ox*stride + offx[tx]])
//
// The GEMM is laid out "weights x pixels": A = W' (Cout x K, reduction-contiguous),
// B = im2col gathered on the fly (pixels x K), so an accumulator register row is one
// output channel and its 32 lanes are 32 consecutive pixels -> coalesced NCHW stores.
//  * Conv2d 3x3 (stride 1 or 2, pad 1): W' is the weight tensor as stored.
//  * ConvTranspose2d 4x4 stride 2 pad 1: four output-parity classes, each a 2x2-tap
//    stride-1 conv over the input grid (K = 4*Cin, no multiplications by zero); the
//    class weights are packed by a small kernel per call.
//  * Cout <= 8 (the 256->3 / 512->4 output layers): a direct VALU kernel, one lane per
//    pixel, weights through the scalar cache -- an MFMA tile would be >90 % padding.
#include "qarig_common.h"
#include "ring_common.h"

namespace qarig {

struct ConvGeom {
    const float* x;
    int N, C, H, W;        // input tensor
    int Ho, Wo;            // logical output grid of this launch
    int stride;            // input step per logical output step
    int nty, ntx;          // taps
    int oy0, oys, ox0, oxs;  // input offset of tap t: o0 + os*t (os = +-1).  Affine on purpose:
                             // a per-tap table indexed at run time lives in scratch memory
    int K;                 // C * nty * ntx
    int P;                 // N * Ho * Wo logical output pixels
};

struct ConvOut {
    float* y;              // (N, Cout, HoP, WoP) physical output
    float* preact;         // same shape or null
    const float* bias;     // [Cout] or null
    int Cout, HoP, WoP;
    int os, py, px;        // physical = logical * os + (py, px)
    int act;
};

// B-side loader: x index = logical output pixel, k index = (ci, ty, tx).
// 32-bit offsets inside one image (host checks C*H*W < 2^31); the per-thread image base is a
// pointer computed once.
struct SrcIm2col {
    ConvGeom g;
    const float* xb;   // x + n * C*H*W
    int iy0, ix0;      // oy*stride + oy0, ox*stride + ox0
    bool valid;

    __device__ __forceinline__ void init(int x0, int tid) {
        const int p = x0 + (tid & 127);
        valid = p < g.P;
        const int per = g.Ho * g.Wo;
        const int n = valid ? p / per : 0;
        const int rem = valid ? p - n * per : 0;
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        xb = g.x + (int64_t)n * g.C * g.H * g.W;
        iy0 = oy * g.stride + g.oy0;
        ix0 = ox * g.stride + g.ox0;
    }
    __device__ __forceinline__ bool interior(int, int, int) const { return false; }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        load(r, x0, k0, tid);
    }
    __device__ __forceinline__ void load(float (&r)[STAGE], int, int k0, int tid) const {
        const int k = k0 + (tid >> 7) * 8;
        int tx = k % g.ntx;
        const int t = k / g.ntx;
        int ty = t % g.nty;
        int ci = t / g.nty;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int iy = iy0 + g.oys * ty, ix = ix0 + g.oxs * tx;
            const bool ok = valid && k + q < g.K && (unsigned)iy < (unsigned)g.H &&
                            (unsigned)ix < (unsigned)g.W;
            const int off = (ci * g.H + iy) * g.W + ix;
            r[q] = ok ? xb[ok ? off : 0] : 0.0f;
            if (++tx == g.ntx) {
                tx = 0;
                if (++ty == g.nty) { ty = 0; ++ci; }
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

// Row-vector form of the gather for launches whose input walk is unit-stride along x
// (Conv2d stride 1 and every ConvTranspose parity class) on output rows that are a multiple
// of 4 wide: a thread owns ONE tap k and FOUR consecutive pixels, which are four consecutive
// input floats -- one (4-B aligned) global_load_dwordx4 and one ds_write_b128 where the
// per-pixel form issues four of each, and a quarter of the index arithmetic.  Only the first
// and last group of an output row can touch the padding and take the per-element path.
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };

struct SrcIm2colRow {
    ConvGeom g;
    const float* xb;
    int iy0, ix0;
    uint32_t m_ntx, m_nty;   // ceil(2^32 / d): k / d == umulhi(k, m) for the k < 2^16 met here
    bool valid;

    __device__ __forceinline__ void init(int x0, int tid) {
        const int p = x0 + (tid & 31) * 4;          // P % 4 == 0: a group is valid as a whole
        valid = p < g.P;
        const int per = g.Ho * g.Wo;
        const int n = valid ? p / per : 0;
        const int rem = valid ? p - n * per : 0;
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        xb = g.x + (int64_t)n * g.C * g.H * g.W;
        iy0 = oy + g.oy0;
        ix0 = ox + g.ox0;
        m_ntx = g.ntx > 1 ? 0xFFFFFFFFu / (uint32_t)g.ntx + 1u : 0u;
        m_nty = g.nty > 1 ? 0xFFFFFFFFu / (uint32_t)g.nty + 1u : 0u;
    }
    __device__ __forceinline__ bool interior(int, int, int) const { return false; }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        load(r, x0, k0, tid);
    }
    __device__ __forceinline__ void load(float (&r)[STAGE], int, int k0, int tid) const {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + (tid >> 5) + 8 * h;
            const int t = g.ntx > 1 ? (int)__umulhi((uint32_t)k, m_ntx) : k;
            const int tx = k - t * g.ntx;
            const int ci = g.nty > 1 ? (int)__umulhi((uint32_t)t, m_nty) : t;
            const int ty = t - ci * g.nty;
            const int iy = iy0 + g.oys * ty, ix = ix0 + g.oxs * tx;
            const bool rowok = valid && k < g.K && (unsigned)iy < (unsigned)g.H;
            const int off = (ci * g.H + iy) * g.W + ix;
            if (rowok && ix >= 0 && ix + 3 < g.W) {
                const F4u v = *reinterpret_cast<const F4u*>(xb + off);
                r[4 * h] = v.x; r[4 * h + 1] = v.y; r[4 * h + 2] = v.z; r[4 * h + 3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = rowok && (unsigned)(ix + j) < (unsigned)g.W;
                    r[4 * h + j] = ok ? xb[ok ? off + j : 0] : 0.0f;
                }
            }
        }
    }
    __device__ __forceinline__ void store(const float (&r)[STAGE], float* T, int tid) const {
        const int x = (tid & 31) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            *reinterpret_cast<float4*>(T + ((tid >> 5) + 8 * h) * LDT + x) =
                make_float4(r[4 * h], r[4 * h + 1], r[4 * h + 2], r[4 * h + 3]);
    }
};

// FAST: Cout % 128 == 0, K % 16 == 0, P % 128 == 0 and vector-loadable weights: the
// branch-free software-pipelined main loop of the GEMM (the gather keeps its own guards).
template <class SB, bool FAST>
__global__ __launch_bounds__(NTHREADS, 2) void conv_mma_kernel(SrcKContig sa, ConvGeom g,
                                                               ConvOut o, int tiles_p) {
    __shared__ __attribute__((aligned(16))) float lds[GEMM_LDS_FLOATS];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = tile / tiles_p, tp = tile - tc * tiles_p;   // pixel tile fastest
    const int c0 = tc * BM, p0 = tp * BN;
    SB sb;
    sb.g = g;
    sb.init(p0, threadIdx.x);
    Acc acc;
    acc_zero(acc);
    contract_loop<FAST>(acc, sa, sb, c0, p0, 0, g.K, lds);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, cl = lane & 31;
    const int per = g.Ho * g.Wo;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = p0 + wn * 64 + j * 32 + cl;
        if (p >= g.P) continue;
        const int n = p / per, rem = p - n * per;
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        const int64_t pix = (int64_t)(oy * o.os + o.py) * o.WoP + (ox * o.os + o.px);
        const int64_t plane = (int64_t)o.HoP * o.WoP;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = c0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (co >= o.Cout) continue;
                float t = acc.t[i][j][r];
                if (o.bias) t += o.bias[co];
                const int64_t idx = ((int64_t)n * o.Cout + co) * plane + pix;
                if (o.preact) o.preact[idx] = t;
                o.y[idx] = act_fwd(t, o.act);
            }
    }
}

// ConvTranspose2d(4,2,1) forward, both column-parity classes of one row parity in one workgroup:
// the same 128 x 128 (channels x logical pixels) tile is contracted twice -- class px = 0 with
// its packed weights and tap offset, then px = 1 -- into two accumulator sets, and the epilogue
// stores the pair (2b, 2b+1) of output columns as one 8-B value per lane: 256 contiguous bytes
// per 32 lanes.  One class per launch writes 4 B at an 8-B stride instead (every line of the
// output visited by two launches, half used each time): 61.8 TF on the 512 -> 256 layer.
template <class SB, bool FAST>
__global__ __launch_bounds__(NTHREADS, 2) void convt_pair_kernel(SrcKContig sa, ConvGeom g, ConvOut o,
                                                                 int tiles_p, int64_t class_stride) {
    __shared__ __attribute__((aligned(16))) float lds[GEMM_LDS_FLOATS];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = tile / tiles_p, tp = tile - tc * tiles_p;
    const int c0 = tc * BM, p0 = tp * BN;
    const int py = blockIdx.y;               // row parity: both in one launch (twice the workgroups)
    g.oy0 = py;                              // tap th reads input row a + py - th
    o.py = py;
    Acc acc[2];
#pragma unroll
    for (int px = 0; px < 2; ++px) {
        SB sb;
        sb.g = g;
        sb.g.ox0 = px;                       // tap tw reads input column b + px - tw
        sb.init(p0, threadIdx.x);
        SrcKContig sw = sa;
        sw.p = sa.p + (py * 2 + px) * class_stride;
        acc_zero(acc[px]);
        contract_loop<FAST>(acc[px], sw, sb, c0, p0, 0, g.K, lds);
        __syncthreads();                     // the tiles of this class are done with the LDS
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, cl = lane & 31;
    const int per = g.Ho * g.Wo;
    const int64_t plane = (int64_t)o.HoP * o.WoP;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = p0 + wn * 64 + j * 32 + cl;
        if (p >= g.P) continue;
        const int n = p / per, rem = p - n * per;
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        const int64_t pix = (int64_t)(oy * 2 + o.py) * o.WoP + ox * 2;     // even: 8-B aligned
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = c0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (co >= o.Cout) continue;
                float t0 = acc[0].t[i][j][r], t1 = acc[1].t[i][j][r];
                if (o.bias) { const float bv = o.bias[co]; t0 += bv; t1 += bv; }
                const int64_t idx = ((int64_t)n * o.Cout + co) * plane + pix;
                if (o.preact) *reinterpret_cast<float2*>(o.preact + idx) = make_float2(t0, t1);
                *reinterpret_cast<float2*>(o.y + idx) = make_float2(act_fwd(t0, o.act), act_fwd(t1, o.act));
            }
    }
}

// Direct kernel for very few output channels (COUT <= 8): lane per logical pixel.
template <int COUT>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvGeom g, const float* __restrict__ w,
                                                          ConvOut o) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= g.P) return;
    const int per = g.Ho * g.Wo;
    const int n = p / per, rem = p - n * per;
    const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
    const float* xb = g.x + (int64_t)n * g.C * g.H * g.W;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.0f;
    const int taps = g.nty * g.ntx;
    for (int ci = 0; ci < g.C; ++ci) {
        for (int ty = 0; ty < g.nty; ++ty) {
            const int iy = oy * g.stride + g.oy0 + g.oys * ty;
            for (int tx = 0; tx < g.ntx; ++tx) {
                const int ix = ox * g.stride + g.ox0 + g.oxs * tx;
                float v = 0.0f;
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                    v = xb[((int64_t)ci * g.H + iy) * g.W + ix];
                const float* wk = w + (int64_t)ci * taps + ty * g.ntx + tx;  // + co*K
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] = fmaf(wk[(int64_t)c * g.K], v, acc[c]);
            }
        }
    }
    const int64_t plane = (int64_t)o.HoP * o.WoP;
    const int64_t pix = (int64_t)(oy * o.os + o.py) * o.WoP + (ox * o.os + o.px);
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        float t = acc[c];
        if (o.bias) t += o.bias[c];
        const int64_t idx = ((int64_t)n * o.Cout + c) * plane + pix;
        if (o.preact) o.preact[idx] = t;
        o.y[idx] = act_fwd(t, o.act);
    }
}

// Register-blocked form of the direct kernel for the 3x3 / stride 1 / pad 1 output layers
// (256->3 decoder pixels, 512->4 latents): a lane owns FOUR consecutive pixels of a row, so
// one aligned 16-B load plus the two neighbours feeds 4 pixels x 3 taps x COUT FMAs (the
// per-pixel form issues one 4-B load per COUT FMAs and is load-issue/latency bound at
// ~3 % of the VALU rate).  The fma order per output (ci, ty, tx ascending) is unchanged.
// FLIP: taps walk the input backwards (offset 1 - t instead of t - 1): the stride-1 data
// gradient of the same layers, whose packed weights are already in that tap order.
template <int COUT, bool FLIP>
__device__ __forceinline__ void conv_direct4_channels(const ConvGeom& g, const float* __restrict__ w,
                                                      const float* __restrict__ xb, int oy, int ox, int c_begin,
                                                      int c_end, float (&acc)[COUT][4]) {
    const bool left = ox > 0, right = ox + 4 < g.W;
    for (int ci = c_begin; ci < c_end; ++ci) {
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int iy = FLIP ? oy + 1 - ty : oy - 1 + ty;
            float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)g.H) {
                const float* row = xb + ((int64_t)ci * g.H + iy) * g.W + ox;
                const float4 m = *reinterpret_cast<const float4*>(row);
                v[1] = m.x; v[2] = m.y; v[3] = m.z; v[4] = m.w;
                if (left) v[0] = row[-1];
                if (right) v[5] = row[4];
            }
            const float* wk = w + (int64_t)ci * 9 + ty * 3;     // + co*K + tx
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int c = 0; c < COUT; ++c) {
                    const float wv = wk[(int64_t)c * g.K + tx];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[c][j] = fmaf(wv, v[j + (FLIP ? 2 - tx : tx)], acc[c][j]);
                }
        }
    }
}

// CS == 1: one wave per 256 pixels walks all input channels.  CS > 1 (few images: a 4-image 128 x 128 output
// is 256 waves, one per CU, each a chain of 768 dependent row loads -- 171 us for 0.9 GFLOP): the workgroup's
// CS waves take C / CS channels each, the partial sums meet in LDS and are added in wave order.
template <int COUT, bool FLIP, int CS>
__global__ __launch_bounds__(64 * CS) void conv_direct4_kernel(ConvGeom g, const float* __restrict__ w,
                                                               ConvOut o) {
    __shared__ float part[CS > 1 ? CS : 1][COUT * 4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = (blockIdx.x * 64 + lane) * 4;
    const bool live = p < g.P;
    if (CS == 1 && !live) return;
    const int per = g.Ho * g.Wo;
    const int n = live ? p / per : 0, rem = live ? p - n * per : 0;
    const int oy = rem / g.Wo, ox = rem - oy * g.Wo;      // ox % 4 == 0, W == Wo
    const float* xb = g.x + (int64_t)n * g.C * g.H * g.W;
    float acc[COUT][4];
#pragma unroll
    for (int c = 0; c < COUT; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = 0.0f;
    const int cper = g.C / CS;
    if (live) conv_direct4_channels<COUT, FLIP>(g, w, xb, oy, ox, wave * cper, (wave + 1) * cper, acc);
    if (CS > 1) {
#pragma unroll
        for (int c = 0; c < COUT; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[wave][c * 4 + j][lane] = acc[c][j];
        __syncthreads();
        if (wave != 0 || !live) return;
#pragma unroll
        for (int c = 0; c < COUT; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = part[0][c * 4 + j][lane];
                for (int z = 1; z < CS; ++z) t += part[z][c * 4 + j][lane];
                acc[c][j] = t;
            }
    }
    const int64_t plane = (int64_t)o.HoP * o.WoP;
    const int64_t pix = (int64_t)oy * o.WoP + ox;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        const float b = o.bias ? o.bias[c] : 0.0f;
        const float4 t = make_float4(acc[c][0] + b, acc[c][1] + b, acc[c][2] + b, acc[c][3] + b);
        const int64_t idx = ((int64_t)n * o.Cout + c) * plane + pix;
        if (o.preact) *reinterpret_cast<float4*>(o.preact + idx) = t;
        *reinterpret_cast<float4*>(o.y + idx) =
            make_float4(act_fwd(t.x, o.act), act_fwd(t.y, o.act), act_fwd(t.z, o.act), act_fwd(t.w, o.act));
    }
}

// ConvTranspose2d weight (Cin, Cout, 4, 4) -> per output-parity class GEMM weights
// packed[cls][co][(ci, th, tw)], cls = py*2+px; taps of class parity p: kh = 1-p + 2*th
// (p=0: kh 1,3 ; p=1: kh 0,2).
__global__ void convt_pack_kernel(const float* __restrict__ w, int Cin, int Cout,
                                  float* __restrict__ packed) {
    const int64_t total = (int64_t)4 * Cout * Cin * 4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int tw = idx & 1, th = (idx >> 1) & 1;
        int64_t t = idx >> 2;
        const int ci = (int)(t % Cin);
        t /= Cin;
        const int co = (int)(t % Cout);
        const int cls = (int)(t / Cout);
        const int py = cls >> 1, px = cls & 1;
        const int kh = 1 - py + 2 * th, kw = 1 - px + 2 * tw;
        packed[idx] = w[(((int64_t)ci * Cout + co) * 4 + kh) * 4 + kw];
    }
}

// The same classes with the reduction tap-major, packed[cls][co][(th, tw)][ci], for the ring kernel.
__global__ void convt_pack_tap_kernel(const float* __restrict__ w, int Cin, int Cout,
                                      float* __restrict__ packed) {
    const int64_t total = (int64_t)4 * Cout * 4 * Cin;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % Cin);
        int64_t t = idx / Cin;
        const int tw = (int)(t & 1), th = (int)((t >> 1) & 1);
        t >>= 2;
        const int co = (int)(t % Cout);
        const int cls = (int)(t / Cout);
        const int py = cls >> 1, px = cls & 1;
        const int kh = 1 - py + 2 * th, kw = 1 - px + 2 * tw;
        packed[idx] = w[(((int64_t)ci * Cout + co) * 4 + kh) * 4 + kw];
    }
}

// ---------------------------------------------------------------------------
// Backward: weight gradient.  dW[cg][(cx,kh,kw)] = sum over pixels p=(n,gy,gx) of
//   G[n][cg][gy][gx] * X[n][cx][gy*s + kh - pad][gx*s + kw - pad]
// a GEMM whose reduction runs over pixels (split over grid.z through fp32 slabs).
// Conv2d: G = dT (N,Cout,Ho,Wo), X = input -> dW in (Cout,Cin,k,k) order.
// ConvTranspose2d(4,2,1): G = input (N,Cin,H,W), X = dT (N,Cout,2H,2W), k=4,s=2,p=1
//   -> dW in (Cin,Cout,4,4) order.  Same kernel.
// ---------------------------------------------------------------------------
struct WgradGeom {
    const float* G; int Cg, Gh, Gw;      // gradient-like tensor (N, Cg, Gh, Gw)
    const float* X; int Cx, H, W;        // tensor that is im2col'ed (N, Cx, H, W)
    int N, k, stride, pad;
    int P;                               // N*Gh*Gw
    int K2;                              // Cx*k*k
};

// A side: x index = cg, reduction index = pixel.
struct SrcGradPix {
    WgradGeom g;
    __device__ __forceinline__ bool interior(int, int, int) const { return false; }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        load(r, x0, k0, tid);
    }
    __device__ __forceinline__ void load(float (&r)[STAGE], int x0, int k0, int tid) const {
        const int cg = x0 + (tid >> 1);
        const int p0 = k0 + (tid & 1) * 8;
        const int per = g.Gh * g.Gw;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = p0 + q;
            float v = 0.0f;
            if (cg < g.Cg && p < g.P) {
                const int n = p / per, sp = p - n * per;
                v = g.G[((int64_t)n * g.Cg + cg) * per + sp];
            }
            r[q] = v;
        }
    }
    __device__ __forceinline__ void store(const float (&r)[STAGE], float* T, int tid) const {
        const int x = tid >> 1, k = (tid & 1) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) T[(k + q) * LDT + x] = r[q];
    }
};

// B side: x index = (cx,kh,kw), reduction index = pixel.
struct SrcIm2colPix {
    WgradGeom g;
    int cx, kh, kw;
    bool valid;
    __device__ __forceinline__ void init(int x0, int tid) {
        const int kidx = x0 + (tid >> 1);
        valid = kidx < g.K2;
        const int kk = g.k * g.k;
        cx = valid ? kidx / kk : 0;
        const int rem = valid ? kidx - cx * kk : 0;
        kh = rem / g.k;
        kw = rem - kh * g.k;
    }
    __device__ __forceinline__ bool interior(int, int, int) const { return false; }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        load(r, x0, k0, tid);
    }
    __device__ __forceinline__ void load(float (&r)[STAGE], int, int k0, int tid) const {
        const int p0 = k0 + (tid & 1) * 8;
        const int per = g.Gh * g.Gw;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = p0 + q;
            float v = 0.0f;
            if (valid && p < g.P) {
                const int n = p / per, sp = p - n * per;
                const int gy = sp / g.Gw, gx = sp - gy * g.Gw;
                const int iy = gy * g.stride + kh - g.pad, ix = gx * g.stride + kw - g.pad;
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                    v = g.X[(((int64_t)n * g.Cx + cx) * g.H + iy) * g.W + ix];
            }
            r[q] = v;
        }
    }
    __device__ __forceinline__ void store(const float (&r)[STAGE], float* T, int tid) const {
        const int x = tid >> 1, k = (tid & 1) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) T[(k + q) * LDT + x] = r[q];
    }
};

// Row forms of the two loaders for gradient rows that are a multiple of 8 wide: the 8
// pixels a thread stages per k-tile are 8 consecutive floats of ONE row of one image, so the
// pixel -> (image, row, column) decomposition happens once per tile (two multiply-high
// divisions) instead of twice per element, and the loads are two 16-B accesses.
// v / d for 0 <= v < 2^31, d > 0 without the ~40-instruction integer divide: float quotient
// (relative error 2^-23, i.e. off by at most one for quotients below 2^22) plus one
// correction step each way.
__device__ __forceinline__ int div_fix(int v, int d, float inv) {
    int q = (int)((float)v * inv);
    q -= (int64_t)q * d > v;
    q += (int64_t)(q + 1) * d <= v;
    return q;
}

struct SrcGradPixRow {
    WgradGeom g;
    float inv_per;
    __device__ __forceinline__ void init() { inv_per = 1.0f / (float)(g.Gh * g.Gw); }
    __device__ __forceinline__ bool interior(int, int, int) const { return false; }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        load(r, x0, k0, tid);
    }
    __device__ __forceinline__ void load(float (&r)[STAGE], int x0, int k0, int tid) const {
        const int cg = x0 + (tid >> 1);
        const int p0 = k0 + (tid & 1) * 8;
        const int per = g.Gh * g.Gw;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (cg < g.Cg && p0 < g.P) {
            const int n = div_fix(p0, per, inv_per), sp = p0 - n * per;
            const float* q = g.G + ((int64_t)n * g.Cg + cg) * per + sp;
            a = *reinterpret_cast<const float4*>(q);
            b = *reinterpret_cast<const float4*>(q + 4);
        }
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
        r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    }
    __device__ __forceinline__ void store(const float (&r)[STAGE], float* T, int tid) const {
        const int x = tid >> 1, k = (tid & 1) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) T[(k + q) * LDT + x] = r[q];
    }
};

struct SrcIm2colPixRow {
    WgradGeom g;
    int cx, kh, kw;
    float inv_per, inv_gw;
    bool valid;
    __device__ __forceinline__ void init(int x0, int tid) {
        const int kidx = x0 + (tid >> 1);
        valid = kidx < g.K2;
        const int kk = g.k * g.k;
        cx = valid ? kidx / kk : 0;
        const int rem = valid ? kidx - cx * kk : 0;
        kh = rem / g.k;
        kw = rem - kh * g.k;
        inv_per = 1.0f / (float)(g.Gh * g.Gw);
        inv_gw = 1.0f / (float)g.Gw;
    }
    __device__ __forceinline__ bool interior(int, int, int) const { return false; }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        load(r, x0, k0, tid);
    }
    __device__ __forceinline__ void load(float (&r)[STAGE], int, int k0, int tid) const {
        const int p0 = k0 + (tid & 1) * 8;
        const int per = g.Gh * g.Gw;
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = 0.0f;
        if (!valid || p0 >= g.P) return;
        const int n = div_fix(p0, per, inv_per), sp = p0 - n * per;
        const int gy = div_fix(sp, g.Gw, inv_gw), gx = sp - gy * g.Gw;
        const int iy = gy * g.stride + kh - g.pad;
        if ((unsigned)iy >= (unsigned)g.H) return;
        const int ix0 = gx * g.stride + kw - g.pad;
        const float* row = g.X + (((int64_t)n * g.Cx + cx) * g.H + iy) * g.W;
        if (g.stride == 1 && ix0 >= 0 && ix0 + 7 < g.W) {
            const F4u a = *reinterpret_cast<const F4u*>(row + ix0);
            const F4u b = *reinterpret_cast<const F4u*>(row + ix0 + 4);
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
            r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ix = ix0 + q * g.stride;
                const bool ok = (unsigned)ix < (unsigned)g.W;
                r[q] = ok ? row[ok ? ix : 0] : 0.0f;
            }
        }
    }
    __device__ __forceinline__ void store(const float (&r)[STAGE], float* T, int tid) const {
        const int x = tid >> 1, k = (tid & 1) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) T[(k + q) * LDT + x] = r[q];
    }
};

// ROW: gradient rows a multiple of 8 wide (row loaders).  FAST: additionally Cg, K2 multiples
// of 128 and every pixel split a multiple of 16: the branch-free pipelined main loop.
template <bool ROW, bool FAST>
__global__ __launch_bounds__(NTHREADS, 2) void conv_wgrad_kernel(WgradGeom g, int tiles_n,
                                                                 int per_split,
                                                                 float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float lds[GEMM_LDS_FLOATS];
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int k_begin = blockIdx.z * per_split;
    const int k_end = min(g.P, k_begin + per_split);
    Acc acc;
    acc_zero(acc);
    if (ROW) {
        SrcGradPixRow sa{g, 0.0f};
        sa.init();
        SrcIm2colPixRow sb;
        sb.g = g;
        sb.init(n0, threadIdx.x);
        contract_loop<FAST>(acc, sa, sb, m0, n0, k_begin, k_end, lds);
    } else {
        SrcGradPix sa{g};
        SrcIm2colPix sb;
        sb.g = g;
        sb.init(n0, threadIdx.x);
        contract_loop<false>(acc, sa, sb, m0, n0, k_begin, k_end, lds);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, cl = lane & 31;
    float* out = slabs + (int64_t)blockIdx.z * g.Cg * g.K2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + cl;
        if (col >= g.K2) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (row < g.Cg) out[(int64_t)row * g.K2 + col] = acc.t[i][j][r];
            }
    }
}

// db[c] = sum_{n,sp} G[n][c][sp]; one 1024-thread block per channel (16 waves per CU keep
// enough 16-B loads in flight to stream the planes at HBM rate), fixed summation order.
__global__ __launch_bounds__(1024) void channel_sum_kernel(const float* __restrict__ G, int N, int C,
                                                           int HW, float* __restrict__ out) {
    __shared__ float red[1024];
    const int c = blockIdx.x;
    float s = 0.0f;
    if ((HW & 3) == 0 && (((uintptr_t)G) & 15) == 0) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = 0; n < N; ++n) {
            const float4* p = reinterpret_cast<const float4*>(G + ((int64_t)n * C + c) * HW);
            for (int i = threadIdx.x; i < (HW >> 2); i += 1024) {
                const float4 v = p[i];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        s = (a.x + a.y) + (a.z + a.w);
    } else {
        for (int n = 0; n < N; ++n) {
            const float* p = G + ((int64_t)n * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += 1024) s += p[i];
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
}

// ---------------------------------------------------------------------------------
// 3x3 / stride 1 / padding 1 convolution on the GEMM's LDS-DMA ring (gemm.hip pf_ring): the same
// 128 x 128 x 16 tiles, 4 stages, fragments of tile t+1 read while tile t's MFMAs issue, and a
// k-loop whose vector-ALU work is ~10 instructions per 32 MFMAs instead of ~100 (the fp32 MFMA and
// the vector ALU do not overlap on this chip: the im2col index arithmetic of conv_mma_kernel --
// magic-number divisions, 64-bit address adds, border compares, a guarded per-element path for the
// first and last group of every image row -- is what holds that kernel at 95-105 TF).
//   * reduction order is TAP-MAJOR: k = tap * C + channel.  A k-tile is 16 channels of ONE tap, so
//     tap, channel base and the input offset of the tile advance on the scalar unit.
//   * A = weights re-packed [M][9][C] (conv_pack_tap_kernel; the flip of the data gradient is
//     applied there, so the kernel only knows offset = tap - 1): k-contiguous rows -> LDS-DMA.
//   * B = im2col(x): lane (k row, 4 consecutive pixels of an image row) issues ONE 16-B buffer
//     load per k row at byte offset lane_const + tile_scalar.  The buffer resource spans exactly
//     the tensor, so a negative or past-the-end offset returns zeros and cannot fault: rows above /
//     below the image are redirected to offset ~0u (one v_cndmask), the element left of column 0 /
//     right of column W-1 is zeroed with one v_cndmask; the 16-B load that would START one float
//     before column 0 is issued one float later and its elements shifted (hardware range-checks a
//     wrapped offset as out of range for all four dwords).  Registers -> ds_write_b128 into the
//     stage's [k][128] image, which is the GEMM's tile-contiguous operand layout.
// Results differ from conv_mma_kernel only by the summation order.
__global__ void conv_pack_tap_kernel(const float* __restrict__ w, int M, int C, int64_t sm, int64_t sc,
                                     int flip, float* __restrict__ packed, int ntaps = 9) {
    const int64_t total = (int64_t)M * ntaps * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const int64_t t = idx / C;
        const int tap = (int)(t % ntaps), m = (int)(t / ntaps);
        packed[idx] = w[m * sm + c * sc + (flip ? ntaps - 1 - tap : tap)];
    }
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buffer_load16(unsigned voff, u32x4_t rsrc) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    return v;
}

// One pass of the ring over the whole reduction (taps x channels) of one 128 x 128 tile into `acc`.
// Taps: g.nty x g.ntx, tap (ty, tx) reads the input at (+ g.oy0 + g.oys * ty, + g.ox0 + g.oxs * tx), every
// offset in {-1, 0, +1}: the 3x3 convolution (oy0 = ox0 = -1, oys = oxs = 1) and the 2x2-tap parity classes
// of ConvTranspose2d(4, 2, 1) (oy0 = py, oys = -1, ...).  wp: this product's weights [M][taps][C].
// T3: the 3x3 geometry as compile-time constants (the general form costs ~4 % there).
// S2: the input is read with stride 2 (Conv2d(3, stride 2, padding 1) forward; H = 2 Ho, W = 2 Wo): a lane's 4
// output pixels read every other input column -- four 4-B buffer loads per k row instead of one 16-B load.
template <int OFF>
__device__ __forceinline__ float buffer_load4(unsigned voff, u32x4_t rsrc) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%3" : "=v"(v) : "v"(voff), "s"(rsrc), "n"(OFF) : "memory");
    return v;
}
template <bool T3, bool S2 = false>
__device__ __forceinline__ void conv_ring_pass(Acc& acc, float* lds, const float* __restrict__ wp, const ConvGeom& g,
                                               int c0m, int p0, unsigned x_bytes, int t0 = 0, int nk_part = 0) {
    const int nty = T3 ? 3 : g.nty, ntx = T3 ? 3 : g.ntx;
    const int oy0 = T3 ? -1 : g.oy0, oys = T3 ? 1 : g.oys, ox0 = T3 ? -1 : g.ox0, oxs = T3 ? 1 : g.oxs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int x = lane & 31, h = lane >> 5;
    const int C = g.C, H = g.H, W = g.W;
    // k-tiles [t0, t0 + nk) of the taps * C / 16 of the whole reduction (a split launch reduces a part)
    const int nk = nk_part > 0 ? nk_part : g.K / BK;
    const int64_t lda = g.K;

    // ---- A: this wave's two DMA instructions per tile (as pf_ring)
    const float* a_org = wp + (int64_t)c0m * lda + (int64_t)t0 * BK;
    const unsigned oa0 = dma_lane_off<true>(lda, wave * 2, lane), oa1 = dma_lane_off<true>(lda, wave * 2 + 1, lane);
    const unsigned my_dma_addr = __builtin_amdgcn_readfirstlane(lds_addr(lds + wave * 512));
    auto issue_a = [&](int t, int stage) {
        const int k = min(t, nk - 1) * BK;   // clamped: a harmless re-load past the end
        const char* ak = reinterpret_cast<const char*>(a_org + k);
        const unsigned dst = my_dma_addr + stage * (DMA_STAGE_FLOATS * 4);
        dma16_saddr(ak, oa0, dst);
        dma16_saddr(ak, oa1, dst + 1024);
    };
    // ---- B: lane constants.  Thread = (k row kr = tid >> 5 and kr + 8, pixel group tid & 31)
    const int kr = tid >> 5, grp = tid & 31;
    const int p = p0 + grp * 4;              // first of this lane's 4 pixels (same image row: W % 4 == 0)
    const int Hq = S2 ? g.Ho : H, Wq = S2 ? g.Wo : W;          // the pixel grid (= the input's unless strided)
    constexpr int ST = S2 ? 2 : 1;
    const int per = Hq * Wq;
    const int n = p / per, rem = p - n * per;
    const int oy = rem / Wq, ox = rem - oy * Wq;
    const unsigned lane_off = (unsigned)((((int64_t)n * C + kr) * H + oy * ST) * W + ox * ST) * 4u;   // host: x_bytes < 2^31
    const unsigned row8 = (unsigned)(8 * H * W) * 4u;                                     // k row + 8
    const bool top = oy == 0, bottom = oy == Hq - 1, left = ox == 0, right = ox + 4 == Wq;
    // raw buffer descriptor: base, stride 0, num_records = bytes, 32-bit untyped data format
    const uint64_t xa = reinterpret_cast<uint64_t>(g.x);
    const u32x4_t rsrc = {(unsigned)xa, (unsigned)(xa >> 32) & 0xffffu, x_bytes, 0x00020000u};
    const unsigned b_dst = lds_addr(lds + DMA_OP_FLOATS + kr * 128 + grp * 4);
    // scalar state of the tile whose B loads are issued next: tap (ty, tx), channel base cb
    int ty = 0, tx = 0, cb = 0;
    if (t0) {                                // a split's first tile: t0 = (ty * ntx + tx) * C / 16 + cb / 16
        const int ct = C >> 4, tap = t0 / ct;
        cb = (t0 - tap * ct) * 16;
        ty = tap / ntx;
        tx = tap - ty * ntx;
    }
    struct BRegs { f32x4 r0, r1; };          // one tile's two k rows in flight
    int U_dx = 0, V_dx = 0;                   // ... and the column offset they were loaded for (scalars: kept apart from the vectors)
    auto load_b = [&](BRegs& br, int& br_dx) {   // issues the two loads of tile (ty, tx, cb), then advances the state
        const int dy = oy0 + oys * ty, dx = ox0 + oxs * tx;
        const int soff = ((cb * H + dy) * W + dx) * 4;
        // strided: rows 2 oy + dy, dy in {-1, 0, 1 (, 2)}: only dy = -1 at the top and dy = 2 at the bottom fall outside
        const bool rinv = S2 ? ((dy < 0 && top) || (dy > 1 && bottom)) : ((dy < 0 && top) || (dy > 0 && bottom));
        unsigned v0 = lane_off + (unsigned)soff + ((!S2 && dx < 0 && left) ? 4u : 0u);   // see above: start at column 0
        unsigned v1 = v0 + row8;
        v0 = rinv ? 0xFFFFFF00u : v0;       // beyond num_records, no wrap: every dword reads as 0
        v1 = rinv ? 0xFFFFFF00u : v1;
        if constexpr (S2) {
            br.r0 = f32x4{buffer_load4<0>(v0, rsrc), buffer_load4<8>(v0, rsrc), buffer_load4<16>(v0, rsrc), buffer_load4<24>(v0, rsrc)};
            br.r1 = f32x4{buffer_load4<0>(v1, rsrc), buffer_load4<8>(v1, rsrc), buffer_load4<16>(v1, rsrc), buffer_load4<24>(v1, rsrc)};
        } else {
            br.r0 = buffer_load16(v0, rsrc);
            br.r1 = buffer_load16(v1, rsrc);
        }
        br_dx = dx;
        cb += 16;
        if (cb == C) { cb = 0; ++tx; if (tx == ntx) { tx = 0; ++ty; } }
        if (ty == nty) { ty = nty - 1; tx = ntx - 1; cb = C - 16; }   // past the end: repeat the last tile (never used)
    };
    auto store_b = [&](BRegs& br, int br_dx, int stage) {   // border selects, then the two k rows into the stage's [k][128] image
        if (S2) {                            // column 2 ox + dx: only dx = -1 at the left edge (dx = 2 at the right) is outside
            const bool zl = br_dx < 0 && left, zr = br_dx > 1 && right;
            br.r0[0] = zl ? 0.f : br.r0[0]; br.r1[0] = zl ? 0.f : br.r1[0];
            br.r0[3] = zr ? 0.f : br.r0[3]; br.r1[3] = zr ? 0.f : br.r1[3];
        } else if (br_dx != 0) {             // uniform
            const bool shl = br_dx < 0 && left, zr = br_dx > 0 && right;
            const f32x4 c0 = br.r0, c1 = br.r1;
            br.r0 = f32x4{shl ? 0.f : c0[0], shl ? c0[0] : c0[1], shl ? c0[1] : c0[2], shl ? c0[2] : (zr ? 0.f : c0[3])};
            br.r1 = f32x4{shl ? 0.f : c1[0], shl ? c1[0] : c1[1], shl ? c1[1] : c1[2], shl ? c1[2] : (zr ? 0.f : c1[3])};
        }
        const unsigned d = b_dst + stage * (DMA_STAGE_FLOATS * 4);
        asm volatile("ds_write_b128 %0, %1" :: "v"(d), "v"(br.r0) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:4096" :: "v"(d), "v"(br.r1) : "memory");   // k row + 8
    };

    FragBase fb;
    frag_bases<true>(lds, wm, x, h, fb.a0, fb.a1);
    frag_bases<false>(lds, wn, x, h, fb.b0, fb.b1);
    // prologue: B tiles 0 and 1 written; then, in the order of every body, B(2) | A(3) | B(3): at the top of
    // body t the queue holds B(t+2) A(t+3)... see the waits below
    issue_a(0, 0);
    issue_a(1, 1);
    issue_a(2, 2);
    BRegs U, V;
    load_b(U, U_dx);
    load_b(V, V_dx);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(U.r0), "+v"(U.r1), "+v"(V.r0), "+v"(V.r1) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    store_b(U, U_dx, 0);
    store_b(V, V_dx, 1);
    load_b(U, U_dx);                                         // tile 2: written at the top of body 0
    issue_a(3, 3);
    load_b(V, V_dx);                                         // tile 3: written at the top of body 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frags8 P, Q;
    frags_read_s<true, false, 0>(P, fb);
    frags_wait(P);
    __builtin_amdgcn_sched_barrier(0);
    int t = 0;
    // Body for tile t (stage S = t % 4), fragments in CUR.  BW holds tile t+2 (loaded two bodies ago), the
    // other register set tile t+3.  Queue at the top, oldest first: [A(t+2) x2,] B(t+2) x2, A(t+3) x2,
    // B(t+3) x2 (x8 each in the strided form): B(t+2) (and every older A) has landed at vmcnt(4) (vmcnt(10)).
#define QARIG_CV_BODY(CUR, NXT, S, BW)                                                             \
    {                                                                                              \
        /* the registers pass THROUGH the wait: their selects cannot be placed above it */         \
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(BW.r0), "+v"(BW.r1) : "n"(S2 ? 10 : 4) : "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        store_b(BW, BW##_dx, (S + 2) % 4);                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        __builtin_amdgcn_s_barrier();          /* B(t+2) published; all reads of tile t retired */   \
        issue_a(t + 4, S);                     /* into the stage tile t has just vacated */        \
        load_b(BW, BW##_dx);                   /* tile t+4 into the registers just written out */  \
        frags_read_s<true, false, (S + 1) % 4>(NXT, fb);                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        frags_mma(acc, CUR);                                                                       \
        frags_wait(NXT);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        ++t;                                                                                       \
    }
    while (t + 4 <= nk) {
        QARIG_CV_BODY(P, Q, 0, U)
        QARIG_CV_BODY(Q, P, 1, V)
        QARIG_CV_BODY(P, Q, 2, U)
        QARIG_CV_BODY(Q, P, 3, V)
    }
    if (t < nk) QARIG_CV_BODY(P, Q, 0, U)
    if (t < nk) QARIG_CV_BODY(Q, P, 1, V)
    if (t < nk) QARIG_CV_BODY(P, Q, 2, U)
#undef QARIG_CV_BODY
    // The last bodies' B loads (tiles past the end) are never stored, but they are still in flight: their
    // destination registers pass through this wait so that the compiler cannot hand them to the epilogue's
    // address arithmetic before the loads have landed (it believes an asm load completes where it is issued;
    // without the tie a late load overwrote an output address: a memory fault at Cin = 256).
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(U.r0), "+v"(U.r1), "+v"(V.r0), "+v"(V.r1) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                         // every wave is done with the ring: the caller may start another pass
}

// PAIR = false: one product per workgroup (3x3 / stride 1 / padding 1 forward and input gradient).
// PAIR = true: ConvTranspose2d(4, 2, 1) forward: blockIdx.y = row parity py, the two column-parity classes px of
// the same 128 x 128 (channels x logical pixels) tile are reduced one after the other into two accumulator
// sets and stored as 8-B pairs of output columns (as convt_pair_kernel); class weights at wp + cls * class_stride.
template <bool PAIR, bool S2 = false, bool T3K = true>
__global__ __launch_bounds__(NTHREADS, 2) void conv3x3_ring_kernel(const float* __restrict__ wp, ConvGeom g,
                                                                   ConvOut o, int tiles_p, unsigned x_bytes,
                                                                   int64_t class_stride, int mode, int splits,
                                                                   int64_t slab_stride) {
    __shared__ __attribute__((aligned(16))) float lds[PF_STAGES * DMA_STAGE_FLOATS];   // 64 KB
    // splits > 1 (few images: too few tiles to fill the chip): blockIdx.z reduces k-tiles
    // [z, z + 1) * K / 16 / splits into its own slab of raw sums (o.y = slab 0, bias / activation / preact
    // left to conv_split_reduce_kernel)
    const int kz = splits > 1 ? (int)blockIdx.z : 0;
    o.y += kz * slab_stride;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = tile / tiles_p, tp = tile - tc * tiles_p;   // pixel tile fastest
    const int c0m = tc * BM, p0 = tp * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, cl = lane & 31;
    const int per = S2 ? g.Ho * g.Wo : g.H * g.W, W = S2 ? g.Wo : g.W;   // the pixel grid
    const int64_t plane = (int64_t)o.HoP * o.WoP;
    if constexpr (!PAIR) {
        Acc acc;
        acc_zero(acc);
        const int nkp = splits > 1 ? g.K / BK / splits : 0;
        conv_ring_pass<T3K, S2>(acc, lds, wp, g, c0m, p0, x_bytes, kz * nkp, nkp);
        // epilogue: as conv_mma_kernel (32 consecutive pixels of one output channel per store instruction)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pp = p0 + wn * 64 + j * 32 + cl;
            const int nn = pp / per, rr = pp - nn * per;
            const int yy = rr / W, xx = rr - yy * W;
            const int64_t pix = (int64_t)yy * o.WoP + xx;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = c0m + wm * 64 + i * 32 + acc_row(r, lane);
                    float v = acc.t[i][j][r];
                    if (o.bias) v += o.bias[co];
                    const int64_t idx = ((int64_t)nn * o.Cout + co) * plane + pix;
                    if (o.preact) o.preact[idx] = v;
                    o.y[idx] = act_fwd(v, o.act);
                }
        }
    } else {
        // mode 0: ConvTranspose2d(4, 2, 1) forward, class (py, px) = 2 x 2 taps at offsets (py - th, px - tw), class
        // weights class_stride apart.  mode 1: input gradient of Conv2d(3, stride 2, padding 1), class (ry, rx) of
        // dx = (1 + ry) x (1 + rx) taps at offsets (ry - ty, rx - tx) over dT; classes packed back to back
        // ([M][taps][C] each: 1, 2, 2, 4 taps).
        const int py = blockIdx.y;
        g.oy0 = py;
        if (mode == 1) g.nty = 1 + py;
        Acc acc[2];
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            g.ox0 = px;
            int64_t woff = (py * 2 + px) * class_stride;
            if (mode == 1) {
                g.ntx = 1 + px;
                g.K = g.C * g.nty * g.ntx;
                const int cls = py * 2 + px;
                woff = (int64_t)o.Cout * g.C * (cls == 0 ? 0 : (cls == 1 ? 1 : (cls == 2 ? 3 : 5)));
            }
            acc_zero(acc[px]);
            const int nkp = splits > 1 ? g.K / BK / splits : 0;     // (mode 0 only: equal classes)
            conv_ring_pass<false>(acc[px], lds, wp + woff, g, c0m, p0, x_bytes, kz * nkp, nkp);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pp = p0 + wn * 64 + j * 32 + cl;
            const int nn = pp / per, rr = pp - nn * per;
            const int yy = rr / W, xx = rr - yy * W;
            const int64_t pix = (int64_t)(yy * 2 + py) * o.WoP + xx * 2;     // even: 8-B aligned
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = c0m + wm * 64 + i * 32 + acc_row(r, lane);
                    float t0 = acc[0].t[i][j][r], t1 = acc[1].t[i][j][r];
                    if (o.bias) { const float bv = o.bias[co]; t0 += bv; t1 += bv; }
                    const int64_t idx = ((int64_t)nn * o.Cout + co) * plane + pix;
                    if (o.preact) *reinterpret_cast<float2*>(o.preact + idx) = make_float2(t0, t1);
                    *reinterpret_cast<float2*>(o.y + idx) = make_float2(act_fwd(t0, o.act), act_fwd(t1, o.act));
                }
        }
    }
}

// ---------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride 1 / padding 1 layers on the same ring:
//   dw[cg][cx][tap] = sum over pixels b of G[cg][b + sG] * X[cx][b + sX],
// a 128 (cg) x 128 (cx) output tile PER TAP, reduced over 16 consecutive base pixels of one image
// row per k-tile.  Both operands are pixel-contiguous rows ([channel][16 pixels], the GEMM's
// reduction-contiguous layout), both come in by range-checked LDS-DMA (buffer_load ... lds: voffset
// = the lane's row/chunk, soffset = the tile, a scalar), so the k-loop has NO vector-ALU work.
// The tap offset (dy, dx) = (ty-1, tx-1) is split so that every shift is non-negative -- the
// hardware range check treats a negative offset as out of range for the whole 16 bytes --:
// sG = (max(0,-dy), max(0,-dx)) on the gradient, sX = (max(0,dy), max(0,dx)) on the input.  What
// falls off the image: a row below the last one = a zero tile (its soffset is pushed out of range:
// the DMA writes zeros), the column right of the last one = element 15 of the row's last k-tile,
// overwritten with 0 in LDS by the wave that DMA'd the row.  Split over pixels across grid.z into
// slabs in dw's own layout; qarig_slab_reduce_f32 sums them.
__global__ __launch_bounds__(NTHREADS, 2) void conv3x3_wgrad_ring_kernel(WgradGeom g, int tiles_cx, int per,
                                                                         int ktiles, unsigned g_bytes,
                                                                         unsigned x_bytes, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float lds[PF_STAGES * DMA_STAGE_FLOATS];   // 64 KB
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tap = tile % 9, tmn = tile / 9;                 // the 9 taps of an output tile share its operand panels
    const int tcx = tmn % tiles_cx, tcg = tmn / tiles_cx;
    const int cg0 = tcg * BM, cx0 = tcx * BN;
    const int ty = tap / 3, tx = tap - ty * 3;
    const int sGy = ty == 0, sGx = tx == 0, sXy = ty == 2, sXx = tx == 2;
    const int k_begin = blockIdx.z * per;
    const int nk = min(per, ktiles - k_begin);               // k-tiles of this split (host: > 0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int x = lane & 31, h = lane >> 5;
    const int H = g.H, W = g.W, HW = H * W;
    const int tiles_row = W / BK;                             // k-tiles per image row

    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.G), 0, g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, x_bytes, 0x00020000);
    const unsigned o0 = dma_lane_off<true>(HW, wave * 2, lane), o1 = dma_lane_off<true>(HW, wave * 2 + 1, lane);
    float* const my_dma = lds + wave * 512;                   // this wave's two 1-KB slots of an operand tile
    // position of the tile issued next: image n, row y, first column xs; tile index ti in the split
    int ti = 0;
    int n, y, xs;
    {
        const int row = k_begin / tiles_row;                  // (n, y) flattened
        xs = (k_begin - row * tiles_row) * BK;
        n = row / H;
        y = row - n * H;
    }
    auto issue = [&](int stage) {
        const bool zg = y + sGy >= H, zx = y + sXy >= H;     // the shifted row is below the image
        const unsigned sg = zg ? 0x80000000u : (unsigned)((((n * g.Cg + cg0) * H + y + sGy) * W + xs + sGx) * 4);
        const unsigned sx = zx ? 0x80000000u : (unsigned)((((n * g.Cx + cx0) * H + y + sXy) * W + xs + sXx) * 4);
        float* dst = my_dma + stage * DMA_STAGE_FLOATS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rG, (lds_ptr_t)dst, 16, o0, sg, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rG, (lds_ptr_t)(dst + 256), 16, o1, sg, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr_t)(dst + DMA_OP_FLOATS), 16, o0, sx, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr_t)(dst + DMA_OP_FLOATS + 256), 16, o1, sx, 0, 0);
        if (ti + 1 < nk) {                                    // past the end: the last tile again (never used)
            ++ti;
            xs += BK;
            if (xs == W) { xs = 0; ++y; if (y == H) { y = 0; ++n; } }
        }
    };
    // border fix: element 15 of this wave's 32 rows of the operand that is shifted by one column, in the
    // k-tile that ends at the last column.  fx = first column of the tile to be fixed next.
    const bool fix_a = sGx != 0, fix_any = sGx != 0 || sXx != 0;
    int fx = (k_begin % tiles_row) * BK;
    const int frow = wave * 32 + (lane & 31);
    const unsigned fix_addr = lds_addr(lds + (fix_a ? 0 : DMA_OP_FLOATS) + frow * 16) + ((3 ^ ((frow >> 2) & 3)) << 4) + 12;
    const float zero = 0.0f;
#define QARIG_WG_FIX(S)                                                                            \
    {                                                                                              \
        if (fix_any && fx == W - BK && lane < 32)                                                   \
            asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(fix_addr), "v"(zero), "n"((S) * DMA_STAGE_FLOATS * 4) : "memory"); \
        fx += BK;                                                                                  \
        if (fx == W) fx = 0;                                                                       \
    }

    Acc acc;
    acc_zero(acc);
    FragBase fb;
    frag_bases<true>(lds, wm, x, h, fb.a0, fb.a1);
    frag_bases<true>(lds, wn, x, h, fb.b0, fb.b1);
    issue(0);
    issue(1);
    issue(2);
    issue(3);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");         // tile 0 landed (1, 2, 3 in flight)
    QARIG_WG_FIX(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frags8 P, Q;
    frags_read_s<true, true, 0>(P, fb);
    frags_wait(P);
    __builtin_amdgcn_sched_barrier(0);
    int t = 0;
#define QARIG_WG_BODY(CUR, NXT, S)                                                                 \
    {                                                                                              \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   /* tile t+1 landed; t+2, t+3 in flight */ \
        QARIG_WG_FIX((S + 1) % 4)                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        __builtin_amdgcn_s_barrier();                                                              \
        issue(S);                              /* tile t+4 into the stage tile t has just vacated */ \
        frags_read_s<true, true, (S + 1) % 4>(NXT, fb);                                            \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        frags_mma(acc, CUR);                                                                       \
        frags_wait(NXT);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        ++t;                                                                                       \
    }
    while (t + 4 <= nk) {
        QARIG_WG_BODY(P, Q, 0)
        QARIG_WG_BODY(Q, P, 1)
        QARIG_WG_BODY(P, Q, 2)
        QARIG_WG_BODY(Q, P, 3)
    }
    if (t < nk) QARIG_WG_BODY(P, Q, 0)
    if (t < nk) QARIG_WG_BODY(Q, P, 1)
    if (t < nk) QARIG_WG_BODY(P, Q, 2)
#undef QARIG_WG_BODY
#undef QARIG_WG_FIX
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int cl = lane & 31;
    float* out = slabs + (int64_t)blockIdx.z * g.Cg * g.K2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cx = cx0 + wn * 64 + j * 32 + cl;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cg = cg0 + wm * 64 + i * 32 + acc_row(r, lane);
                out[(int64_t)cg * g.K2 + cx * 9 + tap] = acc.t[i][j][r];
            }
    }
}

// Conv2d(3, stride 2, padding 1) input gradient on the ring: the four output-parity classes of dx, tap-major,
// back to back: class (ry, rx) has (1 + ry) x (1 + rx) taps, packed[off(cls) + (ci * taps + ty * ntx + tx) * Cout + co]
// = W[co][ci][ky0 + 2 ty][kx0 + 2 tx], ky0 = (ry + 1) % 2; off = Cin * Cout * {0, 1, 3, 5}.
__global__ void conv_s2_dgrad_pack_tap_kernel(const float* __restrict__ w, int Cout, int Cin,
                                              float* __restrict__ packed) {
    const int64_t total = (int64_t)9 * Cin * Cout;
    const int64_t mc = (int64_t)Cin * Cout;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int cls = idx < mc ? 0 : (idx < 3 * mc ? 1 : (idx < 5 * mc ? 2 : 3));
        const int ry = cls >> 1, rx = cls & 1;
        const int nty = 1 + ry, ntx = 1 + rx, taps = nty * ntx;
        const int64_t r = idx - mc * (cls == 0 ? 0 : (cls == 1 ? 1 : (cls == 2 ? 3 : 5)));
        const int co = (int)(r % Cout);
        const int64_t t = r / Cout;
        const int tap = (int)(t % taps), ci = (int)(t / taps);
        const int ty = tap / ntx, tx = tap - ty * ntx;
        const int ky = (ry + 1) % 2 + 2 * ty, kx = (rx + 1) % 2 + 2 * tx;
        packed[idx] = w[(((int64_t)co * Cin + ci) * 3 + ky) * 3 + kx];
    }
}

// Backward-data weight packing for Conv2d: for output-parity class (ry,rx) of dx,
// packed[ci][(co,ty,tx)] = W[co][ci][kh0y + s*ty][kh0x + s*tx].
__global__ void conv_bwd_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int k, int s,
                                     int kh0y, int kh0x, int nty, int ntx,
                                     float* __restrict__ packed) {
    const int64_t total = (int64_t)Cin * Cout * nty * ntx;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int tx = (int)(idx % ntx);
        int64_t t = idx / ntx;
        const int ty = (int)(t % nty);
        t /= nty;
        const int co = (int)(t % Cout);
        const int ci = (int)(t / Cout);
        packed[idx] = w[(((int64_t)co * Cin + ci) * k + kh0y + s * ty) * k + kh0x + s * tx];
    }
}

}  // namespace qarig

using namespace qarig;

extern "C" int qarig_slab_reduce_f32(const float* slabs, float* out, int64_t ldc, int M, int N,
                                     int nslab, int accumulate, void* stream);

// Sum of the split launches' slabs (fixed order) + bias + activation, NCHW: element i belongs to output
// channel (i / plane) % Cout; 16 B per lane (plane % 4 == 0).
__global__ __launch_bounds__(256) void conv_split_reduce_kernel(const float* __restrict__ slabs, int nslab,
                                                                int64_t total, int plane, int Cout,
                                                                const float* __restrict__ bias, int act,
                                                                float* __restrict__ y, float* __restrict__ preact) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    float4 t = *reinterpret_cast<const float4*>(slabs + i);
    for (int z = 1; z < nslab; ++z) {
        const float4 u = *reinterpret_cast<const float4*>(slabs + (int64_t)z * total + i);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    if (bias) {
        const float b = bias[(i / plane) % Cout];
        t.x += b; t.y += b; t.z += b; t.w += b;
    }
    if (preact) *reinterpret_cast<float4*>(preact + i) = t;
    *reinterpret_cast<float4*>(y + i) = make_float4(act_fwd(t.x, act), act_fwd(t.y, act), act_fwd(t.z, act), act_fwd(t.w, act));
}

// Reduction splits of a ring launch of `wgs` workgroups over `nk` k-tiles: 1 when the launch fills the
// chip by itself, else enough to put two workgroups on every CU (measured at 4 images: the 512 -> 512
// layer at 32 x 32 runs 128 tiles; 355 us unsplit).  Parts are whole and at least 16 k-tiles long.
static int conv_ring_splits(long wgs, int nk) {
    if (wgs > 256) return 1;
    for (int s = 2; s <= 8; ++s)
        if (wgs * s >= 512 && nk % s == 0 && nk / s >= 16) return s;
    for (int s = 8; s >= 2; --s)
        if (nk % s == 0 && nk / s >= 16) return s;
    return 1;
}
static int launch_conv_split_reduce(const float* slabs, int splits, const ConvOut& o, int N, hipStream_t st) {
    const int plane = o.HoP * o.WoP;
    const int64_t total = (int64_t)N * o.Cout * plane;
    hipLaunchKernelGGL(conv_split_reduce_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, slabs,
                       splits, total, plane, o.Cout, o.bias, o.act, o.y, o.preact);
    QARIG_CHECK_LAUNCH("conv split reduce");
    return QARIG_OK;
}


static int launch_conv(const float* wmat, const ConvGeom& g, const ConvOut& o, hipStream_t st) {
    const bool fwd_taps = g.oy0 == -1 && g.ox0 == -1 && g.oys == 1 && g.oxs == 1;
    const bool flip_taps = g.oy0 == 1 && g.ox0 == 1 && g.oys == -1 && g.oxs == -1;
    const bool plain3x3 = g.stride == 1 && g.ntx == 3 && g.nty == 3 && (fwd_taps || flip_taps) &&
                          g.Wo == g.W && g.Ho == g.H && g.W % 4 == 0 &&
                          o.os == 1 && o.py == 0 && o.px == 0 && o.WoP == g.Wo && o.HoP == g.Ho &&
                          (((uintptr_t)g.x | (uintptr_t)o.y | (uintptr_t)o.preact) & 15) == 0;
    if (o.Cout <= 4 && plain3x3) {
        // fewer than four waves per CU: split the channels over the 8 waves of a workgroup
        const bool cs8 = (g.P / 4 + 63) / 64 < 1024 && g.C % 8 == 0 && g.C >= 64;
        dim3 grid((g.P / 4 + 63) / 64), block(cs8 ? 512 : 64);
        switch (o.Cout * 2 + (flip_taps ? 1 : 0)) {
#define QARIG_DC4(n)                                                                                             \
    case 2 * n:                                                                                                  \
        if (cs8) hipLaunchKernelGGL((conv_direct4_kernel<n, false, 8>), grid, block, 0, st, g, wmat, o);         \
        else hipLaunchKernelGGL((conv_direct4_kernel<n, false, 1>), grid, block, 0, st, g, wmat, o);             \
        break;                                                                                                   \
    case 2 * n + 1:                                                                                              \
        if (cs8) hipLaunchKernelGGL((conv_direct4_kernel<n, true, 8>), grid, block, 0, st, g, wmat, o);          \
        else hipLaunchKernelGGL((conv_direct4_kernel<n, true, 1>), grid, block, 0, st, g, wmat, o);              \
        break;
            QARIG_DC4(1) QARIG_DC4(2) QARIG_DC4(3) QARIG_DC4(4)
#undef QARIG_DC4
        }
    } else if (o.Cout <= 8) {
        dim3 grid((g.P + 255) / 256), block(256);
        switch (o.Cout) {
#define QARIG_DC(n) case n: hipLaunchKernelGGL((conv_direct_kernel<n>), grid, block, 0, st, g, wmat, o); break;
            QARIG_DC(1) QARIG_DC(2) QARIG_DC(3) QARIG_DC(4) QARIG_DC(5) QARIG_DC(6) QARIG_DC(7) QARIG_DC(8)
#undef QARIG_DC
        }
    } else {
        SrcKContig sa{wmat, (int64_t)g.K, o.Cout, g.K, 1.0f,
                      (((uintptr_t)wmat & 15) == 0) && g.K % 4 == 0};
        const int tiles_c = (o.Cout + BM - 1) / BM, tiles_p = (g.P + BN - 1) / BN;
        const bool rowvec = g.stride == 1 && g.Wo % 4 == 0 && g.K < 65536;
        const bool fast = sa.vec4 && o.Cout % BM == 0 && g.K % BK == 0 && g.P % BN == 0;
        const dim3 grid(tiles_c * tiles_p), block(NTHREADS);
        if (rowvec && fast)
            hipLaunchKernelGGL((conv_mma_kernel<SrcIm2colRow, true>), grid, block, 0, st, sa, g, o, tiles_p);
        else if (rowvec)
            hipLaunchKernelGGL((conv_mma_kernel<SrcIm2colRow, false>), grid, block, 0, st, sa, g, o, tiles_p);
        else if (fast)
            hipLaunchKernelGGL((conv_mma_kernel<SrcIm2col, true>), grid, block, 0, st, sa, g, o, tiles_p);
        else
            hipLaunchKernelGGL((conv_mma_kernel<SrcIm2col, false>), grid, block, 0, st, sa, g, o, tiles_p);
    }
    QARIG_CHECK_LAUNCH("conv");
    return QARIG_OK;
}

// 3x3 / stride 1 / padding 1 on the LDS-DMA ring (conv3x3_ring_kernel) where its tiles are whole:
// M (output channels of this product) % 128, C (reduced channels) % 16, pixels % 128, W % 4, the
// input below 2 GB (32-bit buffer offsets).  w element (m, c, tap) at w[m * sm + c * sc + tap];
// flip: the data gradient's tap order.  `packed`: M * 9 * C floats.  QARIG_CONV_RING=0 disables.
static bool conv3x3_ring_ok(int N, int C, int H, int W, int M, const void* x, const void* packed) {
    const bool on = g_qarig_opt.conv_ring != 0;          // option conv_ring = 0: the gather kernels (cross-check)
    const int64_t P = (int64_t)N * H * W, xb = P * C * 4;
    return on && packed && C % 16 == 0 && C >= 16 && M % BM == 0 && P % BN == 0 && W % 4 == 0 && xb < (1LL << 31) &&
           (int64_t)9 * C < (1 << 20) && (((uintptr_t)x | (uintptr_t)packed) & 15) == 0;
}
// `slabs` / `slab_bytes`: scratch behind the packed weights for a split launch (conv_ring_splits), else
// the launch is not split.
static int launch_conv3x3_ring(const float* w, int64_t sm, int64_t sc, int flip, const float* x, int N, int C,
                               int H, int W, int M, const ConvOut& o, float* packed, hipStream_t st,
                               float* slabs = nullptr, size_t slab_bytes = 0, bool packed_valid = false) {
    if (!packed_valid) {
        const int64_t total = (int64_t)M * 9 * C;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(conv_pack_tap_kernel, dim3(blocks), dim3(256), 0, st, w, M, C, sm, sc, flip, packed);
        QARIG_CHECK_LAUNCH("conv3x3 pack");
    }
    ConvGeom g{x, N, C, H, W, H, W, 1, 3, 3, -1, 1, -1, 1, 9 * C, N * H * W};
    const int tiles_c = M / BM, tiles_p = g.P / BN;
    const int64_t out_elems = (int64_t)N * M * H * W;
    int splits = conv_ring_splits((long)tiles_c * tiles_p, g.K / BK);
    if (splits > 1 && (!slabs || slab_bytes < (size_t)splits * out_elems * sizeof(float) ||
                       o.os != 1 || o.HoP != H || o.WoP != W || (((uintptr_t)o.y | (uintptr_t)o.preact) & 15)))
        splits = 1;
    if (splits > 1) {
        ConvOut raw{slabs, nullptr, nullptr, o.Cout, o.HoP, o.WoP, o.os, o.py, o.px, ACT_NONE};
        hipLaunchKernelGGL((conv3x3_ring_kernel<false>), dim3(tiles_c * tiles_p, 1, splits), dim3(NTHREADS), 0, st,
                           packed, g, raw, tiles_p, (unsigned)((int64_t)N * C * H * W * 4), (int64_t)0, 0, splits,
                           out_elems);
        QARIG_CHECK_LAUNCH("conv3x3 ring (split)");
        return launch_conv_split_reduce(slabs, splits, o, N, st);
    }
    hipLaunchKernelGGL((conv3x3_ring_kernel<false>), dim3(tiles_c * tiles_p), dim3(NTHREADS), 0, st, packed, g, o,
                       tiles_p, (unsigned)((int64_t)N * C * H * W * 4), (int64_t)0, 0, 1, (int64_t)0);
    QARIG_CHECK_LAUNCH("conv3x3 ring");
    return QARIG_OK;
}

// nn.Conv2d(Cin, Cout, k, stride, padding) + bias + activation, NCHW fp32.
// x (N,Cin,H,W); w (Cout,Cin,k,k); y (N,Cout,Ho,Wo), Ho = (H + 2p - k)/s + 1.
// preact (same shape as y) receives the pre-activation when non-null.
static int conv2d_fwd_impl(const float* x, int N, int Cin, int H, int W, const float* w,
                           const float* bias, int Cout, int k, int stride, int pad, int act,
                           float* y, float* preact, void* workspace, size_t ws_bytes, int flags, void* stream) {
    QARIG_CHECK_ARG(x && w && y, "conv2d: null pointer");
    QARIG_CHECK_ARG((flags & ~1) == 0, "conv2d: unknown flags %d", flags);
    const bool packed_valid = flags & 1;
    QARIG_CHECK_ARG(N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv2d: bad extents");
    QARIG_CHECK_ARG(k >= 1 && k <= 4 && stride >= 1 && stride <= 4 && pad >= 0 && pad <= 4,
                    "conv2d: kernel size 1..4, stride 1..4, padding 0..4 only");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cin, H, W}, 1LL << 20, 1LL << 31), "conv2d: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cout, H, W}, 1LL << 20, 1LL << 31), "conv2d: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({Cin, Cout}, 1LL << 20, 1LL << 31), "conv2d: extents too large");

    QARIG_CHECK_ARG(act >= 0 && act <= 3, "conv2d: bad activation id");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    QARIG_CHECK_ARG(Ho > 0 && Wo > 0, "conv2d: empty output");
    QARIG_CHECK_ARG((int64_t)N * Ho * Wo < INT32_MAX && (int64_t)Cin * k * k < INT32_MAX &&
                        (int64_t)Cin * H * W < INT32_MAX,
                    "conv2d: too large");
    ConvGeom g{x, N, Cin, H, W, Ho, Wo, stride, k, k, -pad, 1, -pad, 1, Cin * k * k, N * Ho * Wo};
    ConvOut o{y, preact, bias, Cout, Ho, Wo, 1, 0, 0, act};
    if (k == 3 && stride == 1 && pad == 1 && workspace &&
        ws_bytes >= (size_t)Cout * 9 * Cin * sizeof(float) &&
        conv3x3_ring_ok(N, Cin, H, W, Cout, x, workspace)) {
        const size_t packed_bytes = (size_t)Cout * 9 * Cin * sizeof(float);     // a multiple of 16 B
        return launch_conv3x3_ring(w, (int64_t)Cin * 9, 9, 0, x, N, Cin, H, W, Cout, o, (float*)workspace,
                                   (hipStream_t)stream, (float*)((char*)workspace + packed_bytes),
                                   ws_bytes - packed_bytes, packed_valid);
    }
    {   // stride 2: the same kernel with a strided im2col (four 4-B loads per k row)
        const int64_t P = (int64_t)N * Ho * Wo, xb = (int64_t)N * Cin * H * W * 4;
        if (g_qarig_opt.conv_ring != 0 && k == 3 && stride == 2 && pad == 1 && H == 2 * Ho && W == 2 * Wo && workspace &&
            ws_bytes >= (size_t)Cout * 9 * Cin * sizeof(float) && Cin % 16 == 0 && Cout % BM == 0 && P % BN == 0 &&
            Wo % 4 == 0 && xb < (1LL << 31) && (((uintptr_t)x | (uintptr_t)workspace) & 15) == 0) {
            hipStream_t st = (hipStream_t)stream;
            float* packed = (float*)workspace;
            const int64_t total = (int64_t)Cout * 9 * Cin;
            int blocks = (int)((total + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            if (!packed_valid) {
                hipLaunchKernelGGL(conv_pack_tap_kernel, dim3(blocks), dim3(256), 0, st, w, Cout, Cin, (int64_t)Cin * 9,
                                   (int64_t)9, 0, packed);
                QARIG_CHECK_LAUNCH("conv3x3 pack");
            }
            const int tiles_p = (int)(P / BN);
            const size_t packed_bytes = (size_t)Cout * 9 * Cin * sizeof(float);
            const int64_t out_elems = (int64_t)N * Cout * Ho * Wo;
            int splits = conv_ring_splits((long)(Cout / BM) * tiles_p, 9 * Cin / BK);
            if (splits > 1 && (ws_bytes < packed_bytes + (size_t)splits * out_elems * sizeof(float) ||
                               (((uintptr_t)y | (uintptr_t)preact) & 15)))
                splits = 1;
            if (splits > 1) {
                float* slabs = (float*)((char*)workspace + packed_bytes);
                ConvOut raw{slabs, nullptr, nullptr, Cout, Ho, Wo, 1, 0, 0, ACT_NONE};
                hipLaunchKernelGGL((conv3x3_ring_kernel<false, true>), dim3((Cout / BM) * tiles_p, 1, splits),
                                   dim3(NTHREADS), 0, st, packed, g, raw, tiles_p, (unsigned)xb, (int64_t)0, 0, splits,
                                   out_elems);
                QARIG_CHECK_LAUNCH("conv3x3 stride-2 ring (split)");
                return launch_conv_split_reduce(slabs, splits, o, N, st);
            }
            hipLaunchKernelGGL((conv3x3_ring_kernel<false, true>), dim3((Cout / BM) * tiles_p), dim3(NTHREADS), 0, st,
                               packed, g, o, tiles_p, (unsigned)xb, (int64_t)0, 0, 1, (int64_t)0);
            QARIG_CHECK_LAUNCH("conv3x3 stride-2 ring");
            return QARIG_OK;
        }
    }
    return launch_conv(w, g, o, (hipStream_t)stream);
}

extern "C" int qarig_conv2d_fwd(const float* x, int N, int Cin, int H, int W, const float* w,
                                const float* bias, int Cout, int k, int stride, int pad, int act,
                                float* y, float* preact, void* stream) {
    return conv2d_fwd_impl(x, N, Cin, H, W, w, bias, Cout, k, stride, pad, act, y, preact, nullptr, 0, 0, stream);
}

// The same with a scratch buffer (Cout * Cin * k * k floats) for a re-ordered copy of the weights:
// enables the LDS-DMA ring kernel on the 3x3 / stride 1 / padding 1 layers (conv3x3_ring_kernel).
extern "C" size_t qarig_conv2d_fwd_workspace_bytes(int Cin, int Cout, int k) {
    if (Cin < 1 || Cout < 1 || k < 1 || k > 4 || Cin > (1 << 20) || Cout > (1 << 20)) return 0;
    return (size_t)Cin * Cout * k * k * sizeof(float);
}
// ... and, at few images, room for the split launch's slabs (conv_ring_splits): the 3x3 / stride 1 / padding 1
// layers of a <= 8-image decoder run their reduction in 2-4 parts to fill the chip.
static size_t conv_split_slab_bytes(long wgs, int nk, int64_t out_elems) {
    const int s = conv_ring_splits(wgs, nk);
    return s > 1 ? (size_t)s * out_elems * sizeof(float) : 0;
}
extern "C" size_t qarig_conv2d_fwd_workspace_bytes_n(int N, int Cin, int H, int W, int Cout, int k, int stride) {
    const size_t base = qarig_conv2d_fwd_workspace_bytes(Cin, Cout, k);
    if (!base || N < 1 || H < 1 || W < 1 || !qarig_dims_ok({N, Cout, H, W}, 1LL << 20, 1LL << 31)) return base;
    if (k != 3 || (stride != 1 && stride != 2) || H % stride || W % stride) return base;
    const int Ho = H / stride, Wo = W / stride;          // (padding 1: the ring geometries)
    if (Cin % 16 || Cout % BM || ((int64_t)N * Ho * Wo) % BN) return base;
    return base + conv_split_slab_bytes((long)(Cout / BM) * ((int64_t)N * Ho * Wo / BN), 9 * Cin / BK,
                                        (int64_t)N * Cout * Ho * Wo);
}
extern "C" int qarig_conv2d_fwd_ws(const float* x, int N, int Cin, int H, int W, const float* w,
                                   const float* bias, int Cout, int k, int stride, int pad, int act,
                                   float* y, float* preact, void* workspace, size_t ws_bytes, int flags,
                                   void* stream) {
    return conv2d_fwd_impl(x, N, Cin, H, W, w, bias, Cout, k, stride, pad, act, y, preact, workspace, ws_bytes,
                           flags, stream);
}

extern "C" size_t qarig_conv_transpose2d_workspace_bytes(int Cin, int Cout) {
    if (Cin < 1 || Cout < 1 || Cin > (1 << 20) || Cout > (1 << 20)) return 0;
    return (size_t)16 * Cin * Cout * sizeof(float);
}
extern "C" size_t qarig_conv_transpose2d_workspace_bytes_n(int N, int Cin, int H, int W, int Cout) {
    const size_t base = qarig_conv_transpose2d_workspace_bytes(Cin, Cout);
    if (!base || N < 1 || H < 1 || W < 1 || !qarig_dims_ok({N, Cout, H, W, 4}, 1LL << 20, 1LL << 31)) return base;
    if (Cin % 16 || Cout % BM || ((int64_t)N * H * W) % BN) return base;
    return base + conv_split_slab_bytes(2L * (Cout / BM) * ((int64_t)N * H * W / BN), 4 * Cin / BK,
                                        (int64_t)N * Cout * H * W * 4);
}

// nn.ConvTranspose2d(Cin, Cout, 4, stride 2, padding 1) + bias + activation.
// x (N,Cin,H,W); w (Cin,Cout,4,4); y (N,Cout,2H,2W).
extern "C" int qarig_conv_transpose2d_fwd(const float* x, int N, int Cin, int H, int W,
                                          const float* w, const float* bias, int Cout, int act,
                                          float* y, float* preact, void* workspace,
                                          size_t ws_bytes, int flags, void* stream) {
    QARIG_CHECK_ARG(x && w && y, "conv_transpose2d: null pointer");
    QARIG_CHECK_ARG((flags & ~1) == 0, "conv_transpose2d: unknown flags %d", flags);
    const bool packed_valid = flags & 1;
    QARIG_CHECK_ARG(N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv_transpose2d: bad extents");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cin, H, W, 4}, 1LL << 20, 1LL << 31), "conv_transpose2d: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cout, H, W, 4}, 1LL << 20, 1LL << 31), "conv_transpose2d: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({Cin, Cout, 16}, 1LL << 20, 1LL << 31), "conv_transpose2d: extents too large");

    QARIG_CHECK_ARG(act >= 0 && act <= 3, "conv_transpose2d: bad activation id");
    if (!workspace || ws_bytes < qarig_conv_transpose2d_workspace_bytes(Cin, Cout)) {
        qarig_set_error("conv_transpose2d: workspace too small");
        return QARIG_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* packed = (float*)workspace;
    const int64_t total = (int64_t)16 * Cin * Cout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    // the four parity classes as 2x2-tap stride-1 products on the LDS-DMA ring, two column parities per workgroup
    if (conv3x3_ring_ok(N, Cin, H, W, Cout, x, packed) && (((uintptr_t)y | (uintptr_t)preact) & 7) == 0) {
        if (!packed_valid) {
            hipLaunchKernelGGL(convt_pack_tap_kernel, dim3(blocks), dim3(256), 0, st, w, Cin, Cout, packed);
            QARIG_CHECK_LAUNCH("conv_transpose2d pack");
        }
        const int K = Cin * 4, P = N * H * W;
        ConvGeom g{x, N, Cin, H, W, H, W, 1, 2, 2, 0, -1, 0, -1, K, P};
        ConvOut o{y, preact, bias, Cout, 2 * H, 2 * W, 2, 0, 0, act};
        const int tiles_p = P / BN;
        const size_t packed_bytes = qarig_conv_transpose2d_workspace_bytes(Cin, Cout);
        const int64_t out_elems = (int64_t)N * Cout * H * W * 4;
        int splits = conv_ring_splits(2L * (Cout / BM) * tiles_p, K / BK);
        if (splits > 1 && (ws_bytes < packed_bytes + (size_t)splits * out_elems * sizeof(float) ||
                           (((uintptr_t)y | (uintptr_t)preact) & 15) || W % 2))
            splits = 1;
        if (splits > 1) {
            float* slabs = (float*)((char*)workspace + packed_bytes);
            ConvOut raw{slabs, nullptr, nullptr, Cout, 2 * H, 2 * W, 2, 0, 0, ACT_NONE};
            hipLaunchKernelGGL((conv3x3_ring_kernel<true>), dim3((Cout / BM) * tiles_p, 2, splits), dim3(NTHREADS), 0,
                               st, packed, g, raw, tiles_p, (unsigned)((int64_t)N * Cin * H * W * 4),
                               (int64_t)Cout * K, 0, splits, out_elems);
            QARIG_CHECK_LAUNCH("conv_transpose2d ring (split)");
            return launch_conv_split_reduce(slabs, splits, o, N, st);
        }
        hipLaunchKernelGGL((conv3x3_ring_kernel<true>), dim3((Cout / BM) * tiles_p, 2), dim3(NTHREADS), 0, st, packed, g,
                           o, tiles_p, (unsigned)((int64_t)N * Cin * H * W * 4), (int64_t)Cout * K, 0, 1, (int64_t)0);
        QARIG_CHECK_LAUNCH("conv_transpose2d ring");
        return QARIG_OK;
    }
    if (!packed_valid) {
        hipLaunchKernelGGL(convt_pack_kernel, dim3(blocks), dim3(256), 0, st, w, Cin, Cout, packed);
        QARIG_CHECK_LAUNCH("conv_transpose2d pack");
    }
    // both column parities per launch (8-B stores) where the MFMA kernel applies and y / preact
    // rows are 8-B aligned; QARIG_CONVT_PAIR=0 restores one class per launch
    const bool pair_on = g_qarig_opt.convt_pair != 0;
    // ... and where one class per launch would leave the chip half empty (< 512 workgroups per class:
    // the 512 -> 256 layer at 16 images ran 61.7 TF that way, 88.4 TF paired; at 32 images one class
    // per launch is the faster form, 94.6 against 90.9 TF for the whole decoder)
    const long class_wgs = (long)((Cout + BM - 1) / BM) * (((long)N * H * W + BN - 1) / BN);
    const bool pair = pair_on && Cout > 8 && class_wgs < 512 && (((uintptr_t)y | (uintptr_t)preact) & 7) == 0;
    if (pair) {
        const int K = Cin * 4, P = N * H * W;
        const int tiles_c = (Cout + BM - 1) / BM, tiles_p = (P + BN - 1) / BN;
        const dim3 grid(tiles_c * tiles_p, 2), block(NTHREADS);      // y = row parity
        {
            ConvGeom g{x, N, Cin, H, W, H, W, 1, 2, 2, 0, -1, 0, -1, K, P};
            ConvOut o{y, preact, bias, Cout, 2 * H, 2 * W, 2, 0, 0, act};
            const float* wmat = packed;
            SrcKContig sa{wmat, (int64_t)K, Cout, K, 1.0f, (((uintptr_t)wmat & 15) == 0) && K % 4 == 0};
            const bool rowvec = W % 4 == 0 && K < 65536;
            const bool fast = sa.vec4 && Cout % BM == 0 && K % BK == 0 && P % BN == 0;
            const int64_t cs = (int64_t)Cout * K;
            if (rowvec && fast)
                hipLaunchKernelGGL((convt_pair_kernel<SrcIm2colRow, true>), grid, block, 0, st, sa, g, o, tiles_p, cs);
            else if (rowvec)
                hipLaunchKernelGGL((convt_pair_kernel<SrcIm2colRow, false>), grid, block, 0, st, sa, g, o, tiles_p, cs);
            else if (fast)
                hipLaunchKernelGGL((convt_pair_kernel<SrcIm2col, true>), grid, block, 0, st, sa, g, o, tiles_p, cs);
            else
                hipLaunchKernelGGL((convt_pair_kernel<SrcIm2col, false>), grid, block, 0, st, sa, g, o, tiles_p, cs);
            QARIG_CHECK_LAUNCH("conv_transpose2d pair");
        }
        return QARIG_OK;
    }
    for (int cls = 0; cls < 4; ++cls) {
        const int py = cls >> 1, px = cls & 1;
        // oy = 2a+py: tap th uses kh = 1-py+2th and input row a + (py + 1 - kh)/2 = a + py - th
        ConvGeom g{x, N, Cin, H, W, H, W, 1, 2, 2, py, -1, px, -1, Cin * 4, N * H * W};
        ConvOut o{y, preact, bias, Cout, 2 * H, 2 * W, 2, py, px, act};
        if (int e = launch_conv(packed + (int64_t)cls * Cout * Cin * 4, g, o, st)) return e;
    }
    return QARIG_OK;
}

// ------------------------------------------------------------------ backward entry points

extern "C" size_t qarig_conv2d_bwd_data_workspace_bytes(int Cin, int Cout, int k) {
    if (Cin < 1 || Cout < 1 || k < 1 || k > 4 || Cin > (1 << 20) || Cout > (1 << 20)) return 0;
    return (size_t)Cin * Cout * k * k * sizeof(float);
}

// ... plus room for a few-image launch's split slabs (3x3 / stride 1: the same ring launch as the forward)
extern "C" size_t qarig_conv2d_bwd_data_workspace_bytes_n(int N, int Cin, int H, int W, int Cout, int k, int stride) {
    const size_t base = qarig_conv2d_bwd_data_workspace_bytes(Cin, Cout, k);
    if (!base || N < 1 || H < 1 || W < 1 || !qarig_dims_ok({N, Cin, H, W}, 1LL << 20, 1LL << 31)) return base;
    if (k != 3 || stride != 1 || Cout % 16 || Cin % BM || ((int64_t)N * H * W) % BN) return base;
    return base + conv_split_slab_bytes((long)(Cin / BM) * ((int64_t)N * H * W / BN), 9 * Cout / BK,
                                        (int64_t)N * Cin * H * W);
}

// d(input) of Conv2d: dT (N,Cout,Ho,Wo) -> dx (N,Cin,H,W).  One stride-1 implicit-GEMM
// launch per output-parity class of dx (1 class for stride 1, 4 for stride 2), each with
// exactly the taps that hit it.
extern "C" int qarig_conv2d_bwd_data(const float* dT, int N, int Cout, int Ho, int Wo, const float* w,
                                     int Cin, int k, int stride, int pad, int H, int W, float* dx,
                                     void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(dT && w && dx, "conv2d_bwd_data: null pointer");
    QARIG_CHECK_ARG(k >= 1 && k <= 4 && stride >= 1 && stride <= 2 && pad >= 0 && pad < k,
                    "conv2d_bwd_data: unsupported geometry");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cout, Ho, Wo}, 1LL << 20, 1LL << 31), "conv2d_bwd_data: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cin, H, W}, 1LL << 20, 1LL << 31), "conv2d_bwd_data: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({Cin, Cout, 16}, 1LL << 20, 1LL << 31), "conv2d_bwd_data: extents too large");

    if (!workspace || ws_bytes < qarig_conv2d_bwd_data_workspace_bytes(Cin, Cout, k)) {
        qarig_set_error("conv2d_bwd_data: workspace too small");
        return QARIG_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* packed = (float*)workspace;
    const int s = stride;
    if (k == 3 && s == 1 && pad == 1 && Ho == H && Wo == W && conv3x3_ring_ok(N, Cout, H, W, Cin, dT, packed)) {
        // dx[ci] = sum_{co, tap} dT[co] at offset (1 - tap) * w[co][ci][tap]: a 3x3 conv over dT with M = Cin,
        // C = Cout and the taps flipped
        ConvOut o{dx, nullptr, nullptr, Cin, H, W, 1, 0, 0, ACT_NONE};
        const size_t packed_bytes = qarig_conv2d_bwd_data_workspace_bytes(Cin, Cout, k);
        return launch_conv3x3_ring(w, 9, (int64_t)Cin * 9, 1, dT, N, Cout, H, W, Cin, o, packed, st,
                                   (float*)((char*)workspace + packed_bytes), ws_bytes - packed_bytes);
    }
    if (k == 3 && s == 2 && pad == 1 && H == 2 * Ho && W == 2 * Wo && conv3x3_ring_ok(N, Cout, Ho, Wo, Cin, dT, packed) &&
        ((uintptr_t)dx & 7) == 0) {
        // the four output-parity classes of dx as (1 + ry) x (1 + rx)-tap stride-1 products over dT on the ring,
        // two column parities per workgroup (8-B stores), both row parities in one launch
        const int64_t total = (int64_t)9 * Cin * Cout;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(conv_s2_dgrad_pack_tap_kernel, dim3(blocks), dim3(256), 0, st, w, Cout, Cin, packed);
        QARIG_CHECK_LAUNCH("conv2d_bwd_data pack");
        const int P = N * Ho * Wo;
        ConvGeom g{dT, N, Cout, Ho, Wo, Ho, Wo, 1, 1, 1, 0, -1, 0, -1, Cout, P};
        ConvOut o{dx, nullptr, nullptr, Cin, H, W, 2, 0, 0, ACT_NONE};
        const int tiles_p = P / BN;
        hipLaunchKernelGGL((conv3x3_ring_kernel<true>), dim3((Cin / BM) * tiles_p, 2), dim3(NTHREADS), 0, st, packed, g,
                           o, tiles_p, (unsigned)((int64_t)N * Cout * Ho * Wo * 4), (int64_t)0, 1, 1, (int64_t)0);
        QARIG_CHECK_LAUNCH("conv2d_bwd_data ring");
        return QARIG_OK;
    }
    for (int ry = 0; ry < s; ++ry)
        for (int rx = 0; rx < s; ++rx) {
            const int gh = (H - ry + s - 1) / s, gw = (W - rx + s - 1) / s;
            if (gh <= 0 || gw <= 0) continue;
            const int ky0 = (ry + pad) % s, kx0 = (rx + pad) % s;
            const int nty = ky0 < k ? (k - ky0 + s - 1) / s : 0;
            const int ntx = kx0 < k ? (k - kx0 + s - 1) / s : 0;
            ConvGeom g{dT, N, Cout, Ho, Wo, gh, gw, 1, nty, ntx, (ry + pad - ky0) / s, -1,
                       (rx + pad - kx0) / s, -1, Cout * nty * ntx, N * gh * gw};
            ConvOut o{dx, nullptr, nullptr, Cin, H, W, s, ry, rx, ACT_NONE};
            if (nty == 0 || ntx == 0) return QARIG_ERR_ARG;  // cannot happen for pad < k
            const int64_t total = (int64_t)Cin * Cout * nty * ntx;
            int blocks = (int)((total + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(conv_bwd_pack_kernel, dim3(blocks), dim3(256), 0, st, w, Cout, Cin, k,
                               s, ky0, kx0, nty, ntx, packed);
            QARIG_CHECK_LAUNCH("conv2d_bwd_data pack");
            if (int e = launch_conv(packed, g, o, st)) return e;
        }
    return QARIG_OK;
}

// scratch of qarig_conv_transpose2d_bwd_data_ws: the tap-major weights + a few-image launch's split slabs
extern "C" size_t qarig_conv_transpose2d_bwd_data_workspace_bytes_n(int N, int Cin, int H, int W, int Cout) {
    const size_t base = qarig_conv_transpose2d_workspace_bytes(Cin, Cout);
    if (!base || N < 1 || H < 1 || W < 1 || !qarig_dims_ok({N, Cin, H, W}, 1LL << 20, 1LL << 31)) return base;
    if (Cout % 16 || Cin % BM || ((int64_t)N * H * W) % BN) return base;
    return base + conv_split_slab_bytes((long)(Cin / BM) * ((int64_t)N * H * W / BN), Cout, (int64_t)N * Cin * H * W);
}

// d(input) of ConvTranspose2d(4,2,1): dT (N,Cout,2H,2W) -> dx (N,Cin,H,W) is a
// Conv2d(k=4, stride 2, pad 1) over dT whose weight matrix [Cin][Cout*16] is the
// ConvTranspose weight exactly as stored.
static int convt_bwd_data_impl(const float* dT, int N, int Cout, int H, int W, const float* w, int Cin, float* dx,
                               void* workspace, size_t ws_bytes, void* stream);
extern "C" int qarig_conv_transpose2d_bwd_data(const float* dT, int N, int Cout, int H, int W,
                                               const float* w, int Cin, float* dx, void* stream) {
    return convt_bwd_data_impl(dT, N, Cout, H, W, w, Cin, dx, nullptr, 0, stream);
}
// The same with a scratch buffer (16 * Cin * Cout floats) for a tap-major copy of the weights: the strided
// ring kernel then serves the layers whose tiles are whole (Cout % 16, Cin % 128, N*H*W % 128, W % 4 == 0).
extern "C" int qarig_conv_transpose2d_bwd_data_ws(const float* dT, int N, int Cout, int H, int W, const float* w,
                                                  int Cin, float* dx, void* workspace, size_t ws_bytes, void* stream) {
    return convt_bwd_data_impl(dT, N, Cout, H, W, w, Cin, dx, workspace, ws_bytes, stream);
}
static int convt_bwd_data_impl(const float* dT, int N, int Cout, int H, int W, const float* w, int Cin, float* dx,
                               void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(dT && w && dx, "conv_transpose2d_bwd_data: null pointer");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cout, H, W, 4}, 1LL << 20, 1LL << 31), "conv_transpose2d_bwd_data: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cin, H, W}, 1LL << 20, 1LL << 31), "conv_transpose2d_bwd_data: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({Cin, Cout, 16}, 1LL << 20, 1LL << 31), "conv_transpose2d_bwd_data: extents too large");

    ConvGeom g{dT, N, Cout, 2 * H, 2 * W, H, W, 2, 4, 4, -1, 1, -1, 1, Cout * 16, N * H * W};
    ConvOut o{dx, nullptr, nullptr, Cin, H, W, 1, 0, 0, ACT_NONE};
    {   // Conv2d(4, stride 2, padding 1) over dT on the strided ring: 16 taps at offsets -1 .. 2
        const int64_t P = (int64_t)N * H * W, xb = (int64_t)N * Cout * 4 * H * W * 4;
        if (g_qarig_opt.conv_ring != 0 && workspace && ws_bytes >= (size_t)16 * Cin * Cout * sizeof(float) &&
            Cout % 16 == 0 && Cin % BM == 0 && P % BN == 0 && W % 4 == 0 && xb < (1LL << 31) &&
            (((uintptr_t)dT | (uintptr_t)workspace) & 15) == 0) {
            hipStream_t st = (hipStream_t)stream;
            float* packed = (float*)workspace;
            const int64_t total = (int64_t)Cin * 16 * Cout;
            int blocks = (int)((total + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            // w (Cin, Cout, 4, 4): element (m = ci, c = co, tap) at ci * Cout * 16 + co * 16 + tap
            hipLaunchKernelGGL(conv_pack_tap_kernel, dim3(blocks), dim3(256), 0, st, w, Cin, Cout, (int64_t)Cout * 16,
                               (int64_t)16, 0, packed, 16);
            QARIG_CHECK_LAUNCH("conv_transpose2d_bwd_data pack");
            const int tiles_p = (int)(P / BN);
            const size_t packed_bytes = (size_t)16 * Cin * Cout * sizeof(float);
            const int64_t out_elems = (int64_t)N * Cin * H * W;
            int splits = conv_ring_splits((long)(Cin / BM) * tiles_p, Cout);         // 16 * Cout / 16 k-tiles
            if (splits > 1 && (ws_bytes < packed_bytes + (size_t)splits * out_elems * sizeof(float) ||
                               ((uintptr_t)dx & 15)))
                splits = 1;
            if (splits > 1) {
                float* slabs = (float*)((char*)workspace + packed_bytes);
                ConvOut raw{slabs, nullptr, nullptr, Cin, H, W, 1, 0, 0, ACT_NONE};
                hipLaunchKernelGGL((conv3x3_ring_kernel<false, true, false>), dim3((Cin / BM) * tiles_p, 1, splits),
                                   dim3(NTHREADS), 0, st, packed, g, raw, tiles_p, (unsigned)xb, (int64_t)0, 0, splits,
                                   out_elems);
                QARIG_CHECK_LAUNCH("conv_transpose2d_bwd_data ring (split)");
                return launch_conv_split_reduce(slabs, splits, o, N, st);
            }
            hipLaunchKernelGGL((conv3x3_ring_kernel<false, true, false>), dim3((Cin / BM) * tiles_p), dim3(NTHREADS), 0,
                               st, packed, g, o, tiles_p, (unsigned)xb, (int64_t)0, 0, 1, (int64_t)0);
            QARIG_CHECK_LAUNCH("conv_transpose2d_bwd_data ring");
            return QARIG_OK;
        }
    }
    return launch_conv(w, g, o, (hipStream_t)stream);
}

namespace qarig {

// Weight gradient of the 3x3 / stride 1 / pad 1 layers with very few gradient channels
// (CG <= 4: the 256->3 pixel layer and the 512->4 latent layer).  On the MFMA kernel such a
// layer pays for a 128-row tile to fill 3 rows (3.5 ms at batch 16); here a block owns one
// input channel cx and a slice of the pixels, every lane walks groups of four consecutive
// pixels (three 6-float input rows, CG gradient float4s -> CG*9*4 FMAs) and the CG*9 sums
// are reduced across the block in a fixed order.  slabs[z][cg][cx*9 + kh*3 + kw].
template <int CG>
__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(WgradGeom g, int groups_per_split,
                                                                float* __restrict__ slabs) {
    __shared__ float red[4][CG * 9];
    const int cx = blockIdx.x;
    const int per = g.Gh * g.Gw, gpr = g.Gw >> 2;          // groups per row
    const int total = g.P >> 2;
    const int gbeg = blockIdx.y * groups_per_split;
    const int gend = min(total, gbeg + groups_per_split);
    float acc[CG][9];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.0f;
    for (int grp = gbeg + threadIdx.x; grp < gend; grp += 256) {
        const int rowid = grp / gpr;                        // n * Gh + y
        const int x = (grp - rowid * gpr) * 4;
        const int n = rowid / g.Gh, y = rowid - n * g.Gh;
        float4 gv[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c)
            gv[c] = *reinterpret_cast<const float4*>(g.G + ((int64_t)n * g.Cg + c) * per + y * g.Gw + x);
        const float* xp = g.X + ((int64_t)n * g.Cx + cx) * per;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = y + kh - 1;
            float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)g.H) {
                const float* row = xp + iy * g.W + x;
                const float4 m = *reinterpret_cast<const float4*>(row);
                v[1] = m.x; v[2] = m.y; v[3] = m.z; v[4] = m.w;
                if (x > 0) v[0] = row[-1];
                if (x + 4 < g.W) v[5] = row[4];
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    float t = acc[c][kh * 3 + kw];
                    t = fmaf(gv[c].x, v[kw], t);
                    t = fmaf(gv[c].y, v[kw + 1], t);
                    t = fmaf(gv[c].z, v[kw + 2], t);
                    t = fmaf(gv[c].w, v[kw + 3], t);
                    acc[c][kh * 3 + kw] = t;
                }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float sum = wave_sum(acc[c][t]);
            if (lane == 0) red[wave][c * 9 + t] = sum;
        }
    __syncthreads();
    if (threadIdx.x < CG * 9) {
        const int c = threadIdx.x / 9, t = threadIdx.x - c * 9;
        const float sum = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        slabs[((int64_t)blockIdx.y * g.Cg + c) * g.K2 + cx * 9 + t] = sum;
    }
}

}  // namespace qarig

static bool wgrad_direct_ok(int Cg, int Gh, int Gw, int H, int W, int k, int stride, int pad) {
    return Cg <= 4 && k == 3 && stride == 1 && pad == 1 && Gh == H && Gw == W && W % 4 == 0;
}
static int wgrad_direct_splits(int Cx, int P) {
    int s = (1024 + Cx - 1) / Cx;                      // ~4 blocks per CU
    const int maxs = (P / 4 + 1023) / 1024;            // at least 4 groups per thread
    if (s > maxs) s = maxs;
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}

static int wgrad_splits(int Cg, int K2, int P) {
    const int tiles = ((Cg + BM - 1) / BM) * ((K2 + BN - 1) / BN);
    int s = (768 + tiles - 1) / tiles;
    const int maxs = (P + 4 * BK - 1) / (4 * BK);
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

extern "C" size_t qarig_conv_wgrad_workspace_bytes(int Cg, int K2, int P) {
    if (!qarig_dims_ok({Cg, K2}, 1LL << 24, 1LL << 34) || P < 1 || P > (1 << 30)) return 0;
    const int s = wgrad_splits(Cg, K2, P);
    return (size_t)(Cg <= 4 && s < 64 ? 64 : s) * Cg * K2 * sizeof(float);   // direct path: <= 64 slabs
}

// Weight gradient of Conv2d (G = dT, X = input -> dw (Cout,Cin,k,k)) and of
// ConvTranspose2d(4,2,1) (G = input, X = dT, k=4, stride=2, pad=1 -> dw (Cin,Cout,4,4)).
// G: (N,Cg,Gh,Gw); X: (N,Cx,H,W); dw: (Cg, Cx*k*k) row-major.
extern "C" int qarig_conv_wgrad(const float* G, int N, int Cg, int Gh, int Gw, const float* X,
                                int Cx, int H, int W, int k, int stride, int pad, float* dw,
                                void* workspace, size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(G && X && dw, "conv_wgrad: null pointer");
    QARIG_CHECK_ARG(N > 0 && Cg > 0 && Cx > 0 && k >= 1 && k <= 4, "conv_wgrad: bad extents");
    QARIG_CHECK_ARG(stride >= 1 && stride <= 2 && pad >= 0 && pad <= 4, "conv_wgrad: unsupported geometry");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cg, Gh, Gw}, 1LL << 20, 1LL << 31), "conv_wgrad: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({N, Cx, H, W}, 1LL << 20, 1LL << 31), "conv_wgrad: extents too large");
    QARIG_CHECK_ARG(qarig_dims_ok({Cg, Cx, 16}, 1LL << 20, 1LL << 31), "conv_wgrad: extents too large");

    QARIG_CHECK_ARG((int64_t)N * Gh * Gw < INT32_MAX, "conv_wgrad: too many pixels");
    WgradGeom g{G, Cg, Gh, Gw, X, Cx, H, W, N, k, stride, pad, N * Gh * Gw, Cx * k * k};
    if (!workspace || ws_bytes < qarig_conv_wgrad_workspace_bytes(Cg, g.K2, g.P)) {
        qarig_set_error("conv_wgrad: workspace too small");
        return QARIG_ERR_WORKSPACE;
    }
    hipStream_t st0 = (hipStream_t)stream;
    if (wgrad_direct_ok(Cg, Gh, Gw, H, W, k, stride, pad) && ((((uintptr_t)G | (uintptr_t)X)) & 15) == 0) {
        const int ns = wgrad_direct_splits(Cx, g.P);
        const int gps = (g.P / 4 + ns - 1) / ns;
        const dim3 dgrid(Cx, ns), dblock(256);
        switch (Cg) {
            case 1: hipLaunchKernelGGL((conv_wgrad_direct_kernel<1>), dgrid, dblock, 0, st0, g, gps, (float*)workspace); break;
            case 2: hipLaunchKernelGGL((conv_wgrad_direct_kernel<2>), dgrid, dblock, 0, st0, g, gps, (float*)workspace); break;
            case 3: hipLaunchKernelGGL((conv_wgrad_direct_kernel<3>), dgrid, dblock, 0, st0, g, gps, (float*)workspace); break;
            default: hipLaunchKernelGGL((conv_wgrad_direct_kernel<4>), dgrid, dblock, 0, st0, g, gps, (float*)workspace); break;
        }
        QARIG_CHECK_LAUNCH("conv_wgrad direct");
        return qarig_slab_reduce_f32((const float*)workspace, dw, g.K2, Cg, g.K2, ns, 0, stream);
    }
    const int splits = wgrad_splits(Cg, g.K2, g.P);
    {   // 3x3 / stride 1 / padding 1 with whole tiles: the ring kernel (QARIG_CONV_RING=0 disables)
        const int64_t gb = (int64_t)N * Cg * Gh * Gw * 4, xb = (int64_t)N * Cx * H * W * 4;
        if (g_qarig_opt.conv_ring != 0 && k == 3 && stride == 1 && pad == 1 && Gh == H && Gw == W && W % BK == 0 &&
            Cg % BM == 0 && Cx % BN == 0 && gb < (1LL << 31) && xb < (1LL << 31) && (int64_t)H * W < (1 << 22) &&
            ((((uintptr_t)G | (uintptr_t)X)) & 15) == 0) {
            const int ktiles = g.P / BK;
            const int perk = (ktiles + splits - 1) / splits;
            const int nz = (ktiles + perk - 1) / perk;
            const int tiles_cx = Cx / BN;
            const dim3 rgrid((Cg / BM) * tiles_cx * 9, 1, nz), rblock(NTHREADS);
            hipLaunchKernelGGL(conv3x3_wgrad_ring_kernel, rgrid, rblock, 0, (hipStream_t)stream, g, tiles_cx, perk,
                               ktiles, (unsigned)gb, (unsigned)xb, (float*)workspace);
            QARIG_CHECK_LAUNCH("conv_wgrad ring");
            return qarig_slab_reduce_f32((const float*)workspace, dw, g.K2, Cg, g.K2, nz, 0, stream);
        }
    }
    const int per = ((g.P + splits - 1) / splits + BK - 1) / BK * BK;
    const int nsp = (g.P + per - 1) / per;
    const int tiles_m = (Cg + BM - 1) / BM, tiles_n = (g.K2 + BN - 1) / BN;
    hipStream_t st = (hipStream_t)stream;
    const bool row = Gw % 8 == 0 && (((uintptr_t)G) & 15) == 0;
    const bool fast = row && Cg % BM == 0 && g.K2 % BN == 0 && per % BK == 0 && g.P % BK == 0;
    const dim3 grid(tiles_m * tiles_n, 1, nsp), block(NTHREADS);
    if (fast)
        hipLaunchKernelGGL((conv_wgrad_kernel<true, true>), grid, block, 0, st, g, tiles_n, per, (float*)workspace);
    else if (row)
        hipLaunchKernelGGL((conv_wgrad_kernel<true, false>), grid, block, 0, st, g, tiles_n, per, (float*)workspace);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<false, false>), grid, block, 0, st, g, tiles_n, per, (float*)workspace);
    QARIG_CHECK_LAUNCH("conv_wgrad");
    return qarig_slab_reduce_f32((const float*)workspace, dw, g.K2, Cg, g.K2, nsp, 0, stream);
}

// db[c] = sum_{n,y,x} G[n][c][y][x]  (bias gradient of both conv kinds).
extern "C" int qarig_conv_bias_grad(const float* G, int N, int C, int HW, float* db, void* stream) {
    QARIG_CHECK_ARG(G && db && N > 0 && C > 0 && HW > 0, "conv_bias_grad: bad arguments");
    QARIG_CHECK_ARG(qarig_dims_ok({N, C, HW}, 1LL << 20, 1LL << 31), "conv_bias_grad: extents too large");

    hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, G, N, C, HW,
                       db);
    QARIG_CHECK_LAUNCH("conv_bias_grad");
    return QARIG_OK;
}
