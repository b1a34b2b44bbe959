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

namespace qarig {

struct ConvGeom {
    const float* x;
    int N, C, H, W;        // input tensor
    int Ho, Wo;            // logical output grid of this launch
    int stride;            // input step per logical output step
    int nty, ntx;          // taps
    int offy[4], offx[4];  // input offset per tap
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
struct SrcIm2col {
    ConvGeom g;
    int64_t base;   // n * C*H*W
    int iy0, ix0;   // oy*stride, ox*stride
    bool valid;

    __device__ __forceinline__ void init(int x0, int tid) {
        const int p = x0 + (tid & 127);
        valid = p < g.P;
        const int per = g.Ho * g.Wo;
        const int n = valid ? p / per : 0;
        const int rem = valid ? p - n * per : 0;
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        base = (int64_t)n * g.C * g.H * g.W;
        iy0 = oy * g.stride;
        ix0 = ox * g.stride;
    }
    __device__ __forceinline__ bool interior(int, int, int) const { return false; }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        load(r, x0, k0, tid);
    }
    __device__ __forceinline__ void load(float (&r)[STAGE], int, int k0, int tid) const {
        int k = k0 + (tid >> 7) * 8;
        int tx = k % g.ntx;
        int t = k / g.ntx;
        int ty = t % g.nty;
        int ci = t / g.nty;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float v = 0.0f;
            if (valid && k + q < g.K) {
                const int iy = iy0 + g.offy[ty], ix = ix0 + g.offx[tx];
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                    v = g.x[base + ((int64_t)ci * g.H + iy) * g.W + ix];
            }
            r[q] = v;
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

__global__ __launch_bounds__(NTHREADS, 2) void conv_mma_kernel(SrcKContig sa, ConvGeom g,
                                                               ConvOut o, int tiles_p) {
    __shared__ __attribute__((aligned(16))) float lds[GEMM_LDS_FLOATS];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tc = tile / tiles_p, tp = tile - tc * tiles_p;   // pixel tile fastest
    const int c0 = tc * BM, p0 = tp * BN;
    SrcIm2col sb;
    sb.g = g;
    sb.init(p0, threadIdx.x);
    Acc acc;
    acc_zero(acc);
    contract_loop<false>(acc, sa, sb, c0, p0, 0, g.K, lds);

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
            const int iy = oy * g.stride + g.offy[ty];
            for (int tx = 0; tx < g.ntx; ++tx) {
                const int ix = ox * g.stride + g.offx[tx];
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

}  // namespace qarig

using namespace qarig;

static int launch_conv(const float* wmat, const ConvGeom& g, const ConvOut& o, hipStream_t st) {
    if (o.Cout <= 8) {
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
        hipLaunchKernelGGL(conv_mma_kernel, dim3(tiles_c * tiles_p), dim3(NTHREADS), 0, st, sa, g, o,
                           tiles_p);
    }
    QARIG_CHECK_LAUNCH("conv");
    return QARIG_OK;
}

// nn.Conv2d(Cin, Cout, k, stride, padding) + bias + activation, NCHW fp32.
// x (N,Cin,H,W); w (Cout,Cin,k,k); y (N,Cout,Ho,Wo), Ho = (H + 2p - k)/s + 1.
// preact (same shape as y) receives the pre-activation when non-null.
extern "C" int qarig_conv2d_fwd(const float* x, int N, int Cin, int H, int W, const float* w,
                                const float* bias, int Cout, int k, int stride, int pad, int act,
                                float* y, float* preact, void* stream) {
    QARIG_CHECK_ARG(x && w && y, "conv2d: null pointer");
    QARIG_CHECK_ARG(N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv2d: bad extents");
    QARIG_CHECK_ARG(k >= 1 && k <= 4 && stride >= 1 && pad >= 0, "conv2d: kernel size 1..4 only");
    QARIG_CHECK_ARG(act >= 0 && act <= 3, "conv2d: bad activation id");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    QARIG_CHECK_ARG(Ho > 0 && Wo > 0, "conv2d: empty output");
    QARIG_CHECK_ARG((int64_t)N * Ho * Wo < INT32_MAX && (int64_t)Cin * k * k < INT32_MAX,
                    "conv2d: too large");
    ConvGeom g{x, N, Cin, H, W, Ho, Wo, stride, k, k, {0, 0, 0, 0}, {0, 0, 0, 0}, Cin * k * k,
               N * Ho * Wo};
    for (int t = 0; t < k; ++t) g.offy[t] = g.offx[t] = t - pad;
    ConvOut o{y, preact, bias, Cout, Ho, Wo, 1, 0, 0, act};
    return launch_conv(w, g, o, (hipStream_t)stream);
}

extern "C" size_t qarig_conv_transpose2d_workspace_bytes(int Cin, int Cout) {
    return (size_t)16 * Cin * Cout * sizeof(float);
}

// nn.ConvTranspose2d(Cin, Cout, 4, stride 2, padding 1) + bias + activation.
// x (N,Cin,H,W); w (Cin,Cout,4,4); y (N,Cout,2H,2W).
extern "C" int qarig_conv_transpose2d_fwd(const float* x, int N, int Cin, int H, int W,
                                          const float* w, const float* bias, int Cout, int act,
                                          float* y, float* preact, void* workspace,
                                          size_t ws_bytes, void* stream) {
    QARIG_CHECK_ARG(x && w && y, "conv_transpose2d: null pointer");
    QARIG_CHECK_ARG(N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv_transpose2d: bad extents");
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
    hipLaunchKernelGGL(convt_pack_kernel, dim3(blocks), dim3(256), 0, st, w, Cin, Cout, packed);
    QARIG_CHECK_LAUNCH("conv_transpose2d pack");
    for (int cls = 0; cls < 4; ++cls) {
        const int py = cls >> 1, px = cls & 1;
        ConvGeom g{x, N, Cin, H, W, H, W, 1, 2, 2, {0, 0, 0, 0}, {0, 0, 0, 0}, Cin * 4, N * H * W};
        // oy = 2a+py: tap th uses kh = 1-py+2th and input row a + (py + 1 - kh)/2
        g.offy[0] = py;      g.offy[1] = py - 1;
        g.offx[0] = px;      g.offx[1] = px - 1;
        ConvOut o{y, preact, bias, Cout, 2 * H, 2 * W, 2, py, px, act};
        if (int e = launch_conv(packed + (int64_t)cls * Cout * Cin * 4, g, o, st)) return e;
    }
    return QARIG_OK;
}
