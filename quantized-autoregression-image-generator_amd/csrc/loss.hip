// Cross-entropy (mean over rows) forward fused with its gradient w.r.t. the logits:
// nn.CrossEntropyLoss() as the reference trains with
// (train_quantized_transformer.py:337,496-502).  One wave per row.
#include "qarig_common.h"

namespace qarig {

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits,
                                                      const int64_t* __restrict__ target, int M,
                                                      int C, float inv_m,
                                                      float* __restrict__ row_loss,
                                                      float* __restrict__ dlogits,
                                                      int* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* x = logits + (int64_t)row * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s += expf(x[c] - mx);
    s = wave_sum(s);
    const float lse = mx + logf(s);
    const int64_t t = target[row];
    const bool ok = t >= 0 && t < C;
    if (lane == 0) {
        if (!ok) atomicExch(bad, 1);
        row_loss[row] = ok ? lse - x[t] : 0.0f;
    }
    if (dlogits) {
        float* d = dlogits + (int64_t)row * C;
        for (int c = lane; c < C; c += 64) {
            const float p = expf(x[c] - lse);
            d[c] = (p - ((int64_t)c == t ? 1.0f : 0.0f)) * inv_m;
        }
    }
}

// mean of row_loss[M], fixed order: 256 strided partial sums, then a tree.
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int M, float inv_m,
                                                   float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < M; i += 256) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * inv_m;
}

// F.mse_loss(pred, target) (mean) and d/dpred = 2 (pred - target) / n, fixed-order sum.
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ b, int64_t n,
                                                          float inv_n, float* __restrict__ part,
                                                          float* __restrict__ da) {
    __shared__ float red[256];
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = a[i] - b[i];
        s = fmaf(d, d, s);
        if (da) da[i] = 2.0f * d * inv_n;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

}  // namespace qarig

using namespace qarig;

constexpr int MSE_BLOCKS = 1024;

extern "C" size_t qarig_mse_workspace_bytes(void) { return MSE_BLOCKS * sizeof(float); }

// F.mse_loss (train_autoencoder.py:215-217, train_codebook.py:233-235): loss (1 float),
// dpred (n floats or NULL).  part_ws: qarig_mse_workspace_bytes().
extern "C" int qarig_mse_fwd(const float* pred, const float* target, int64_t n, float* loss,
                             float* dpred, float* part_ws, void* stream) {
    QARIG_CHECK_ARG(pred && target && loss && part_ws && n > 0 && n <= (1LL << 40), "mse: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const float inv_n = 1.0f / (float)n;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(MSE_BLOCKS), dim3(256), 0, st, pred, target, n,
                       inv_n, part_ws, dpred);
    QARIG_CHECK_LAUNCH("mse partial");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, part_ws, MSE_BLOCKS, inv_n, loss);
    QARIG_CHECK_LAUNCH("mse mean");
    return QARIG_OK;
}

// logits (M,C) fp32, target int64 (M,).  loss: 1 float.  dlogits (M,C) or NULL =
// d(mean CE)/d(logits).  row_ws: M floats of scratch.  *bad_flag set on a target
// outside [0,C).
extern "C" int qarig_cross_entropy_fwd(const float* logits, const int64_t* target, int M, int C,
                                       float* loss, float* dlogits, float* row_ws, int* bad_flag,
                                       void* stream) {
    QARIG_CHECK_ARG(logits && target && loss && row_ws && bad_flag && M > 0 && C > 0,
                    "cross_entropy: bad arguments");
    QARIG_CHECK_DIMS("cross_entropy", M, C);
    hipStream_t st = (hipStream_t)stream;
    const float inv_m = 1.0f / (float)M;
    hipLaunchKernelGGL(ce_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, st, logits, target, M, C,
                       inv_m, row_ws, dlogits, bad_flag);
    QARIG_CHECK_LAUNCH("cross_entropy rows");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_ws, M, inv_m, loss);
    QARIG_CHECK_LAUNCH("cross_entropy mean");
    return QARIG_OK;
}
