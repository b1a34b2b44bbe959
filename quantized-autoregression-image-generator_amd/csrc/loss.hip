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

}  // namespace qarig

using namespace qarig;

// logits (M,C) fp32, target int64 (M,).  loss: 1 float.  dlogits (M,C) or NULL =
// d(mean CE)/d(logits).  row_ws: M floats of scratch.  *bad_flag set on a target
// outside [0,C).
extern "C" int qarig_cross_entropy_fwd(const float* logits, const int64_t* target, int M, int C,
                                       float* loss, float* dlogits, float* row_ws, int* bad_flag,
                                       void* stream) {
    QARIG_CHECK_ARG(logits && target && loss && row_ws && bad_flag && M > 0 && C > 0,
                    "cross_entropy: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const float inv_m = 1.0f / (float)M;
    hipLaunchKernelGGL(ce_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, st, logits, target, M, C,
                       inv_m, row_ws, dlogits, bad_flag);
    QARIG_CHECK_LAUNCH("cross_entropy rows");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_ws, M, inv_m, loss);
    QARIG_CHECK_LAUNCH("cross_entropy mean");
    return QARIG_OK;
}
