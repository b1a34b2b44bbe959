// Small HBM-bound elementwise kernels of the hot path (gating multiply of
// ResidualLinearLayer, models/layers.py:293-295, and its backward; activation
// backward; scaling by a device scalar).
#include "qarig_common.h"

namespace qarig {

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                           float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a[i] * b[i];
}

// da = dy*b, db = dy*a
__global__ void mul_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                               const float* __restrict__ b, float* __restrict__ da,
                               float* __restrict__ db, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float d = dy[i];
        da[i] = d * b[i];
        db[i] = d * a[i];
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                               float* __restrict__ dz, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dz[i] = dy[i] * act_grad(z[i], act);
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                               int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        y[i] = act_fwd(x[i], act);
}

__global__ void scale_by_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                float* __restrict__ y, int64_t n) {
    const float f = s[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i] * f;
}

}  // namespace qarig

using namespace qarig;

static dim3 ew_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return dim3((unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)));
}

extern "C" int qarig_mul_fwd(const float* a, const float* b, float* y, int64_t n, void* stream) {
    QARIG_CHECK_ARG(a && b && y && n > 0 && n <= (1LL << 40), "mul_fwd: bad arguments");
    hipLaunchKernelGGL(mul_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    QARIG_CHECK_LAUNCH("mul_fwd");
    return QARIG_OK;
}

extern "C" int qarig_mul_bwd(const float* dy, const float* a, const float* b, float* da, float* db,
                             int64_t n, void* stream) {
    QARIG_CHECK_ARG(dy && a && b && da && db && n > 0 && n <= (1LL << 40), "mul_bwd: bad arguments");
    hipLaunchKernelGGL(mul_bwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, dy, a, b, da,
                       db, n);
    QARIG_CHECK_LAUNCH("mul_bwd");
    return QARIG_OK;
}

extern "C" int qarig_act_fwd(const float* x, float* y, int64_t n, int act, void* stream) {
    QARIG_CHECK_ARG(x && y && n > 0 && n <= (1LL << 40) && act >= 0 && act <= 3, "act_fwd: bad arguments");
    hipLaunchKernelGGL(act_fwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, x, y, n, act);
    QARIG_CHECK_LAUNCH("act_fwd");
    return QARIG_OK;
}

// dz = dy * act'(z), z the pre-activation.
extern "C" int qarig_act_bwd(const float* dy, const float* z, float* dz, int64_t n, int act,
                             void* stream) {
    QARIG_CHECK_ARG(dy && z && dz && n > 0 && n <= (1LL << 40) && act >= 0 && act <= 3, "act_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, dy, z, dz, n,
                       act);
    QARIG_CHECK_LAUNCH("act_bwd");
    return QARIG_OK;
}

// y = x * s[0], s a device scalar (upstream gradient of a scalar loss).
extern "C" int qarig_scale_by(const float* x, const float* s, float* y, int64_t n, void* stream) {
    QARIG_CHECK_ARG(x && s && y && n > 0 && n <= (1LL << 40), "scale_by: bad arguments");
    hipLaunchKernelGGL(scale_by_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, x, s, y, n);
    QARIG_CHECK_LAUNCH("scale_by");
    return QARIG_OK;
}
