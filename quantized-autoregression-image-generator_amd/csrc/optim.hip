// Adam step over one flat fp32 buffer (all parameters of a model live in one
// allocation; so do their gradients and moments): one launch per step, and the same
// flat gradient buffer is what the DP all-reduce moves.  Semantics of
// torch.optim.Adam(lr, betas) without weight decay / amsgrad
// (train_quantized_transformer.py:317-320; train_autoencoder.py:133-136).
#include "qarig_common.h"

namespace qarig {

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, int64_t n, float beta1,
                            float beta2, float eps, float step_size, float bc2_sqrt,
                            float grad_scale, const float* __restrict__ dev_step,
                            unsigned short* __restrict__ shadow) {
    if (dev_step) {     // captured-graph replay: the per-step scalars live in device memory
        step_size = dev_step[0];
        bc2_sqrt = dev_step[1];
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        // exp_avg.lerp_(grad, 1 - beta1), torch's two-branch lerp
        const float w = 1.0f - beta1;
        const float mi = m[i];
        const float mn = w < 0.5f ? mi + w * (gi - mi) : gi - (gi - mi) * (1.0f - w);
        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float vn = v[i] * beta2 + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vn) / bc2_sqrt + eps;
        const float pn = p[i] - step_size * (mn / denom);
        p[i] = pn;
        m[i] = mn;
        v[i] = vn;
        if (shadow) {       // reduced-precision mode: the bf16 copy the GEMMs read, from the same pass
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t f = {pn, 0.0f};
            shadow[i] = (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t)) & 0xffffu);
        }
    }
}

}  // namespace qarig

using namespace qarig;

// step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t), both computed by the
// host in double as torch does.  grad_scale multiplies g first (1/world after a sum
// all-reduce; 1 otherwise).  dev_step (optional, device, 2 floats {step_size, bc2_sqrt}) overrides
// the two per-step scalars, so that a captured graph of the training step can be replayed while
// the host refreshes them between replays.  shadow_bf16 (optional, n bf16 values): receives the updated
// parameters rounded to nearest-even bf16 -- the operands of the reduced-precision GEMMs -- so that no
// per-weight cast launch follows an optimiser step.
extern "C" int qarig_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float beta1,
                               float beta2, float eps, float step_size, float bc2_sqrt,
                               float grad_scale, const float* dev_step, void* shadow_bf16, void* stream) {
    QARIG_CHECK_ARG(p && g && m && v && n > 0 && n <= (1LL << 40), "adam: bad arguments");
    int64_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       beta1, beta2, eps, step_size, bc2_sqrt, grad_scale, dev_step, (unsigned short*)shadow_bf16);
    QARIG_CHECK_LAUNCH("adam");
    return QARIG_OK;
}
