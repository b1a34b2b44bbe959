// Microbenchmark: can packed-fp32 VALU FMAs run beside fp32 MFMAs on the same SIMD?
// mode 0: every wave MFMA; mode 1: every wave VALU pk_fma; mode 2: even waves MFMA, odd waves VALU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(int mode, int iters, int viters, float* out) {
    const int wave = threadIdx.x >> 6;
    const int half = blockDim.x >> 7;   // first half of the waves: MFMA, second half: VALU (same SIMDs)
    const bool mfma = mode == 0 || (mode == 2 && wave < half);
    const bool valu = mode == 1 || (mode == 2 && wave >= half);
    float r = 0.f;
    if (mfma) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = threadIdx.x * 1e-3f, y = 1.0f;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    }
    if (valu) {
        f32x2 c[16];
        for (int j = 0; j < 16; ++j) c[j] = f32x2{0.f, 0.f};
        f32x2 a = {threadIdx.x * 1e-3f, 1.0f}, b = {1.0001f, 0.9999f};
        for (int i = 0; i < viters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
        }
        for (int j = 0; j < 16; ++j) r += c[j][0] + c[j][1];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 512 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpb : {4, 8}) for (int mode = 0; mode < 3; ++mode) for (int vmul : {2, 3}) {
        if (mode != 2 && vmul != 2) continue;
        const int viters = iters * vmul;
        // one block per CU, wpb waves per block (4 -> one wave per SIMD, 8 -> two per SIMD)
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * wpb), 0, 0, mode, 100, 100, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * wpb), 0, 0, mode, iters, viters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double nm = (mode == 0 ? wpb : mode == 2 ? wpb / 2 : 0), nv = (mode == 1 ? wpb : mode == 2 ? wpb / 2 : 0);
        double mf = 256.0 * nm * iters * 4 * (32 * 32 * 2 * 2.0) / (ms * 1e-3) / 1e12;
        double vf = 256.0 * nv * viters * 16 * (64 * 2 * 2.0) / (ms * 1e-3) / 1e12;
        printf("waves/block %d mode %d vmul %d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  total %.1f TF\n", wpb, mode, vmul, ms, mf, vf, mf + vf);
    }
    return 0;
}
