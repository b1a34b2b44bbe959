// fp32-accurate GEMM on the bf16 matrix pipe -- an accuracy probe.
// Every fp32 operand is split into three bf16 pieces that add up to it exactly (h + m + l, as the BMU coarse pass does);
// x.w = hh + hm + mh + hl + lh + mm (+ three dropped products <= 3 x 2^-24 |x||w|), six v_mfma_f32_32x32x16_bf16 per
// 16-deep k-step with fp32 accumulation.  The question a GEMM built this way must answer first: does it pass the
// fp32 parity tolerance of the Linear products (tests/test_gpu_core.py: max|err| / max|ref| < 2e-6 sqrt(K / 512)
// against fp64) that the fp32-MFMA kernels pass?  One wave computes one 32 x 32 output tile both ways:
//   A  v_mfma_f32_32x32x2_f32 over k ascending (the arithmetic of the product kernels: an exact fp32 fma chain)
//   B  the six bf16 products per k-step, one accumulator (small products first)
//   C  two accumulators: hh in one, the five small products in the other, added at the end
// and prints max|err| / max|ref| and the rms error against the fp64 contraction for K = 512 ... 8192 on N(0,1)
// operands and on operands of mixed magnitude.
//   hipcc --offload-arch=gfx950 -O2 tools/gemm_x3_probe.hip -o scratch/gemm_x3_probe && scratch/gemm_x3_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf16_round(float v) {          // round to nearest even to 8 significant bits
    const __bf16 b = (__bf16)v;
    return (float)b;
}
__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    const float fh = bf16_round(v);
    const float r = v - fh;
    const float fm = bf16_round(r);
    const float fl = bf16_round(r - fm);
    h = (__bf16)fh; m = (__bf16)fm; l = (__bf16)fl;
}

// X, W: [32][K] row-major.  out: [3][32][32] (A, B, C), out[v][i][j] = sum_k X[i][k] W[j][k].
__global__ void probe(const float* __restrict__ X, const float* __restrict__ W, int K, float* __restrict__ out) {
    const int lane = threadIdx.x, r = lane & 31, hf = lane >> 5;
    f32x16 accA, accB, accH, accS;
    for (int q = 0; q < 16; ++q) { accA[q] = 0.f; accB[q] = 0.f; accH[q] = 0.f; accS[q] = 0.f; }
    // A: 32x32x2 f32: lane (row r, k-half hf) holds X[r][k0 + hf] / W[r][k0 + hf]
    for (int k0 = 0; k0 < K; k0 += 2)
        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(X[r * K + k0 + hf], W[r * K + k0 + hf], accA, 0, 0, 0);
    // B / C: 32x32x16 bf16: lane (row r, k-half hf) holds 8 consecutive k: k0 + 8 hf + j
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 xh, xm, xl, wh, wm, wl;
        for (int j = 0; j < 8; ++j) {
            __bf16 a, b, c;
            split3(X[r * K + k0 + 8 * hf + j], a, b, c); xh[j] = a; xm[j] = b; xl[j] = c;
            split3(W[r * K + k0 + 8 * hf + j], a, b, c); wh[j] = a; wm[j] = b; wl[j] = c;
        }
#define MF(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC, 0, 0, 0)
        MF(accB, xl, wh); MF(accB, xh, wl); MF(accB, xm, wm); MF(accB, xm, wh); MF(accB, xh, wm); MF(accB, xh, wh);
        MF(accS, xl, wh); MF(accS, xh, wl); MF(accS, xm, wm); MF(accS, xm, wh); MF(accS, xh, wm);
        MF(accH, xh, wh);
#undef MF
    }
    // D[i][j]: A operand rows = i (X), B operand columns = j (W): lane holds column j = r, rows i = 8 (q >> 2) + 4 hf + (q & 3)
    for (int q = 0; q < 16; ++q) {
        const int i = 8 * (q >> 2) + 4 * hf + (q & 3);
        out[(0 * 32 + i) * 32 + r] = accA[q];
        out[(1 * 32 + i) * 32 + r] = accB[q];
        out[(2 * 32 + i) * 32 + r] = accH[q] + accS[q];
    }
}

static double frand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static float gauss() { return (float)(sqrt(-2.0 * log(frand())) * cos(6.283185307179586 * frand())); }

int main() {
    srand(12345);
    printf("max|err| / max|ref| (rms err / max|ref|) against the fp64 contraction; tolerance of the fp32 parity tests: 2e-6 sqrt(K / 512)\n");
    const int Ks[4] = {512, 2048, 4096, 8192};
    for (int kind = 0; kind < 2; ++kind)
        for (int t = 0; t < 4; ++t) {
            const int K = Ks[t];
            std::vector<float> X(32 * K), W(32 * K), out(3 * 32 * 32);
            for (int i = 0; i < 32 * K; ++i) {
                X[i] = gauss(); W[i] = gauss() * 0.05f;
                if (kind == 1) { X[i] *= expf(2.0f * gauss()); W[i] *= expf(2.0f * gauss()); }   // mixed magnitudes
            }
            float *dX, *dW, *dO;
            hipMalloc(&dX, X.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dO, out.size() * 4);
            hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dX, dW, K, dO);
            hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
            hipFree(dX); hipFree(dW); hipFree(dO);
            double mx = 0, e[3] = {0, 0, 0}, s[3] = {0, 0, 0};
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ref = 0;
                    for (int k = 0; k < K; ++k) ref += (double)X[i * K + k] * (double)W[j * K + k];
                    mx = fmax(mx, fabs(ref));
                    for (int v = 0; v < 3; ++v) {
                        const double d = fabs((double)out[(v * 32 + i) * 32 + j] - ref);
                        e[v] = fmax(e[v], d); s[v] += d * d;
                    }
                }
            printf("%s K = %4d  tol %.2e | A fp32 MFMA chain %.2e (%.2e) | B six bf16 products, one accumulator %.2e (%.2e) | C two accumulators %.2e (%.2e)\n",
                   kind ? "mixed " : "normal", K, 2e-6 * sqrt(K / 512.0), e[0] / mx, sqrt(s[0] / 1024) / mx, e[1] / mx,
                   sqrt(s[1] / 1024) / mx, e[2] / mx, sqrt(s[2] / 1024) / mx);
        }
    return 0;
}
