// How v_mfma_f32_32x32x16_bf16 rounds: the BMU coarse pass's certificate (csrc/bmu.hip, BMU_COARSE_EPS)
// bounds the error of seven such instructions per tile assuming every fp32 addition inside the matrix
// core rounds to nearest.  This probe feeds the instruction sums whose exact value lies a known fraction of
// an ulp above a representable fp32 and prints what comes back:
//   case A  C = 1, one product = f * 2^-23          (the final add into the accumulator)
//   case B  C = 0, products 1 and f * 2^-23          (an add between two products of one instruction)
//   case C  C = 0, sixteen products 1/16 + e_k        (a 16-term sum: how many roundings it sees)
// for f = 0.25, 0.5, 0.75, 1.25, 1.5 ulp.  Round-to-nearest-even gives 1, 1, 1 + ulp, 1 + ulp, 1 + 2 ulp (ties
// to even); truncation gives 1, 1, 1, 1 + ulp, 1 + ulp.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_rounding_probe.hip -o tools/mfma_rounding_probe && tools/mfma_rounding_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A (32 x 16) row i = lane % 32, k = 8 * (lane / 32) + j; B (16 x 32) column i = lane % 32, same k.
// Every row of A and every column of B hold the same 16 values: a[k] and b[k]; D[i][j] = c + sum_k a[k] b[k].
__global__ void probe(const float* a, const float* b, float c, float* out) {
    const int lane = threadIdx.x;
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) {
        av[j] = (__bf16)a[8 * (lane / 32) + j];
        bv[j] = (__bf16)b[8 * (lane / 32) + j];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

static float run(const float* a, const float* b, float c) {
    float *da, *db, *dout, r = 0.0f;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    hipMemcpy(da, a, 64, hipMemcpyHostToDevice);
    hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return r;
}

int main() {
    const float ulp = 1.1920928955078125e-7f;   // 2^-23
    const float fr[5] = {0.25f, 0.5f, 0.75f, 1.25f, 1.5f};
    printf("fraction of an ulp above 1.0 | A: C=1 + product | B: product 1 + product | C: 16-term sum (exact: 1 + f ulp)\n");
    for (int t = 0; t < 5; ++t) {
        float a[16] = {0}, b[16] = {0};
        // f * 2^-23 as a product of two bf16 numbers: (f as bf16: 0.25 .. 1.5 are exact) * 2^-23
        a[0] = fr[t]; b[0] = ulp;
        const float ra = run(a, b, 1.0f);
        a[1] = 1.0f; b[1] = 1.0f;
        const float rb = run(a, b, 0.0f);
        // sixteen products: fifteen of 1/16 and one of 1/16 + f ulp  (1/16 + f * 2^-23 is not a bf16: use two
        // products in slot 0 and 1: 1/16 * 1 and f * 2^-23 ... slot 1 then carries f ulp alone, the other 14 = 1/16,
        // and slot 15 = 2/16, so that the exact sum is 1 + f ulp with partial sums off the power-of-two grid)
        float a2[16], b2[16];
        for (int k = 0; k < 16; ++k) { a2[k] = 0.0625f; b2[k] = 1.0f; }
        a2[1] = fr[t]; b2[1] = ulp;
        a2[15] = 0.125f;
        const float rc = run(a2, b2, 0.0f);
        printf("f = %.2f | %.1f ulp | %.1f ulp | %.1f ulp\n", fr[t], (ra - 1.0f) / ulp, (rb - 1.0f) / ulp, (rc - 1.0f) / ulp);
    }
    // many small addends: 16 products of (1/16)(1 + 2^-7) -- exact sum 1 + 2^-7, every partial sum needs <= 12 bits:
    // exact in any mode; and 16 products of 1/16 + 2^-26 each via two-term splits is not representable in bf16, so
    // instead: C = 1 and sixteen products of 2^-27 (sum = 2^-23 = 1 ulp): sequential rounding to nearest loses every
    // one of them (each is 1/16 ulp), a wide internal accumulator keeps the whole ulp
    float a3[16], b3[16];
    for (int k = 0; k < 16; ++k) { a3[k] = 1.0f; b3[k] = 7.450580596923828e-09f; }   // 2^-27
    printf("C = 1 + 16 x 2^-27 (exact 1 + 1 ulp): %.2f ulp above 1  (0: every add rounded on its own; 1: summed wide, rounded once)\n",
           (run(a3, b3, 1.0f) - 1.0f) / ulp);
    return 0;
}
