/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (quantized-autoregression-image-generator_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Plain-C restatement of the BMU search of the reference:
 *   Codebook.get_patches_bmu   models/Codebook.py:77-99
 *     patchify                 models/layers.py:8-34   (channel-major, row, col)
 *     torch.cdist(x, W)        models/Codebook.py:86-88 (p=2, default compute mode:
 *                              matmul form |x|^2+|w|^2-2x.w, clamp_min 0, sqrt when
 *                              either side has > 25 rows; direct form otherwise)
 *     torch.argmin(dim=-1)     models/Codebook.py:91-94 (first minimal index, int64)
 * torch's sgemm summation order is not part of any contract and cannot be restated;
 * this oracle fixes a sequential fp32 fma order (the one the HIP kernel's
 * v_mfma_f32_32x32x2_f32 chain produces):  d2 = (chain(-2 w.x) + |w|^2) + |x|^2,
 * each norm its own ascending fma chain; it is pinned to the reference by
 * tests/golden/bmu_*.npz (indices produced by the reference's own Codebook class in
 * the build container, oracle/make_goldens.py).
 *
 * Build: make -C oracle   ->  oracle/_build/libqarig_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <omp.h>

void oracle_set_threads(int n) { omp_set_num_threads(n); }
int oracle_get_threads(void) { return omp_get_max_threads(); }

/* element (c,i,j) of patch `row` of x (N,C,H,W), patch grid gh x gw */
static inline int64_t patch_base(int row, int C, int H, int W, int pH, int pW, int gh, int gw) {
    int per = gh * gw;
    int n = row / per;
    int rem = row - n * per;
    int ph = rem / gw, pw = rem - ph * gw;
    return ((int64_t)n * C * H + (int64_t)ph * pH) * W + (int64_t)pw * pW;
}

/* Gathers patch `row` into buf[D] in patchify order. */
static void gather_patch(const float* x, int row, int C, int H, int W, int pH, int pW, int gh,
                         int gw, float* buf) {
    int64_t base = patch_base(row, C, H, W, pH, pW, gh, gw);
    int e = 0;
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < pH; ++i)
            for (int j = 0; j < pW; ++j) buf[e++] = x[base + ((int64_t)c * H + i) * W + j];
}

/* layers.py:8-34 -- out (N*Seq, D) */
void oracle_patchify(const float* x, int N, int C, int H, int W, int pH, int pW, float* out) {
    int gh = H / pH, gw = W / pW, D = C * pH * pW;
    int R = N * gh * gw;
    for (int r = 0; r < R; ++r) gather_patch(x, r, C, H, W, pH, pW, gh, gw, out + (int64_t)r * D);
}

/* Codebook.py:77-99.  Returns 0. out: int64[N*gh*gw]. */
int oracle_bmu(const float* x, int N, int C, int H, int W, int pH, int pW, const float* w, int K,
               int64_t* out) {
    int gh = H / pH, gw = W / pW, D = C * pH * pW;
    int R = N * gh * gw;
    float* bufs = (float*)malloc(sizeof(float) * (size_t)D * (size_t)omp_get_max_threads());
    float* w2 = (float*)malloc(sizeof(float) * (size_t)K);
    const int mm_form = (R > 25) || (K > 25);
    for (int k = 0; k < K; ++k) {
        float acc = 0.0f;
        for (int e = 0; e < D; ++e) acc = fmaf(w[(int64_t)k * D + e], w[(int64_t)k * D + e], acc);
        w2[k] = acc;
    }
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        float* buf = bufs + (size_t)omp_get_thread_num() * D;
        gather_patch(x, r, C, H, W, pH, pW, gh, gw, buf);
        float x2 = 0.0f;
        for (int e = 0; e < D; ++e) x2 = fmaf(buf[e], buf[e], x2);
        float best = INFINITY;
        int64_t idx = 0;
        for (int k = 0; k < K; ++k) {
            const float* wk = w + (int64_t)k * D;
            float d;
            if (mm_form) {
                float acc = 0.0f;
                for (int e = 0; e < D; ++e) acc = fmaf(-2.0f * wk[e], buf[e], acc);
                float d2 = (acc + w2[k]) + x2;
                d = sqrtf(d2 > 0.0f ? d2 : 0.0f);
            } else {
                float acc = 0.0f;
                for (int e = 0; e < D; ++e) {
                    float t = buf[e] - wk[e];
                    acc = fmaf(t, t, acc);
                }
                d = sqrtf(acc);
            }
            if (d < best) { best = d; idx = k; }
        }
        out[r] = idx;
    }
    free(bufs);
    free(w2);
    return 0;
}

/* Exact-arithmetic companion used to classify rows where fp32 implementations may
 * legitimately disagree: top-2 gap of the true (double) distances, per row.
 * gap[r] = (second smallest d) - (smallest d), idx64[r] = argmin in double. */
int oracle_bmu_f64(const float* x, int N, int C, int H, int W, int pH, int pW, const float* w,
                   int K, int64_t* idx64, double* gap) {
    int gh = H / pH, gw = W / pW, D = C * pH * pW;
    int R = N * gh * gw;
    float* buf = (float*)malloc(sizeof(float) * (size_t)D);
    for (int r = 0; r < R; ++r) {
        gather_patch(x, r, C, H, W, pH, pW, gh, gw, buf);
        double b1 = INFINITY, b2 = INFINITY;
        int64_t i1 = 0;
        for (int k = 0; k < K; ++k) {
            const float* wk = w + (int64_t)k * D;
            double acc = 0.0;
            for (int e = 0; e < D; ++e) {
                double t = (double)buf[e] - (double)wk[e];
                acc += t * t;
            }
            double d = sqrt(acc);
            if (d < b1) { b2 = b1; b1 = d; i1 = k; }
            else if (d < b2) { b2 = d; }
        }
        idx64[r] = i1;
        gap[r] = b2 - b1;
    }
    free(buf);
    return 0;
}
