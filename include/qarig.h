/* qarig.h -- C ABI of libqarig_hip.so: the MI355X (gfx950) hot path of the
 * quantized-autoregressive image pipeline (conv autoencoder -> SOM/BMU codebook ->
 * cascaded Transformer).
 *
 * The reference (Vinmwaura/Quantized-Autoregression-Image-Generator) has no FFI: its
 * boundary for this path is the Python class surface of models/ (SURVEY.md 8b).
 * Each entry point below replaces the body of one reference method / ATen call
 * site, cited as `reference file:line`.  The Python mirror of that class surface
 * (quantized-autoregression-image-generator_amd/models/) binds these with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HIP), fp32 unless stated, contiguous in the
 *    stated layout; token / index tensors are int64;
 *  - the library never allocates, frees or synchronises: outputs and workspaces are
 *    caller-owned, `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - return 0 on success, <0 on error (QARIG_ERR_*); qarig_last_error() gives the
 *    message of the calling thread's last failure.  Nothing throws across the ABI.
 */
#ifndef QARIG_H
#define QARIG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QARIG_OK 0
#define QARIG_ERR_ARG -1
#define QARIG_ERR_LAUNCH -2
#define QARIG_ERR_WORKSPACE -3

/* activation ids -- reference models/layers.py:74-80 get_activation */
#define QARIG_ACT_NONE 0
#define QARIG_ACT_SILU 1
#define QARIG_ACT_TANH 2
#define QARIG_ACT_SIGMOID 3

int qarig_version(void);
const char* qarig_target_arch(void);
int qarig_last_error(char* buf, size_t n);

/* ---- Codebook ---------------------------------------------------------------- */

/* Codebook.get_patches_bmu -- models/Codebook.py:77-99 (patchify layers.py:8-34,
 * torch.cdist + torch.argmin).  x: (N,C,H,W); codebook: (K,D), D = C*pH*pW;
 * out_idx: int64 (N * (H/pH) * (W/pW)), patch-grid row-major. */
size_t qarig_bmu_workspace_bytes(int64_t rows, int K);
int qarig_bmu_fwd(const float* x, int N, int C, int H, int W, int pH, int pW,
                  const float* codebook, int K, int D, int64_t* out_idx, void* workspace,
                  size_t ws_bytes, void* stream);

/* ---- Linear algebra core ----------------------------------------------------- */

/* C[M,N] = epilogue(sum_k A(m,k) B(n,k)).  a_kcontig: A stored [M][K] (1) or
 * [K][M] (0); same for B over N.  Epilogue, in order: + bias[n]; + residual[m][n];
 * store to preact (if given); act(); * act'(gradz[m][n]) with activation id gact (if
 * gradz given); store to C.  splitk > 1 (plain epilogue only) splits the reduction
 * over grid.z through fp32 slabs in `workspace`, summed in fixed order.
 * Replaces nn.Linear (+activation) inside LinearLayer / ResidualLinearLayer
 * (models/layers.py:234-304) forward, and the three autograd contractions. */
size_t qarig_gemm_workspace_bytes(int M, int N, int splitk);
int qarig_gemm_f32(const float* A, int64_t lda, int a_kcontig, const float* B, int64_t ldb,
                   int b_kcontig, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                   const float* residual, int64_t ldr, float* preact, int64_t ldp, int act,
                   const float* gradz, int64_t ldz, int gact, int splitk, void* workspace,
                   size_t ws_bytes, void* stream);

/* out[N] = column sums of X[M][N] in a fixed order (bias / LayerNorm-affine grads). */
size_t qarig_colsum_workspace_bytes(int M, int N);
int qarig_colsum_f32(const float* X, int64_t ldx, int M, int N, float* out, void* workspace,
                     size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QARIG_H */
