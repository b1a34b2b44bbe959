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
/* Kernel-selection options: each only chooses between kernels that must give the same results
 * (tests/test_gpu_switches.py runs the parity tests under every one).  Names: gemm_dma (1), gemm_pair
 * (-1 auto), bmu_cs (0 auto), bmu_groups (-1 auto), bmu_coarse (-1 auto), attn_qw / attn_bw (0 auto),
 * lp_big (-1 auto), lp_mfma16 (1), convt_pair (1), conv_ring (1), gemm_xcd_splits (1).  Returns the previous value, INT_MIN
 * for an unknown name.  No reference counterpart. */
int qarig_set_option(const char* name, int value);

/* ---- Codebook ---------------------------------------------------------------- */

/* Codebook.get_patches_bmu -- models/Codebook.py:77-99 (patchify layers.py:8-34,
 * torch.cdist + torch.argmin).  x: (N,C,H,W); codebook: (K,D), D = C*pH*pW;
 * out_idx: int64 (N * (H/pH) * (W/pW)), patch-grid row-major. */
size_t qarig_bmu_workspace_bytes(int64_t rows, int K);
int qarig_bmu_fwd(const float* x, int N, int C, int H, int W, int pH, int pW,
                  const float* codebook, int K, int D, int64_t* out_idx, void* workspace,
                  size_t ws_bytes, void* stream);

/* qarig_bmu_fwd with the codebook's prepared image (qarig_bmu_prepare below; NULL = none): where the
 * dispatch takes the coarse-pass kernel, its workgroups copy the image into LDS (global -> LDS DMA) instead
 * of each converting the codebook.  For a codebook that does not change between calls (tokenising a
 * dataset, the Transformer training loop); same indices. */
int qarig_bmu_fwd_prepared(const float* x, int N, int C, int H, int W, int pH, int pW,
                           const float* codebook, int K, int D, int64_t* out_idx, void* workspace,
                           size_t ws_bytes, const void* prepared, void* stream);

/* The same search through a coarse pass on the bf16 MFMA (every operand split into three bf16
 * pieces that add up to it exactly, six products on top of |w|^2 per 32 x 32 tile), a per-row certificate
 * (second-smallest - smallest > 3 eps, eps a proven bound of the coarse error) and the literal
 * fp32 re-scan of every row without one: bit-identical indices to qarig_bmu_fwd's exact kernels.
 * qarig_bmu_fwd takes this form by itself where it applies (D <= 16, D % 4 == 0, K % 32 == 0,
 * K <= 1024, >= 24576 rows); this entry forces it and counts the re-scanned rows into
 * uncertified[0] (device unsigned[8], caller-zeroed, may be NULL; [1..3] += clock64() cycles of the
 * staging / scan / finish phases of every block, [4] += blocks: tools/bmu_bench.py).  models/Codebook.py:77-99. */
int qarig_bmu_fwd_coarse(const float* x, int N, int C, int H, int W, int pH, int pW,
                         const float* codebook, int K, int D, int64_t* out_idx,
                         unsigned* uncertified, const void* prepared, void* stream);
/* The staged form of a codebook for `prepared` above (may be NULL: every workgroup then stages the
 * codebook itself): qarig_bmu_prepare_bytes(K, D) bytes (0 = the coarse form does not apply), written
 * by qarig_bmu_prepare.  A frozen codebook -- tokenising a dataset (generate_fmap_dataset.py /
 * train_quantized_transformer.py:412-421 call get_patches_bmu with fixed codebooks every step) -- is
 * prepared once; its image must be rebuilt after the codebook changes. */
size_t qarig_bmu_prepare_bytes(int K, int D);
int qarig_bmu_prepare(const float* codebook, int K, int D, void* image, void* stream);

/* ---- Linear algebra core ----------------------------------------------------- */

/* C[M,N] = epilogue(sum_k A(m,k) B(n,k)).  a_kcontig: A stored [M][K] (1) or
 * [K][M] (0); same for B over N.  Epilogue, in order: + bias[n]; + residual[m][n];
 * store to preact (if given); act(); * act'(gradz[m][n]) with activation id gact (if
 * gradz given); store to C.  splitk > 1 splits the reduction
 * over grid.z through fp32 slabs in `workspace`, summed in fixed order (the epilogue then
 * runs in the reduce pass; used for skinny-M decode GEMMs).  accumulate != 0
 * (plain epilogue only) adds the product to what C already holds (weight gradients are
 * accumulated straight into the flat .grad buffer).  a_rowsum (optional, [M]) receives
 * sum_k A(m,k) (added to it when accumulate): with A = dT^T this is the bias gradient,
 * computed from the A tiles the weight-gradient GEMM stages anyway.
 * Replaces nn.Linear (+activation) inside LinearLayer / ResidualLinearLayer
 * (models/layers.py:234-304) forward, and the three autograd contractions. */
size_t qarig_gemm_workspace_bytes(int M, int N, int splitk);
int qarig_gemm_f32(const float* A, int64_t lda, int a_kcontig, const float* B, int64_t ldb,
                   int b_kcontig, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                   const float* residual, int64_t ldr, float* preact, int64_t ldp, int act,
                   const float* gradz, int64_t ldz, int gact, int splitk, int accumulate,
                   float* a_rowsum, void* workspace, size_t ws_bytes, void* stream);

/* `groups` (1..16) products of ONE shape as one launch: C_g = epilogue(A_g B_g^T), every argument of
 * qarig_gemm_f32 with the pointers given as host arrays of `groups` device pointers (a NULL array =
 * that epilogue input absent for every group; entries of one array may repeat, e.g. a shared A).
 * Strides, layouts, activation ids, splitk and accumulate are common to the groups.
 * sum_groups != 0: C[0] (+)= sum_g A_g B_g^T (plain epilogue), summed over (g, split) in ascending
 * order.  Interior shapes only (qarig_gemm_grouped_supported: M, N multiples of 128, whole 16-deep
 * tiles per split, 16-B aligned operands).  The q / k / v two-layer MLPs of an attention layer
 * (models/layers.py:389-418) -- forward, both input gradients and both weight gradients -- run as
 * one such launch per product instead of three; the cross-attention k / v MLPs of every decoder
 * layer (models/layers.py:538-599, all reading the encoder output, models/Transformer.py:179-191)
 * as one launch per product for all layers. */
int qarig_gemm_grouped_supported(int M, int N, int K, int splitk);
size_t qarig_gemm_grouped_workspace_bytes(int groups, int M, int N, int splitk, int sum_groups);
int qarig_gemm_f32_grouped(int groups, const float* const* A, int64_t lda, int a_kcontig,
                           const float* const* B, int64_t ldb, int b_kcontig, float* const* C,
                           int64_t ldc, int M, int N, int K, const float* const* bias,
                           const float* const* residual, int64_t ldr, float* const* preact,
                           int64_t ldp, int act, const float* const* gradz, int64_t ldz, int gact,
                           int splitk, int accumulate, int sum_groups, float* const* a_rowsum,
                           void* workspace, size_t ws_bytes, void* stream);

/* Opt-in reduced precision (BASELINE config 5: "fp8/bf16 MFMA attn/FFN"; never the fp32 parity
 * mode).  The Linear contractions of models/layers.py:234-304, 330-340, 389-418 with bf16
 * OPERANDS IN HBM, products on v_mfma_f32_32x32x16_bf16, fp32 accumulation and the same fused
 * fp32 epilogue as qarig_gemm_f32.  layout 0 = NT: A (M,K), B (N,K), reduction-contiguous
 * (forward x W^T; input gradient dT W with the W^T shadow as B); layout 1 = TN: A (K,M), B (K,N),
 * reduction-major (weight gradient dT^T x from the row-major activations, transposed on the LDS
 * read by ds_read_b64_tr_b16); layout 2 = NN: A (M,K) reduction-contiguous, B (K,N)
 * reduction-major (input gradient dT W on the weight shadow exactly as stored: no W^T copy).
 * A, B: bf16 (16-bit) elements, lda / ldb in elements.
 * C: fp32 output (may be NULL when Cb is given); Cb / Pb: optional bf16 copies of the output /
 * of the saved pre-activation for a consumer GEMM.  Shapes: qarig_gemm_lp_supported. */
int qarig_gemm_lp_supported(int M, int N, int K, int splitk);
size_t qarig_gemm_lp_workspace_bytes(int M, int N, int splitk);
int qarig_gemm_lp(const void* A, int64_t lda, const void* B, int64_t ldb, int layout, float* C,
                  int64_t ldc, int M, int N, int K, const float* bias, const float* residual,
                  int64_t ldr, float* preact, int64_t ldp, int act, const void* gradz, int64_t ldz,
                  int gradz_is_bf16, int gact, int splitk, int accumulate, void* Cb, int64_t ldcb,
                  void* Pb, int64_t ldpb, void* workspace, size_t ws_bytes, void* stream);

/* fp8 (OCP e4m3) forward products of the same Linear layers (BASELINE config 5 names the fp8
 * MFMA; models/layers.py:234-254, 389-418): A (M,K) and B (N,K) are e4m3 BYTES quantised per
 * tensor by qarig_cast_fp8 (scale 448 / max|x|; lda / ldb in bytes), products on
 * v_mfma_f32_32x32x64_f8f6f4, fp32 accumulation, the accumulator multiplied by the two device
 * scalars inv_a[0] * inv_b[0] before the usual epilogue.  qarig_cast_fp8: dst = n bytes,
 * inv_scale[0] receives max|x| / 448, scratch = 4 bytes of device memory (receives max|x|),
 * bf16_dst = optional bf16 copy of src from the same pass (the backward products read it). */
int qarig_gemm_f8_supported(int M, int N, int K);
int qarig_cast_fp8(const float* src, int64_t n, void* dst, float* inv_scale, void* scratch,
                   void* bf16_dst, void* stream);
int qarig_gemm_f8(const void* A, int64_t lda, const void* B, int64_t ldb, const float* inv_a,
                  const float* inv_b, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                  const float* residual, int64_t ldr, float* preact, int64_t ldp, int act, void* Cb,
                  int64_t ldcb, void* Pb, int64_t ldpb, void* stream);

/* Operand conversion of the reduced-precision mode (no reference counterpart): fp32 -> bf16,
 * round to nearest even; n contiguous elements, or the transpose (C, R) of a (R, C) matrix
 * with row stride ld (the W^T shadow of an nn.Linear weight). */
int qarig_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
int qarig_cast_transpose_bf16(const float* src, int64_t ld, int R, int C, void* dst, void* stream);
/* colsum[n] (+)= sum_m src[m][n] in a fixed order -- the bias gradient of nn.Linear
 * (models/layers.py:243-250) -- and, when dst is given, dst = bf16(src) in the same pass over
 * src (fp32, or bf16 with src_is_bf16 = 1 and dst NULL). */
size_t qarig_cast_colsum_workspace_bytes(int M, int N);
int qarig_cast_colsum(const void* src, int64_t ld, int src_is_bf16, int M, int N, void* dst,
                      float* colsum, int accumulate, void* workspace, size_t ws_bytes, void* stream);

/* 1 when qarig_gemm_f32 runs an (M, N, K) product on 64 x 64 tiles (whole 64-tiles, 16-deep k-tiles, fewer
 * than 192 tiles of 128 x 128: the 512-wide Linear products of a 2,048-row shard, models/layers.py:291-304);
 * a caller that chooses the reduction split counts those tiles. */
int qarig_gemm_tile64(int M, int N, int K);

/* 1 when, under option gemm_x3 (qarig_set_option("gemm_x3", 1); off by default), qarig_gemm_f32 runs an
 * (M, N, K) product with `splitk` reduction splits on the bf16 matrix pipe: both fp32 operands split on
 * the fly into three bf16 pieces that add up to them exactly, six products per block accumulated in fp32
 * (csrc/gemm_x3.hip: whole 128 x 128 tiles and 32-deep k-tiles per split).  Same operands, epilogue and
 * tolerances as the fp32-MFMA kernels (models/layers.py:234-254, 330-340); non-finite operands give NaN. */
int qarig_gemm_x3_ok(int M, int N, int K, int splitk);

/* `groups` independent skinny products in one launch, C_g = act(A_g W_g^T + bias_g) with
 * X_g = X + g * x_gs (a_gs == 0 shares the activations).  Decode steps use it for the q/k/v
 * MLPs (models/layers.py:389-418) and for every projection of the conditioning vector
 * (ScaleLayer/ShiftLayer, models/layers.py:100-153, 258-304) of all layers at once.
 * Requires M <= 512, K % 256 == 0, reduction-contiguous 16-B aligned operands. */
int qarig_gemm_grouped_skinny_f32(const float* A, int64_t lda, int64_t a_gs, const float* W,
                                  int64_t ldw, int64_t w_gs, float* C, int64_t ldc, int64_t c_gs,
                                  const float* bias, int64_t bias_gs, int groups, int M, int N,
                                  int K, int act, void* stream);

/* The decode step's fused form of the launch above: C_g = act(LN(X) W_g^T + bias_g) [* mul].
 * The activations X (M, K) are shared by the groups and LayerNorm'ed over K on the way in
 * (biased variance, eps) -- nn.LayerNorm's affine form (gamma, beta: K) or the AdaLN form
 * scale(cond) * LN(x) + shift(cond) (scale, shift: (M, K) rows at ldmod; reference
 * models/layers.py:130-153); all four null = no normalisation.  mul (M, N) at ldmul, optional, is
 * an elementwise factor on the output, the same for every group (ResidualLinearLayer's
 * x * scale(cond), models/layers.py:258-304).  Replaces the LayerNorm launch in front of a block's
 * first Linear and the gate multiply behind its last (generate_images.py:283-290 evaluates these
 * per generated token).  Same shape rules as qarig_gemm_grouped_skinny_f32. */
int qarig_gemm_skinny_ln_f32(const float* X, int64_t ldx, float eps, const float* gamma,
                             const float* beta, const float* scale, const float* shift,
                             int64_t ldmod, const float* W, int64_t ldw, int64_t w_gs, float* C,
                             int64_t ldc, int64_t c_gs, const float* bias, int64_t bias_gs,
                             const float* mul, int64_t ldmul, int groups, int M, int N, int K,
                             int act, void* stream);

/* The Linear of a single-token decode step (generate_images.py:283-286 evaluates the decoder per
 * sampled token; with a key/value cache only the new row of each sequence is computed):
 *   C_g = act(LN(X_g) W_g^T + bias_g + residual) * mul        M <= 16 rows, `groups` products.
 * A weight-streaming kernel whose loads are all issued before its first wait (the step is a chain
 * of ~80 dependent launches, each bound by its memory round trips, not by bytes).  LN over K:
 * nn.LayerNorm's affine form (gamma, beta: K) or AdaLN's scale(cond) * LN(x) + shift(cond)
 * (scale, shift rows at ldmod, models/layers.py:130-153; ldmod == 0: ONE row for every activation
 * row -- the rows of a decode step share their window position); residual (M, N): the skip input
 * of ResidualLinearLayer (models/layers.py:291-304); mul (M, N) at ldmul (0: one row): its
 * x * scale(cond) gate applied by the producer.  X_g = X + g * x_gs (0 shares X).
 * qarig_decode_linear_supported: M <= 16 and K in {256, 512, 1024} (2048, 4096 when ln == 0). */
int qarig_decode_linear_supported(int M, int N, int K, int ln);
int qarig_decode_linear_f32(const float* X, int64_t ldx, int64_t x_gs, float eps,
                            const float* gamma, const float* beta, const float* scale,
                            const float* shift, int64_t ldmod, const float* W, int64_t ldw,
                            int64_t w_gs, const float* bias, int64_t bias_gs, const float* residual,
                            int64_t ldr, const float* mul, int64_t ldmul, float* C, int64_t ldc,
                            int64_t c_gs, int groups, int M, int N, int K, int act, void* stream);

/* ---- Device-resident sampling loop (generate_images.py:256-345) ------------------------------
 * The reference samples a token, appends it on the host and re-runs the decoder.  Here the loop's
 * state stays on the device and the host only enqueues launches, never reading a token back: `ctl` is an int32 array
 * of QARIG_DECODE_CTL_WORDS words -- [0] window index of the token the next step evaluates,
 * [1] window index at which the current chunk of beam_width tokens starts, [2] draws made so far
 * (row of the uniform / forced / log buffers), [3] candidate chunks evaluated at this position,
 * [4] decoder steps since the candidate began (the chunk slot the next draw fills). */
#define QARIG_DECODE_CTL_WORDS 8

/* First launch of a step: x[b] = table[ids[b]] + pe[len] (models/Transformer.py:154-167; pe may be
 * NULL), len = ctl[0] (ctl NULL: the `len` argument), and the copy of row `len` of proj_table
 * (max_len rows of proj_row_floats floats: every ScaleLayer / ShiftLayer projection of `cond` --
 * models/layers.py:100-153, 258-304 -- evaluated once per window position of the stage) into
 * proj_row.  With ctl it also counts the step: ctl[4] += 1.  An id outside [0, V) sets *bad_flag and
 * yields a zero row. */
int qarig_decode_embed(const int64_t* ids, int B, int D, int V, const float* table, const float* pe,
                       int* ctl, int len, int max_len, const float* proj_table,
                       int64_t proj_row_floats, float* x, float* proj_row, int* bad_flag, void* stream);

/* qarig_attention_decode with the cache layout given by strides -- row j of head h of sequence n at
 * n * batch_stride + h * head_stride + j * row_stride: row-major (head_stride = d, row_stride = H * d) or
 * head-major (row_stride = d, head_stride >= max_len * d: a head's keys contiguous, what the kernel's
 * lane-per-key loads coalesce on) -- and the o_mul factor at row stride ldmul (0: one row for every
 * sequence). */
int qarig_decode_attention(const float* q, const float* k_new, const float* v_new, float* kcache,
                           float* vcache, int B, int H, int d, int len, const int* len_dev,
                           int max_len, int64_t batch_stride, int64_t head_stride, int64_t row_stride,
                           float sqrt_d, const float* o_mul, int64_t ldmul, float* o, void* stream);

/* One sampling draw per row as generate_images.py:289-304 makes it (train_quantized_transformer.py:
 * 626-636 with generate_mode == 0): probs = softmax(logits / temperature), generate mode zeroes
 * probs[end_token], a token is drawn in proportion to probs by inverse CDF from uniforms[draw][b]
 * (draw = ctl[2] + slot; slot == -1: the slot the device counts, ctl[4]; the reference's
 * torch.multinomial consumes its generator differently: same distribution, not the same stream), comb[b] *= probs[token], train mode maps <end> to 0,
 * token + shift goes to ids[b] and chunk[b][slot]; inc_len != 0 advances ctl[0].  forced (optional):
 * entries >= 0 are taken instead of drawing; probs_log (optional, (max_draws, B, V)): the rows
 * sampled from.  beams > 0: the B rows are `beams` independent candidate chunks per image evaluated as
 * one batch while every draw keeps the number the reference's candidate-after-candidate loop gives it
 * (generate_images.py:262-304): row image * beams + c reads column `image` of draw row
 * ctl[2] + c * beam_width + slot; uniforms / forced / probs_log then have B / beams columns. */
int qarig_decode_sample(const float* logits, int64_t ldl, int B, int V, float temperature,
                        int end_token, int generate_mode, int64_t shift, const float* uniforms,
                        const int64_t* forced, int* ctl, int slot, int beam_width, int max_draws,
                        int inc_len, int beams, int64_t* ids, int64_t* chunk, float* comb,
                        float* probs_log, void* stream);

/* After a candidate chunk: per image the beam with the largest product (first on ties) replaces the
 * kept chunk unless the kept product is >= (generate_images.py:325-337); take[n] = 1 + that beam or
 * 0; comb is reset to 1, ctl: candidate + 1, draws + `draws` (the draw rows the candidate set consumed),
 * len back to the chunk start. */
int qarig_decode_decide(int* ctl, int N, int NB, int beam_width, int draws, float* comb,
                        const int64_t* chunk, float* best_p, int64_t* best_chunk, int* take, void* stream);

/* Cache rows [ctl[1], ctl[1] + R) of the head-major kv (layers2 = layers * 2, N * NB, H, max_len, d):
 * restore == 0 copies the winning beam's rows of the images with take[n] > 0 into staged
 * (layers2, N, H, R, d); restore == 1 writes the staged rows into every beam of every image. */
int qarig_decode_rows(const int* ctl, float* kv, float* staged, const int* take, int layers2, int N,
                      int NB, int H, int R, int d, int max_len, int restore, void* stream);

/* tokens[n][ctl[1] + j] = best_chunk[n][j]; ids of every beam = the chunk's last token;
 * ctl[0] = ctl[1] + beam_width - 1 (the step that follows produces the next chunk's first logits). */
int qarig_decode_commit(int* ctl, int N, int NB, int beam_width, const int64_t* best_chunk,
                        int64_t* tokens, int64_t ldt, int64_t* ids, void* stream);

/* ctl[1] += beam_width; ctl[0] = ctl[1]; ctl[3] = 0. */
int qarig_decode_advance(int* ctl, int beam_width, void* stream);

/* out[N] = column sums of X[M][N] in a fixed order (bias / LayerNorm-affine grads). */
size_t qarig_colsum_workspace_bytes(int M, int N);
int qarig_colsum_f32(const float* X, int64_t ldx, int M, int N, float* out, int accumulate,
                     void* workspace, size_t ws_bytes, void* stream);

/* ---- Codebook (continued) ------------------------------------------------------ */

/* patchify / unpatchify -- models/layers.py:8-34 / :37-71.  image (N,C,H,W) <->
 * patches (N*Seq, C*pH*pW).  H,W must be multiples of the patch here. */
int qarig_patchify_fwd(const float* image, int N, int C, int H, int W, int pH, int pW,
                       float* patches, void* stream);
int qarig_unpatchify_fwd(const float* patches, int N, int C, int H, int W, int pH, int pW,
                         float* image, void* stream);

/* Codebook.get_quantized_image -- models/Codebook.py:138-154: image = unpatchify(
 * codebook[ids]).  ids int64 (N*Seq); *bad_flag (device int, caller-zeroed) is set on
 * an id outside [0,K). */
int qarig_codebook_gather_image(const int64_t* ids, int N, int C, int H, int W, int pH, int pW,
                                const float* codebook, int K, float* image, int* bad_flag,
                                void* stream);

/* nn.Embedding row gather out[r] = table[ids[r]] -- models/Codebook.py:132,144. */
int qarig_gather_rows(const int64_t* ids, int64_t R, int D, int K, const float* table, float* out,
                      int* bad_flag, void* stream);

/* Gaussian index-neighbourhood weights g[r][j] = exp(-(j-bmu[r])^2 / two_var) --
 * models/Codebook.py:112-126 (the (R,K)@(K,D) product then goes through qarig_gemm_f32). */
int qarig_som_weights_fwd(const int64_t* bmu, int64_t R, int K, float two_var, float* g,
                          void* stream);

/* The same neighbourhood without the (R,K) matrix: out[j] = sum_{|j-b| <= reach} exp(-(j-b)^2 / two_var) in[b]
 * over a (K,D) table.  models/Codebook.py:112-130's product equals gather_rows(bmu, band(codebook)) and its
 * codebook gradient band(embedding_bwd(bmu, dq)); `reach` bounds the dropped weights (host: < 2^-40). */
int qarig_som_band(const float* in, int K, int D, float two_var, int reach, float* out, void* stream);

/* counts (int64 [K]) += histogram of ids -- the per-unit BMU usage count of
 * prune_codebook.py:129-142. */
int qarig_index_histogram(const int64_t* ids, int64_t n, int K, int64_t* counts, int* bad_flag,
                          void* stream);

/* ---- Transformer pieces ------------------------------------------------------- */

/* get_positional_embeddings -- models/layers.py:83-96.  pos fp32 (R,), freq (D/2,)
 * (host-computed exactly as the reference does), out (R,D) = [sin | cos]. */
int qarig_posemb_fwd(const float* pos, int R, int D, const float* freq, float* out, void* stream);

/* Token assembly of the training hot loop (train_quantized_transformer.py:423-484) in one launch:
 * from the LR / HR BMU indices to the windowed decoder input, target and absolute positions
 * (base: [lr | hr + k_lr] / enc-dec: [<start> | hr]; target [hr | <end>]; window of W tokens
 * starting at offs[n]; pos = offs[n] + w).  Replaces cat + unfold + gather + arange on the host.
 * *bad_flag (device int, caller-zeroed, may be NULL) is set on a window start outside
 * [0, S_in - W]; the window is then clamped (torch.gather would raise). */
int qarig_assemble_tokens(const int64_t* lr, int S_lr, const int64_t* hr, int S_hr, int N, int base,
                          int k_lr, int k_hr, const int64_t* offs, int W, int64_t* hr_in,
                          int64_t* hr_tg, int64_t* pos, int* bad_flag, void* stream);

/* nn.Embedding + additive position table -- models/Transformer.py:127-139,154-167.
 * ids int64 (M = N*S); pe (S,D) or NULL; out (M,D). */
int qarig_embedding_fwd(const int64_t* ids, int M, int S, int D, int V, const float* table,
                        const float* pe, float* out, int* bad_flag, void* stream);
/* dtable[v] = sum_{m: ids[m]==v} dy[m], m ascending (autograd of nn.Embedding). */
int qarig_embedding_bwd(const int64_t* ids, int M, int D, int V, const float* dy, float* dtable,
                        void* stream);

/* nn.LayerNorm(D) (gamma,beta) / AdaLNZero modulation (scale,shift per token) --
 * models/layers.py:130-153,327,499,559.  Exactly one of the pairs, or neither. */
int qarig_layernorm_fwd(const float* x, int M, int D, float eps, const float* gamma,
                        const float* beta, const float* scale, const float* shift,
                        const int* mod_idx, float* y, float* mean, float* rstd, void* stream);
/* dx_add (optional, (M,D)): added to dx -- the gradient that reaches the same x through the block's
 * skip connection (models/layers.py:362-366, 530-534, 595-599), so that autograd's accumulation
 * of the two contributions is not a separate elementwise launch. */
int qarig_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                        const float* gamma, const float* scale, const int* mod_idx, int M, int D,
                        float* dx, float* dy_xhat, const float* dx_add, void* stream);

/* Position-table form of the conditioning path.  `cond` (models/Transformer.py:154-167) is a
 * function of the token's integer position alone, so pos_cond_layer and every
 * ScaleLayer/ShiftLayer projection (models/layers.py:100-153, 258-304) are evaluated once per
 * distinct position into (P,D) tables and the per-token consumers index them:
 *   - qarig_layernorm_fwd/bwd with mod_idx (int32 (M,)): scale/shift row = table[mod_idx[row]];
 *     mod_idx == NULL keeps the per-token (M,D) scale/shift form;
 *   - qarig_mul_rows_fwd/bwd: y[r] = a[r] * tab[idx[r]] (ResidualLinearLayer's x * scale(cond));
 *   - qarig_rowmap_build + qarig_segment_sum: gradient of a table row = sum of its tokens'
 *     per-token gradients in ascending token order (deterministic, no atomics).
 * counts: P ints scratch; offsets: P+1 ints; rows: M ints; *bad_flag is set when an index is
 * outside [0,P). */
int qarig_rowmap_build(const int* idx, int M, int P, int* counts, int* offsets, int* rows,
                       int* bad_flag, void* stream);
int qarig_segment_sum(const float* src, const int* offsets, const int* rows, int P, int D,
                      float* out, void* stream);
int qarig_mul_rows_fwd(const float* a, const float* tab, const int* idx, float* y, int M, int D,
                       void* stream);
int qarig_mul_rows_bwd(const float* dy, const float* a, const float* tab, const int* idx, float* da,
                       float* db_tok, int M, int D, void* stream);

/* AttentionLayer core -- models/layers.py:433-474.  q (N,Sq,H*d); k,v (N,Sk,H*d);
 * o (N,Sq,H*d); lse (N,H,Sq); sqrt_d = float(d ** 0.5). */
int qarig_attention_fwd(const float* q, const float* k, const float* v, int N, int Sq, int Sk,
                        int H, int d, int causal, float sqrt_d, float* o, float* lse,
                        void* stream);
int qarig_attention_bwd(const float* q, const float* k, const float* v, const float* o,
                        const float* dO, const float* lse, int N, int Sq, int Sk, int H, int d,
                        int causal, float sqrt_d, float* dq, float* dk, float* dv, float* delta,
                        void* stream);

/* The same attention (models/layers.py:433-474) with the QK^T / PV products and their backward
 * counterparts on the bf16 MFMA (operands rounded to bf16; fp32 accumulation, softmax, LSE and
 * tensors): BASELINE config 5's reduced-precision attention.  Opt-in; same arguments. */
int qarig_attention_lp_fwd(const float* q, const float* k, const float* v, int N, int Sq, int Sk, int H,
                           int d, int causal, float sqrt_d, float* o, float* lse, void* stream);
int qarig_attention_lp_bwd(const float* q, const float* k, const float* v, const float* o,
                           const float* dO, const float* lse, int N, int Sq, int Sk, int H, int d,
                           int causal, float sqrt_d, float* dq, float* dk, float* dv, float* delta,
                           void* stream);

/* Single-token decode step against a KV cache.  The reference has no cache: it re-runs the
 * whole window for every sampled token (generate_images.py:283-307,
 * train_quantized_transformer.py:600-640); this computes the same attention row
 * (models/layers.py:433-474 for the last query) from cached keys/values.  q,k_new,v_new,o
 * (B,H*d); cache row j of sequence n at n*batch_stride + j*H*d.  k_new/v_new non-NULL:
 * stored at row len and attended as the last key; NULL: read-only cache (cross-attention).
 * len_dev (device int, optional) overrides len for graph replay.  o_mul (B,H*d), optional:
 * multiplies the output (ResidualLinearLayer's x * scale(cond), models/layers.py:293-295). */
int qarig_attention_decode(const float* q, const float* k_new, const float* v_new, float* kcache,
                           float* vcache, int B, int H, int d, int len, const int* len_dev,
                           int max_len, int64_t batch_stride, float sqrt_d, const float* o_mul,
                           float* o, void* stream);

/* nn.CrossEntropyLoss() mean over rows + d/dlogits --
 * train_quantized_transformer.py:337,496-502.  row_ws: M floats. */
int qarig_cross_entropy_fwd(const float* logits, const int64_t* target, int M, int C, float* loss,
                            float* dlogits, float* row_ws, int* bad_flag, void* stream);

/* F.mse_loss (mean) + d/dpred -- train_autoencoder.py:215-217, train_codebook.py:233-235. */
size_t qarig_mse_workspace_bytes(void);
int qarig_mse_fwd(const float* pred, const float* target, int64_t n, float* loss, float* dpred,
                  float* part_ws, void* stream);

/* torch.optim.Adam step on a flat buffer -- train_quantized_transformer.py:317-320.
 * dev_step (optional device float[2] = {lr / (1 - beta1^t), sqrt(1 - beta2^t)}) overrides the two
 * per-step scalars (graph replay).  shadow_bf16 (optional, n x 2 bytes): the updated parameters rounded to
 * bf16 from the same pass (the reduced-precision GEMMs' weight operands). */
int qarig_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float beta1,
                    float beta2, float eps, float step_size, float bc2_sqrt, float grad_scale,
                    const float* dev_step, void* shadow_bf16, void* stream);

/* elementwise helpers (ResidualLinearLayer gate, models/layers.py:293-295) */
int qarig_mul_fwd(const float* a, const float* b, float* y, int64_t n, void* stream);
int qarig_mul_bwd(const float* dy, const float* a, const float* b, float* da, float* db, int64_t n,
                  void* stream);
int qarig_act_fwd(const float* x, float* y, int64_t n, int act, void* stream);
int qarig_act_bwd(const float* dy, const float* z, float* dz, int64_t n, int act, void* stream);
int qarig_scale_by(const float* x, const float* s, float* y, int64_t n, void* stream);

/* ---- Conv autoencoder ----------------------------------------------------------- */

/* nn.Conv2d(k<=4, stride, padding) + bias + activation, NCHW -- ConvLayer /
 * DownsampleConvLayer, models/layers.py:157-184, 211-230 (3x3, stride 1|2, pad 1 in the
 * reference).  x (N,Cin,H,W); w (Cout,Cin,k,k); y (N,Cout,Ho,Wo); preact (optional,
 * same shape as y) receives the pre-activation. */
int qarig_conv2d_fwd(const float* x, int N, int Cin, int H, int W, const float* w,
                     const float* bias, int Cout, int k, int stride, int pad, int act, float* y,
                     float* preact, void* stream);
/* The same with a scratch buffer for a tap-major copy of the weights: the 3x3 / stride 1 / padding 1
 * layers whose tiles are whole (Cin % 16, Cout % 128, N*H*W % 128, W % 4 == 0, input < 2 GB) then run on
 * the LDS-DMA ring kernel; every other geometry takes the path of qarig_conv2d_fwd.  Results differ
 * by the summation order only (tap-major instead of channel-major). */
size_t qarig_conv2d_fwd_workspace_bytes(int Cin, int Cout, int k);
/* ... plus, at few images (generate_images.py:366 decodes the 1-8 images just sampled), room for the
 * partial sums of a launch whose reduction is split 2-8 ways because its tiles alone would leave CUs
 * idle (<= 256 workgroups): given this much scratch the call splits, given only the size above it
 * does not.  The split changes the summation order (parts added in order after the k-loop). */
size_t qarig_conv2d_fwd_workspace_bytes_n(int N, int Cin, int H, int W, int Cout, int k, int stride);
/* flags: QARIG_CONV_PACKED_VALID = the head of `workspace` still holds the re-ordered weights an earlier
 * call with the same w, geometry and workspace wrote there (inference: the weights do not change between
 * calls) -- the re-ordering launch is skipped.  0 otherwise. */
#define QARIG_CONV_PACKED_VALID 1
int qarig_conv2d_fwd_ws(const float* x, int N, int Cin, int H, int W, const float* w,
                        const float* bias, int Cout, int k, int stride, int pad, int act, float* y,
                        float* preact, void* workspace, size_t ws_bytes, int flags, void* stream);

/* nn.ConvTranspose2d(4, stride 2, padding 1) + bias + activation -- UpsampleConvLayer,
 * models/layers.py:188-207.  x (N,Cin,H,W); w (Cin,Cout,4,4); y (N,Cout,2H,2W). */
size_t qarig_conv_transpose2d_workspace_bytes(int Cin, int Cout);
/* the same plus the split launch's slabs at few images (see qarig_conv2d_fwd_workspace_bytes_n) */
size_t qarig_conv_transpose2d_workspace_bytes_n(int N, int Cin, int H, int W, int Cout);
int qarig_conv_transpose2d_fwd(const float* x, int N, int Cin, int H, int W, const float* w,
                               const float* bias, int Cout, int act, float* y, float* preact,
                               void* workspace, size_t ws_bytes, int flags, void* stream);

/* autograd of the conv layers (dT = dy * act'(preact), via qarig_act_bwd, first) */
size_t qarig_conv2d_bwd_data_workspace_bytes(int Cin, int Cout, int k);
/* ... plus room for the split slabs of a few-image launch (3x3 / stride 1; see qarig_conv2d_fwd_workspace_bytes_n) */
size_t qarig_conv2d_bwd_data_workspace_bytes_n(int N, int Cin, int H, int W, int Cout, int k, int stride);
int qarig_conv2d_bwd_data(const float* dT, int N, int Cout, int Ho, int Wo, const float* w, int Cin,
                          int k, int stride, int pad, int H, int W, float* dx, void* workspace,
                          size_t ws_bytes, void* stream);
int qarig_conv_transpose2d_bwd_data(const float* dT, int N, int Cout, int H, int W, const float* w,
                                    int Cin, float* dx, void* stream);
/* ... with a scratch buffer (16 * Cin * Cout floats) for tap-major weights: whole-tile layers on the strided ring
 * kernel, every other geometry as above. */
/* scratch of the _ws form: the tap-major weights (qarig_conv_transpose2d_workspace_bytes) + the split slabs of a
 * few-image launch */
size_t qarig_conv_transpose2d_bwd_data_workspace_bytes_n(int N, int Cin, int H, int W, int Cout);
int qarig_conv_transpose2d_bwd_data_ws(const float* dT, int N, int Cout, int H, int W, const float* w,
                                       int Cin, float* dx, void* workspace, size_t ws_bytes, void* stream);
/* G (N,Cg,Gh,Gw) correlated with im2col_{k,stride,pad}(X (N,Cx,H,W)) -> dw (Cg, Cx*k*k).
 * Conv2d: G=dT, X=input.  ConvTranspose2d(4,2,1): G=input, X=dT, k=4, stride=2, pad=1. */
size_t qarig_conv_wgrad_workspace_bytes(int Cg, int K2, int P);
int qarig_conv_wgrad(const float* G, int N, int Cg, int Gh, int Gw, const float* X, int Cx, int H,
                     int W, int k, int stride, int pad, float* dw, void* workspace, size_t ws_bytes,
                     void* stream);
int qarig_conv_bias_grad(const float* G, int N, int C, int HW, float* db, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QARIG_H */
