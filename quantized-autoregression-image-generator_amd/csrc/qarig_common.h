// Shared device/host helpers for the qarig HIP library (gfx950 / CDNA4 only).
//
// The contraction core in here is the one every matmul-shaped kernel of the hot
// path is built on (Linear fwd/bwd, BMU distance, SOM neighbourhood, conv as
// implicit GEMM): a 128x128x16 block tile, 4 waves in a 2x2 arrangement, each wave
// owning a 64x64 sub-tile as 2x2 v_mfma_f32_32x32x2_f32 accumulators.  fp32 in,
// fp32 accumulate: the MFMA is bit-for-bit a k-ordered fmaf chain, which is what
// the parity contract (<=1e-5 on pixels, bit-exact BMU indices) needs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QARIG_OK 0
#define QARIG_ERR_ARG -1
#define QARIG_ERR_LAUNCH -2
#define QARIG_ERR_WORKSPACE -3

extern "C" void qarig_set_error(const char* fmt, ...);

// csrc/decode.hip (include/qarig.h): the weight-streaming Linear of a single-token decode step; the
// skinny entry points of gemm.hip hand their <= 16-row calls to it.
extern "C" int qarig_gemm_tile64(int M, int N, int K);
extern "C" int qarig_decode_linear_supported(int M, int N, int K, int ln);
extern "C" int qarig_decode_linear_f32(const float* X, int64_t ldx, int64_t x_gs, float eps, const float* gamma,
                                       const float* beta, const float* scale, const float* shift, int64_t ldmod,
                                       const float* W, int64_t ldw, int64_t w_gs, const float* bias, int64_t bias_gs,
                                       const float* residual, int64_t ldr, const float* mul, int64_t ldmul, float* C,
                                       int64_t ldc, int64_t c_gs, int groups, int M, int N, int K, int act,
                                       void* stream);
extern "C" int qarig_decode_attention(const float* q, const float* k_new, const float* v_new, float* kcache,
                                      float* vcache, int B, int H, int d, int len, const int* len_dev, int max_len,
                                      int64_t batch_stride, int64_t head_stride, int64_t row_stride, float sqrt_d,
                                      const float* o_mul, int64_t ldmul, float* o, void* stream);

// Kernel-selection options (qarig_set_option): every one only chooses between kernels that must give
// the same results; -1 / 0 = the library's own choice where stated.  Defined in capi.hip.
struct QarigOptions {
    int gemm_dma = 1;      // 0: interior GEMM shapes on the register-staged kernel instead of the LDS-DMA ring
    int gemm_xcd_splits = 1;   // split reductions of a multiple of 8 splits: one split per XCD (1) or tiles per XCD (0)
    int gemm_pair = -1;    // paired (two-team) GEMM kernel: -1 auto (<= 256 workgroups), 0 never, 1 wherever eligible
    int bmu_cs = 0;        // resident BMU kernel, waves sharing a row tile: 0 auto, else 1 / 2 / 4
    int bmu_groups = -1;   // resident BMU kernel, group-minimum scan: -1 auto, 0 / 1
    int bmu_coarse = -1;   // coarse-pass BMU kernel: -1 auto (>= 24,576 rows), 0 never, 1 wherever it applies
    int attn_qw = 0;       // attention forward, waves per head: 0 auto, else 1 / 2 / 4
    int attn_bw = 0;       // attention backward, waves per head: 0 auto, else 1 / 2 / 4
    int lp_big = -1;       // reduced precision, 256 x 256 tiles: -1 auto (>= 224 tiles), 0 / 1
    int lp_mfma16 = 1;     // reduced precision, v_mfma_f32_16x16x32_bf16 (1) or 32x32x16 (0)
    int convt_pair = 1;    // ConvTranspose parity classes paired per workgroup where a class is < 512 workgroups
    int gemm_x3 = 0;       // fp32 GEMMs as six bf16-MFMA products of exact three-way operand splits (gemm_x3.hip): opt-in
                           // (1: 128 x 128 and 64 x 64 tiles; 2: the 128 x 128-tile form only -- the A/B of the other)
    int gemm_tile64 = -1;  // 64 x 64-tile GEMM: -1 auto (fewer than 192 tiles of 128 x 128), 0 never, 1 wherever eligible
    int decode_rows = 1;   // 5 ... 16-row decode Linear layers: rows split over the waves (1) or all rows in every lane (0)
    int decode_stream = 1; // decode-step Linear layers of <= 16 rows on decode_linear_kernel (1) or gemm_skinny_kernel (0)
    int conv_ring = 1;     // 0: every convolution on the gather (im2col-in-registers) kernels, none on the LDS-DMA ring
};
extern QarigOptions g_qarig_opt;

#define QARIG_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            qarig_set_error(__VA_ARGS__);          \
            return QARIG_ERR_ARG;                  \
        }                                          \
    } while (0)

#define QARIG_CHECK_LAUNCH(name)                                              \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) {                                              \
            qarig_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return QARIG_ERR_LAUNCH;                                          \
        }                                                                     \
    } while (0)

// Extents an entry point accepts: every one positive and at most `max_each`, their product at
// most `max_prod` -- after this check the host-side index / grid / byte arithmetic derived from
// them cannot overflow (tests/test_sanitizers.py drives every entry point under UBSan with
// INT_MAX-sized arguments).
#include <initializer_list>
static inline bool qarig_dims_ok(std::initializer_list<long long> dims, long long max_each = 1LL << 24,
                                 long long max_prod = 1LL << 40) {
    long long p = 1;
    for (long long d : dims) {
        if (d < 1 || d > max_each) return false;
        if (p > max_prod / d) return false;
        p *= d;
    }
    return true;
}
#define QARIG_CHECK_DIMS(what, ...)                                                             \
    QARIG_CHECK_ARG(qarig_dims_ok({__VA_ARGS__}),                                               \
                    what ": extents must be positive, at most 2^24 each, product at most 2^40")

namespace qarig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Activation ids shared with the host side (models/layers.py get_activation:
// reference layers.py:74-80).
enum Act : int { ACT_NONE = 0, ACT_SILU = 1, ACT_TANH = 2, ACT_SIGMOID = 3 };

// 1 / (1 + e^-x) on the hardware transcendentals: v_exp_f32 (2^t, ~1 ulp) on t = -x*log2(e)
// and v_rcp_f32 (~1 ulp).  Error vs the exact sigmoid: <= ~3e-7 relative (the product
// rounding contributes |x| * 2^-24 * ln2 only where e^-x matters), inside the 1e-5 parity
// budget with >30x margin; used in the GEMM / conv epilogues, where the libm expf + IEEE
// divide sequence was ~40 % of a K=512 tile's instruction stream.
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

__device__ __forceinline__ float act_fwd(float x, int act) {
    switch (act) {
        case ACT_SILU: return x * sigmoid_f(x);
        case ACT_TANH: return tanhf(x);
        case ACT_SIGMOID: return sigmoid_f(x);
        default: return x;
    }
}

// d act(x) / dx, from the pre-activation x.
__device__ __forceinline__ float act_grad(float x, int act) {
    switch (act) {
        case ACT_SILU: {
            float s = sigmoid_f(x);
            return s * (1.0f + x * (1.0f - s));
        }
        case ACT_TANH: {
            float t = tanhf(x);
            return 1.0f - t * t;
        }
        case ACT_SIGMOID: {
            float s = sigmoid_f(x);
            return s * (1.0f - s);
        }
        default: return 1.0f;
    }
}

// ---------------------------------------------------------------------------
// Block-tile geometry of the contraction core.
// ---------------------------------------------------------------------------
constexpr int BM = 128;       // rows of the block tile  (A side)
constexpr int BN = 128;       // cols of the block tile  (B side)
constexpr int BK = 16;        // reduction depth per staged tile
constexpr int LDT = BM + 4;   // LDS row stride in floats (16-B aligned rows; the +4
                              // makes the transposing ds_write_b32 at most 2-way,
                              // which costs nothing on gfx950)
constexpr int NTHREADS = 256; // 4 waves
constexpr int STAGE = 8;      // floats each thread stages per operand per tile
constexpr int TILE_FLOATS = BK * LDT;
constexpr int GEMM_LDS_FLOATS = 4 * TILE_FLOATS;  // A,B double-buffered

struct Acc {
    f32x16 t[2][2];  // [m-subtile][n-subtile] of the wave's 64x64
};

__device__ __forceinline__ void acc_zero(Acc& a) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) a.t[i][j][r] = 0.0f;
}

// Row (A-side index) inside a 32x32 accumulator tile held by this lane in
// register r; the column is lane & 31.  (C/D map of v_mfma_f32_32x32x2_f32.)
__device__ __forceinline__ int acc_row(int r, int lane) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// --- operand tile loaders ---------------------------------------------------
// A staged tile is BK x 128 in LDS, reduction-major ("k-major"): T[k][x], so that
// the MFMA fragment read (lane -> x = lane&31, k = 2*s + lane>>5) is a
// conflict-free ds_read_b32.  Three kinds of global source feed it.

// Source stored [X][K] (reduction index contiguous): Linear weights (N,K),
// activations (M,K).  Thread -> (x = tid>>1, 8 consecutive k).
struct SrcKContig {
    const float* p;
    int64_t ld;    // elements between consecutive x
    int X, K;      // extents (for guards)
    float scale;   // multiplied in while staging (exact for powers of two)
    bool vec4;     // rows 16-B aligned and ld % 4 == 0

    __device__ __forceinline__ void load(float (&r)[STAGE], int x0, int k0, int tid) const {
        const int x = x0 + (tid >> 1);
        const int k = k0 + (tid & 1) * 8;
        if (x < X && vec4 && k + 8 <= K) {
            const float4* q = reinterpret_cast<const float4*>(p + (int64_t)x * ld + k);
            float4 a = q[0], b = q[1];
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
            r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r[j] = (x < X && k + j < K) ? p[(int64_t)x * ld + k + j] : 0.0f;
        }
    }
    // whole 128 x [k_begin,k_end) panel in range and vector-loadable (block-uniform)
    __device__ __forceinline__ bool interior(int x0, int k_begin, int k_end) const {
        return vec4 && x0 + 128 <= X && k_end <= K && ((k_end - k_begin) % BK) == 0;
    }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        const float4* q = reinterpret_cast<const float4*>(
            p + (int64_t)(x0 + (tid >> 1)) * ld + k0 + (tid & 1) * 8);
        const float4 a = q[0], b = q[1];
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
        r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    }
    __device__ __forceinline__ void store(const float (&r)[STAGE], float* T, int tid) const {
        const int x = tid >> 1;
        const int k = (tid & 1) * 8;
        if (scale == 1.0f) {   // uniform: no multiplies beside the fp32 MFMAs where there is nothing to scale
#pragma unroll
            for (int j = 0; j < 8; ++j) T[(k + j) * LDT + x] = r[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) T[(k + j) * LDT + x] = r[j] * scale;
        }
    }
};

// Source stored [K][X] (tile index contiguous): the transposed operands of the
// backward contractions.  Thread -> (k = tid>>5 (+8), 4 consecutive x).
struct SrcXContig {
    const float* p;
    int64_t ld;    // elements between consecutive k
    int X, K;
    float scale;
    bool vec4;

    __device__ __forceinline__ void load(float (&r)[STAGE], int x0, int k0, int tid) const {
        const int x = x0 + (tid & 31) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + (tid >> 5) + 8 * h;
            if (k < K && vec4 && x + 4 <= X) {
                float4 a = *reinterpret_cast<const float4*>(p + (int64_t)k * ld + x);
                r[4 * h + 0] = a.x; r[4 * h + 1] = a.y; r[4 * h + 2] = a.z; r[4 * h + 3] = a.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    r[4 * h + j] = (k < K && x + j < X) ? p[(int64_t)k * ld + x + j] : 0.0f;
            }
        }
    }
    __device__ __forceinline__ bool interior(int x0, int k_begin, int k_end) const {
        return vec4 && x0 + 128 <= X && k_end <= K && ((k_end - k_begin) % BK) == 0;
    }
    __device__ __forceinline__ void load_fast(float (&r)[STAGE], int x0, int k0, int tid) const {
        const float* q = p + (int64_t)(k0 + (tid >> 5)) * ld + x0 + (tid & 31) * 4;
        const float4 a = *reinterpret_cast<const float4*>(q);
        const float4 b = *reinterpret_cast<const float4*>(q + 8 * ld);
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
        r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    }
    __device__ __forceinline__ void store(const float (&r)[STAGE], float* T, int tid) const {
        const int x = (tid & 31) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = (tid >> 5) + 8 * h;
            float4 v = make_float4(r[4 * h] * scale, r[4 * h + 1] * scale, r[4 * h + 2] * scale,
                                   r[4 * h + 3] * scale);
            *reinterpret_cast<float4*>(T + k * LDT + x) = v;
        }
    }
};

// One BK-deep slab of MFMAs from staged tiles TA/TB for this wave.  KS = number of
// 2-deep MFMA steps, a compile-time constant on the main path so the whole slab is
// straight-line code: all fragment reads are issued up front and the compiler waits
// with counted lgkmcnt, never on the global prefetch that is in flight.
template <int KS>
__device__ __forceinline__ void mma_tile(Acc& acc, const float* TA, const float* TB, int wm,
                                         int wn, int lane) {
    const int x = lane & 31;
    const int h = lane >> 5;
    // volatile: one ds_read_b32 with a 16-bit immediate offset per value.  Left to itself hipcc pairs
    // the reads into ds_read2_b32, whose 8-bit offsets do not reach the next k row (132 floats), and
    // pays one vector add per pair for a new base -- vector-ALU cycles the fp32 MFMA does not hide.
    typedef const volatile __attribute__((address_space(3))) float* lds_cvf_t;
    lds_cvf_t pa = (lds_cvf_t)(TA + h * LDT + wm * 64 + x);
    lds_cvf_t pb = (lds_cvf_t)(TB + h * LDT + wn * 64 + x);
    // fragment reads run one k-step ahead of the MFMAs that consume them
    float a0 = pa[0], a1 = pa[32], b0 = pb[0], b1 = pb[32];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
        if (s + 1 < KS) {
            na0 = pa[(2 * s + 2) * LDT];
            na1 = pa[(2 * s + 2) * LDT + 32];
            nb0 = pb[(2 * s + 2) * LDT];
            nb1 = pb[(2 * s + 2) * LDT + 32];
        }
        acc.t[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc.t[0][0], 0, 0, 0);
        acc.t[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc.t[0][1], 0, 0, 0);
        acc.t[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc.t[1][0], 0, 0, 0);
        acc.t[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc.t[1][1], 0, 0, 0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    // pin that order (hipcc otherwise sinks each read pair back in front of its MFMAs):
    // reads(0) | { reads(s+1), 4 MFMA(s) } x (KS-1) | 4 MFMA(KS-1)
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int s = 0; s + 1 < KS; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
}

// Ragged tail (reduction extent not a multiple of BK): runtime step count.
__device__ __forceinline__ void mma_tile_tail(Acc& acc, const float* TA, const float* TB, int wm,
                                              int wn, int lane, int ksteps) {
    const int x = lane & 31;
    const int h = lane >> 5;
    const float* pa = TA + h * LDT + wm * 64 + x;
    const float* pb = TB + h * LDT + wn * 64 + x;
    for (int s = 0; s < ksteps; ++s) {
        const float a0 = pa[(2 * s) * LDT];
        const float a1 = pa[(2 * s) * LDT + 32];
        const float b0 = pb[(2 * s) * LDT];
        const float b1 = pb[(2 * s) * LDT + 32];
        acc.t[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc.t[0][0], 0, 0, 0);
        acc.t[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc.t[0][1], 0, 0, 0);
        acc.t[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc.t[1][0], 0, 0, 0);
        acc.t[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc.t[1][1], 0, 0, 0);
    }
}

// acc += A[m0:m0+128, k_begin:k_end] * B[n0:n0+128, k_begin:k_end]^T through
// double-buffered LDS tiles.  `lds` holds GEMM_LDS_FLOATS floats.  All threads of
// the block must call it; it ends with the tiles no longer in use (a barrier has
// been passed), so the caller may reuse `lds`.
//
// FAST (block-uniform): both panels are interior and vector-loadable, so the loop is
// branch-free straight-line code -- the prefetch of tile t+1 is issued before the MFMA
// slab of tile t and only waited for (vmcnt) at its LDS write, after the slab.
//
// hook(TA, TB) is called once per staged tile pair (tiles in ascending k order, zero-filled
// past the extents), between barriers, by every thread: side reductions that need the
// operands anyway ride along with the contraction -- the bias gradient (row sums of
// A = dT^T) in the weight-gradient GEMM, the |w|^2 / |x|^2 chains in the BMU search.
struct NoHook {
    __device__ __forceinline__ void operator()(const float*, const float*) {}
};

template <bool FAST, class SA, class SB, class Hook>
__device__ __forceinline__ void contract_loop(Acc& acc, const SA& sa, const SB& sb, int m0, int n0,
                                              int k_begin, int k_end, float* lds, Hook& hook) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    float* TA0 = lds;
    float* TA1 = lds + TILE_FLOATS;
    float* TB0 = lds + 2 * TILE_FLOATS;
    float* TB1 = lds + 3 * TILE_FLOATS;

    float ra[STAGE], rb[STAGE];
    const int nk = (k_end - k_begin + BK - 1) / BK;
    if (nk <= 0) return;
    if (FAST) {
        // Software pipeline, one register set, two LDS buffers:
        //   slab t, first half | write tile t+1 (loaded half a slab + one slab ago) to the
        //   other buffer | issue the loads of tile t+2 | slab t, second half | barrier.
        // The LDS write and the global latency both sit under MFMAs; only the barrier and
        // the first fragment read separate two slabs.
        const int last = k_begin + (nk - 1) * BK;
        sa.load_fast(ra, m0, k_begin, tid);
        sb.load_fast(rb, n0, k_begin, tid);
        sa.store(ra, TA0, tid);
        sb.store(rb, TB0, tid);
        const int k1 = min(k_begin + BK, last);
        sa.load_fast(ra, m0, k1, tid);
        sb.load_fast(rb, n0, k1, tid);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int kb = k_begin + kt * BK;
            float* ta = (kt & 1) ? TA1 : TA0;
            float* tb = (kt & 1) ? TB1 : TB0;
            float* na = (kt & 1) ? TA0 : TA1;
            float* nb = (kt & 1) ? TB0 : TB1;
            mma_tile<BK / 4>(acc, ta, tb, wm, wn, lane);
            hook(ta, tb);
            __builtin_amdgcn_sched_barrier(0);
            sa.store(ra, na, tid);       // tile kt+1 (or a harmless repeat of the last tile)
            sb.store(rb, nb, tid);
            const int kn = min(kb + 2 * BK, last);
            sa.load_fast(ra, m0, kn, tid);
            sb.load_fast(rb, n0, kn, tid);
            __builtin_amdgcn_sched_barrier(0);
            mma_tile<BK / 4>(acc, ta + (BK / 2) * LDT, tb + (BK / 2) * LDT, wm, wn, lane);
            __syncthreads();
        }
        return;
    }
    // Guarded path (ragged extents, gathered operands): same shape of pipeline -- the next
    // tile's (predicated, zero-filling) loads are issued above the MFMA slab and written to
    // LDS below it; on the last tile the loads repeat that tile so the loop has no
    // data-dependent branch for hipcc to serialise behind a vmcnt(0).
    sa.load(ra, m0, k_begin, tid);
    sb.load(rb, n0, k_begin, tid);
    sa.store(ra, TA0, tid);
    sb.store(rb, TB0, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int kb = k_begin + kt * BK;
        const int kn = kt + 1 < nk ? kb + BK : kb;
        float* ta = (kt & 1) ? TA1 : TA0;
        float* tb = (kt & 1) ? TB1 : TB0;
        float* na = (kt & 1) ? TA0 : TA1;
        float* nb = (kt & 1) ? TB0 : TB1;
        sa.load(ra, m0, kn, tid);
        sb.load(rb, n0, kn, tid);
        __builtin_amdgcn_sched_barrier(0);
        const int rem = k_end - kb;
        if (rem >= BK) mma_tile<BK / 2>(acc, ta, tb, wm, wn, lane);
        else mma_tile_tail(acc, ta, tb, wm, wn, lane, (rem + 1) >> 1);
        hook(ta, tb);
        __builtin_amdgcn_sched_barrier(0);
        sa.store(ra, na, tid);
        sb.store(rb, nb, tid);
        __syncthreads();
    }
}

template <bool FAST, class SA, class SB>
__device__ __forceinline__ void contract_loop(Acc& acc, const SA& sa, const SB& sb, int m0, int n0,
                                              int k_begin, int k_end, float* lds) {
    NoHook none;
    contract_loop<FAST>(acc, sa, sb, m0, n0, k_begin, k_end, lds, none);
}

// XCD-aware remap of a linear block id (T1 of the CDNA4 guide, bijective form):
// blocks b and b+8 share an XCD, so give each XCD a contiguous chunk of the tile
// order and neighbouring tiles (which share operand panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace qarig
