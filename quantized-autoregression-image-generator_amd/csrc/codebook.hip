// Codebook side kernels besides the BMU search: patchify / unpatchify copies
// (models/layers.py:8-71), the fused gather + unpatchify of get_quantized_image
// (models/Codebook.py:138-154) and the Gaussian index-neighbourhood weights of
// get_quantized_patches (models/Codebook.py:112-126).
#include "qarig_common.h"

namespace qarig {

struct PGeom {
    int N, C, H, W, pH, pW, gh, gw, D;
};

// image element (n,c,y,x) <-> (patch row r, element e)
__device__ __forceinline__ void img_to_patch(const PGeom& g, int64_t idx, int64_t& r, int& e) {
    const int x = (int)(idx % g.W);
    int64_t t = idx / g.W;
    const int y = (int)(t % g.H);
    t /= g.H;
    const int c = (int)(t % g.C);
    const int n = (int)(t / g.C);
    const int ph = y / g.pH, i = y - ph * g.pH;
    const int pw = x / g.pW, j = x - pw * g.pW;
    r = ((int64_t)n * g.gh + ph) * g.gw + pw;
    e = (c * g.pH + i) * g.pW + j;
}

// dir 0: patches[r][e] = image[idx];  dir 1: image[idx] = patches[r][e]
__global__ void patch_copy_kernel(PGeom g, const float* __restrict__ src, float* __restrict__ dst,
                                  int dir) {
    const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r; int e;
        img_to_patch(g, idx, r, e);
        if (dir == 0) dst[r * g.D + e] = src[idx];
        else dst[idx] = src[r * g.D + e];
    }
}

// image[n][c][y][x] = W[idx[r]][e]
__global__ void gather_image_kernel(PGeom g, const int64_t* __restrict__ ids,
                                    const float* __restrict__ w, int K, float* __restrict__ img,
                                    int* __restrict__ bad) {
    const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r; int e;
        img_to_patch(g, idx, r, e);
        const int64_t id = ids[r];
        if (id < 0 || id >= K) { atomicExch(bad, 1); img[idx] = 0.0f; continue; }
        img[idx] = w[id * g.D + e];
    }
}

// out[r][:] = W[ids[r]][:]
__global__ void gather_rows_kernel(const int64_t* __restrict__ ids, int64_t R, int D, int K,
                                   const float* __restrict__ w, float* __restrict__ out,
                                   int* __restrict__ bad) {
    const int64_t total = R * D;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / D;
        const int e = (int)(idx - r * D);
        const int64_t id = ids[r];
        if (id < 0 || id >= K) { atomicExch(bad, 1); out[idx] = 0.0f; continue; }
        out[idx] = w[id * D + e];
    }
}

// g[r][j] = exp(-((j - bmu[r])^2 / two_var)), the division in fp32 as torch does for
// an int64 tensor divided by a python float.
__global__ void som_weights_kernel(const int64_t* __restrict__ bmu, int64_t R, int K, float two_var,
                                   float* __restrict__ g) {
    const int64_t total = R * K;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / K;
        const int64_t d = (idx - r * K) - bmu[r];
        g[idx] = expf(-((float)(d * d) / two_var));
    }
}

// out[j][:] = sum over b in [j - reach, j + reach] of exp(-(j-b)^2 / two_var) * in[b][:], b ascending.
// The Gaussian neighbourhood of models/Codebook.py:112-130 depends on the row only through its BMU,
// so (R,K) weights @ (K,D) codebook == gather of this (K,D) table at the BMUs, and its transpose
// product == this band over the per-code sums of the incoming gradient: R*K*D flops and an R x K
// matrix become K*(2 reach + 1)*D flops and two K x D tables.  Weights use the expression of
// som_weights_kernel; terms beyond `reach` are below 2^-40 of the centre weight (host picks it).
__global__ __launch_bounds__(256) void som_band_kernel(const float* __restrict__ in, int K, int D,
                                                       float two_var, int reach, float* __restrict__ out) {
    const int64_t total = (int64_t)K * D;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx / D), c = (int)(idx - (int64_t)j * D);
        const int lo = j - reach < 0 ? 0 : j - reach, hi = j + reach > K - 1 ? K - 1 : j + reach;
        float acc = 0.0f;
        for (int b = lo; b <= hi; ++b) {
            const int64_t d = j - b;
            acc = fmaf(expf(-((float)(d * d) / two_var)), in[(int64_t)b * D + c], acc);
        }
        out[idx] = acc;
    }
}

// counts[id] += 1 over a stream of BMU indices (prune_codebook.py:129-142 keeps a Python
// dict; here a device histogram).  Integer atomics: exact and order-independent.
__global__ void histogram_kernel(const int64_t* __restrict__ ids, int64_t n, int K,
                                 unsigned long long* __restrict__ counts, int* __restrict__ bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = ids[i];
        if (id < 0 || id >= K) { atomicExch(bad, 1); continue; }
        atomicAdd(counts + id, 1ULL);
    }
}

}  // namespace qarig

using namespace qarig;

static dim3 cb_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return dim3((unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)));
}

static int make_geom(PGeom& g, int N, int C, int H, int W, int pH, int pW) {
    QARIG_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && pH > 0 && pW > 0, "patch geometry: bad extents");
    QARIG_CHECK_ARG(H % pH == 0 && W % pW == 0, "patch geometry: H,W must be multiples of the patch");
    QARIG_CHECK_DIMS("patch geometry", N, C, H, W);
    QARIG_CHECK_DIMS("patch geometry", C, pH, pW);
    g = PGeom{N, C, H, W, pH, pW, H / pH, W / pW, C * pH * pW};
    return QARIG_OK;
}

// patchify, models/layers.py:8-34.  image (N,C,H,W) -> patches (N*Seq, C*pH*pW).
extern "C" int qarig_patchify_fwd(const float* image, int N, int C, int H, int W, int pH, int pW,
                                  float* patches, void* stream) {
    QARIG_CHECK_ARG(image && patches, "patchify: null pointer");
    PGeom g;
    if (int e = make_geom(g, N, C, H, W, pH, pW)) return e;
    hipLaunchKernelGGL(patch_copy_kernel, cb_grid((int64_t)N * C * H * W), dim3(256), 0,
                       (hipStream_t)stream, g, image, patches, 0);
    QARIG_CHECK_LAUNCH("patchify");
    return QARIG_OK;
}

// unpatchify, models/layers.py:37-71.
extern "C" int qarig_unpatchify_fwd(const float* patches, int N, int C, int H, int W, int pH, int pW,
                                    float* image, void* stream) {
    QARIG_CHECK_ARG(image && patches, "unpatchify: null pointer");
    PGeom g;
    if (int e = make_geom(g, N, C, H, W, pH, pW)) return e;
    hipLaunchKernelGGL(patch_copy_kernel, cb_grid((int64_t)N * C * H * W), dim3(256), 0,
                       (hipStream_t)stream, g, patches, image, 1);
    QARIG_CHECK_LAUNCH("unpatchify");
    return QARIG_OK;
}

// Codebook.get_quantized_image, models/Codebook.py:138-154 (gather + unpatchify fused).
extern "C" int qarig_codebook_gather_image(const int64_t* ids, int N, int C, int H, int W, int pH,
                                           int pW, const float* codebook, int K, float* image,
                                           int* bad_flag, void* stream) {
    QARIG_CHECK_ARG(ids && codebook && image && bad_flag && K > 0, "gather_image: bad arguments");
    PGeom g;
    if (int e = make_geom(g, N, C, H, W, pH, pW)) return e;
    hipLaunchKernelGGL(gather_image_kernel, cb_grid((int64_t)N * C * H * W), dim3(256), 0,
                       (hipStream_t)stream, g, ids, codebook, K, image, bad_flag);
    QARIG_CHECK_LAUNCH("gather_image");
    return QARIG_OK;
}

// nn.Embedding row gather (codebook rows; models/Codebook.py:132,144).
extern "C" int qarig_gather_rows(const int64_t* ids, int64_t R, int D, int K, const float* table,
                                 float* out, int* bad_flag, void* stream) {
    QARIG_CHECK_DIMS("gather_rows", R, D);
    QARIG_CHECK_DIMS("gather_rows", K, D);
    QARIG_CHECK_ARG(ids && table && out && bad_flag && R > 0 && D > 0 && K > 0,
                    "gather_rows: bad arguments");
    hipLaunchKernelGGL(gather_rows_kernel, cb_grid(R * D), dim3(256), 0, (hipStream_t)stream, ids, R,
                       D, K, table, out, bad_flag);
    QARIG_CHECK_LAUNCH("gather_rows");
    return QARIG_OK;
}

// Gaussian neighbourhood weights, models/Codebook.py:112-126.  two_var = 2*sigma^2.
extern "C" int qarig_som_weights_fwd(const int64_t* bmu, int64_t R, int K, float two_var, float* g,
                                     void* stream) {
    QARIG_CHECK_ARG(bmu && g && R > 0 && K > 0 && two_var > 0, "som_weights: bad arguments");
    QARIG_CHECK_DIMS("som_weights", R, K);
    hipLaunchKernelGGL(som_weights_kernel, cb_grid(R * K), dim3(256), 0, (hipStream_t)stream, bmu, R,
                       K, two_var, g);
    QARIG_CHECK_LAUNCH("som_weights");
    return QARIG_OK;
}

// The Gaussian neighbourhood applied to a (K,D) table along the code axis (see som_band_kernel):
// forward quantised rows = gather_rows(bmu, band(codebook)); codebook gradient = band(embedding_bwd(bmu, dq)).
extern "C" int qarig_som_band(const float* in, int K, int D, float two_var, int reach, float* out,
                              void* stream) {
    QARIG_CHECK_ARG(in && out && in != out && K > 0 && D > 0 && two_var > 0 && reach >= 0,
                    "som_band: bad arguments");
    QARIG_CHECK_DIMS("som_band", K, D);
    hipLaunchKernelGGL(som_band_kernel, cb_grid((int64_t)K * D), dim3(256), 0, (hipStream_t)stream, in, K,
                       D, two_var, reach, out);
    QARIG_CHECK_LAUNCH("som_band");
    return QARIG_OK;
}

// counts (int64 [K], caller-initialised) += histogram of ids -- the BMU usage count of
// prune_codebook.py:129-142.
extern "C" int qarig_index_histogram(const int64_t* ids, int64_t n, int K, int64_t* counts,
                                     int* bad_flag, void* stream) {
    QARIG_CHECK_ARG(ids && counts && bad_flag && n > 0 && K > 0, "index_histogram: bad arguments");
    QARIG_CHECK_ARG(n <= (1LL << 40) && K <= (1 << 24), "index_histogram: extents too large");
    hipLaunchKernelGGL(histogram_kernel, cb_grid(n), dim3(256), 0, (hipStream_t)stream, ids, n, K,
                       (unsigned long long*)counts, bad_flag);
    QARIG_CHECK_LAUNCH("index_histogram");
    return QARIG_OK;
}
