"""Raw (non-autograd) wrappers: one Python function per C-ABI entry point.

Shapes/dtypes/contiguity are validated here, before the call, so the C side only
sees well-formed arguments (SURVEY.md 8b "Errors").
"""
import torch

from . import _lib
from ._lib import check, f32c, ptr, require_cuda, stream, workspace

ACT_IDS = {None: 0, "none": 0, "silu": 1, "tanh": 2, "sigmoid": 3}


def act_id(name):
    if name not in ACT_IDS:
        raise KeyError(name)  # same failure mode as the reference's ModuleDict lookup
    return ACT_IDS[name]


# --------------------------------------------------------------------------- BMU
def bmu(x, codebook, patch_dim):
    """int64 (N*Seq,) best-matching-unit indices of every patch of x (N,C,H,W).

    Replaces patchify + torch.cdist + torch.argmin in Codebook.get_patches_bmu
    (reference models/Codebook.py:77-99)."""
    require_cuda(x, codebook)
    x = f32c(x)
    codebook = f32c(codebook)
    N, C, H, W = x.shape
    pH, pW = patch_dim
    K, D = codebook.shape
    rows = N * (H // pH) * (W // pW)
    out = torch.empty(rows, dtype=torch.int64, device=x.device)
    lib = _lib.load()
    nb = lib.qarig_bmu_workspace_bytes(rows, K)
    ws = workspace(nb, x.device)
    check(lib.qarig_bmu_fwd(ptr(x), N, C, H, W, pH, pW, ptr(codebook), K, D, ptr(out), ptr(ws),
                            ws.numel(), stream()), "qarig_bmu_fwd")
    return out


# -------------------------------------------------------------------------- GEMM
def gemm(A, B, a_kcontig=True, b_kcontig=True, bias=None, residual=None, want_preact=False,
         act=0, gradz=None, gact=0, splitk=1, out=None):
    """C[M,N] = epilogue(sum_k A(m,k) B(n,k)); see include/qarig.h qarig_gemm_f32.

    A is (M,K) if a_kcontig else (K,M); B is (N,K) if b_kcontig else (K,N).
    Returns C, or (C, preact) when want_preact."""
    require_cuda(A, B, bias, residual, gradz)
    assert A.dim() == 2 and B.dim() == 2 and A.dtype == torch.float32 and B.dtype == torch.float32
    assert A.stride(1) == 1 and B.stride(1) == 1, "operands must be row-contiguous"
    M, K = (A.shape if a_kcontig else (A.shape[1], A.shape[0]))
    N, K2 = (B.shape if b_kcontig else (B.shape[1], B.shape[0]))
    assert K == K2, f"reduction mismatch {K} vs {K2}"
    C = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    assert C.shape == (M, N) and C.stride(1) == 1
    pre = torch.empty((M, N), dtype=torch.float32, device=A.device) if want_preact else None
    if bias is not None:
        assert bias.shape == (N,) and bias.is_contiguous()
    for t in (residual, gradz):
        if t is not None:
            assert t.shape == (M, N) and t.stride(1) == 1
    lib = _lib.load()
    ws = None
    nws = 0
    if splitk > 1:
        nws = lib.qarig_gemm_workspace_bytes(M, N, splitk)
        ws = workspace(nws, A.device, "gemm")
        nws = ws.numel()
    check(lib.qarig_gemm_f32(
        ptr(A), A.stride(0), int(a_kcontig), ptr(B), B.stride(0), int(b_kcontig),
        ptr(C), C.stride(0), M, N, K, ptr(bias),
        ptr(residual), residual.stride(0) if residual is not None else 0,
        ptr(pre), pre.stride(0) if pre is not None else 0, act,
        ptr(gradz), gradz.stride(0) if gradz is not None else 0, gact,
        splitk, ptr(ws), nws, stream()), "qarig_gemm_f32")
    return (C, pre) if want_preact else C


def colsum(X):
    """(N,) column sums of X (M,N) in a fixed order."""
    require_cuda(X)
    assert X.dim() == 2 and X.stride(1) == 1 and X.dtype == torch.float32
    M, N = X.shape
    out = torch.empty(N, dtype=torch.float32, device=X.device)
    lib = _lib.load()
    nb = lib.qarig_colsum_workspace_bytes(M, N)
    ws = workspace(nb, X.device, "colsum")
    check(lib.qarig_colsum_f32(ptr(X), X.stride(0), M, N, ptr(out), ptr(ws), ws.numel(), stream()),
          "qarig_colsum_f32")
    return out


def pick_splitk(M, N, K):
    """Reduction split for weight-gradient shaped GEMMs (small M x N, long K)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 256 or K < 2048:
        return 1
    s = min(max(1, 512 // tiles), K // 512)
    return max(1, min(s, 32))
