"""Raw (non-autograd) wrappers: one Python function per C-ABI entry point.

Shapes/dtypes/contiguity are validated here, before the call, so the C side only
sees well-formed arguments (SURVEY.md 8b "Errors").
"""
import os

import torch

from . import _lib
from ._lib import check, f32c, ptr, require_cuda, stream, workspace

ACT_IDS = {None: 0, "none": 0, "silu": 1, "tanh": 2, "sigmoid": 3}


def act_id(name):
    if name not in ACT_IDS:
        raise KeyError(name)  # same failure mode as the reference's ModuleDict lookup
    return ACT_IDS[name]


# --------------------------------------------------------------------------- BMU
_bmu_images = {}
_bmu_seen = {}               # id(codebook) -> state key of the last search (bmu_image only_if_stable)
BMU_IMAGE_MIN_ROWS = 24576   # the dispatch takes the coarse-pass kernel (the image's consumer) from here up


def bmu_invalidate():
    """Forget every prepared codebook image (after a write into parameters that neither torch's version counter
    nor FlatAdam's step count sees, e.g. parallel.broadcast_params)."""
    _bmu_images.clear()
    _bmu_seen.clear()


def _bmu_key(codebook):
    owner = getattr(codebook, "_qarig_owner", None)
    K, D = codebook.shape
    return (codebook.data_ptr(), K, D, codebook._version, owner.step_count if owner is not None else -1)


def bmu_image(codebook, only_if_stable=False):
    """Prepared image of a codebook for the coarse-pass search (include/qarig.h qarig_bmu_prepare), or
    None where that form does not apply.  Cached per codebook tensor until it changes: torch's version
    counter, and -- for a parameter FlatAdam owns, whose updates bypass that counter -- the optimiser's
    step count.  only_if_stable: build the image only for a codebook that an earlier call already saw in
    this state (a codebook under training changes between searches: preparing it every time would cost a
    launch for nothing)."""
    K, D = codebook.shape
    lib = _lib.load()
    nb = lib.qarig_bmu_prepare_bytes(K, D)
    if nb == 0 or codebook.data_ptr() % 16:
        return None
    key = _bmu_key(codebook)
    hit = _bmu_images.get(id(codebook))
    if hit is not None and hit[0] == key and hit[1]() is codebook:
        return hit[2]
    if torch.cuda.is_current_stream_capturing():
        return None                       # no allocation / caching inside a capture: the kernel stages by itself
    if only_if_stable:
        seen = _bmu_seen.get(id(codebook))
        _bmu_seen[id(codebook)] = key
        if len(_bmu_seen) > 256:
            _bmu_seen.clear()
        if seen != key:
            return None
    import weakref
    img = torch.empty(nb, dtype=torch.uint8, device=codebook.device)
    check(lib.qarig_bmu_prepare(ptr(codebook), K, D, ptr(img), stream()), "qarig_bmu_prepare")
    if len(_bmu_images) > 64:
        _bmu_images.clear()
    _bmu_images[id(codebook)] = (key, weakref.ref(codebook), img)
    return img


def bmu(x, codebook, patch_dim):
    """int64 (N*Seq,) best-matching-unit indices of every patch of x (N,C,H,W).

    Replaces patchify + torch.cdist + torch.argmin in Codebook.get_patches_bmu
    (reference models/Codebook.py:77-99)."""
    require_cuda(x, codebook)
    x = f32c(x)
    cb = codebook if (codebook.dtype == torch.float32 and codebook.is_contiguous()) else f32c(codebook)
    N, C, H, W = x.shape
    pH, pW = patch_dim
    K, D = cb.shape
    rows = N * (H // pH) * (W // pW)
    out = torch.empty(rows, dtype=torch.int64, device=x.device)
    lib = _lib.load()
    # Large launches take the coarse-pass kernel inside qarig_bmu_fwd.  A codebook that an earlier search saw in
    # its present state (frozen: tokenising a dataset, the Transformer training loop) goes in as its prepared
    # image, which the workgroups DMA into LDS instead of each converting the codebook (65,536 x 512 x 16:
    # 14.3 -> 13.x us).  Never inside a graph capture: a replay would search the image of the codebook as it was
    # when the graph was captured.
    # Only for an nn.Parameter: its identity is stable, torch's version counter sees every in-place update
    # through torch, FlatAdam's step count (p._qarig_owner) those made through the flat buffer.
    img = None
    if rows >= BMU_IMAGE_MIN_ROWS and cb is codebook and isinstance(codebook, torch.nn.Parameter) and \
            not torch.cuda.is_current_stream_capturing():
        img = bmu_image(cb, only_if_stable=True)
    nb = lib.qarig_bmu_workspace_bytes(rows, K)
    ws = workspace(nb, x.device)
    check(lib.qarig_bmu_fwd_prepared(ptr(x), N, C, H, W, pH, pW, ptr(cb), K, D, ptr(out), ptr(ws),
                                     ws.numel(), ptr(img), stream()), "qarig_bmu_fwd_prepared")
    return out


def bmu_coarse(x, codebook, patch_dim, prepared=False):
    """(indices, rows that needed the exact re-scan): the coarse-pass form of `bmu` forced
    (include/qarig.h qarig_bmu_fwd_coarse); raises where it does not apply.  prepared: through the
    cached codebook image (bmu_image) instead of staging the codebook in every workgroup."""
    require_cuda(x, codebook)
    x = f32c(x)
    codebook = f32c(codebook)
    N, C, H, W = x.shape
    pH, pW = patch_dim
    K, D = codebook.shape
    out = torch.empty(N * (H // pH) * (W // pW), dtype=torch.int64, device=x.device)
    cnt = torch.zeros(8, dtype=torch.int32, device=x.device)    # [0] re-scanned rows, [1..4] phase clocks
    img = bmu_image(codebook) if prepared else None
    check(_lib.load().qarig_bmu_fwd_coarse(ptr(x), N, C, H, W, pH, pW, ptr(codebook), K, D, ptr(out), ptr(cnt),
                                           ptr(img), stream()), "qarig_bmu_fwd_coarse")
    return out, cnt[:1] if not BMU_COARSE_PHASES else cnt


BMU_COARSE_PHASES = False      # tools: return all eight counters


# -------------------------------------------------------------------------- GEMM
# bench.py sets this to a list to bracket every GEMM launch with HIP events on the
# launch stream: entries are (flops, start_event, end_event).
GEMM_EVENTS = None

# "f32" (default; the parity mode: exact fp32 fma chains on the fp32 MFMA) or "bf16"
# (opt-in: Linear-layer contractions on the bf16 MFMA with fp32 accumulation and fp32
# tensors; BASELINE config 5 direction, tolerance stated in tests/test_gpu_bf16.py).
# QARIG_PRECISION in the environment sets the default.
PRECISION = os.environ.get("QARIG_PRECISION", "f32")
PRECISIONS = ("f32", "bf16", "fp8")


def lp_mode():
    """bf16 storage / bf16-MFMA nodes active ("bf16", and "fp8" which adds e4m3 forward products)."""
    return PRECISION in ("bf16", "fp8")


def set_precision(name):
    """Selects the contraction precision of the Linear / attention products for the process."""
    global PRECISION
    if name not in PRECISIONS:
        raise ValueError(f"qarig.ops precision must be one of {PRECISIONS}, not {name!r}")
    PRECISION = name


def gemm(A, B, a_kcontig=True, b_kcontig=True, bias=None, residual=None, want_preact=False,
         act=0, gradz=None, gact=0, splitk=None, out=None, accumulate=False, a_rowsum=None,
         flop_frac=1.0):
    """C[M,N] = epilogue(sum_k A(m,k) B(n,k)); see include/qarig.h qarig_gemm_f32.
    flop_frac: fraction of the 2MNK products that are algorithmic work (zero-padded operands
    of the ragged classifier): only bench.py's FLOP accounting reads it.

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
    if splitk is None:
        splitk = auto_splitk(M, N, K)
    lib = _lib.load()
    ws = None
    nws = 0
    if a_rowsum is not None:
        assert a_rowsum.shape == (M,) and a_rowsum.is_contiguous() and a_rowsum.dtype == torch.float32
    if splitk > 1 or a_rowsum is not None:
        nws = lib.qarig_gemm_workspace_bytes(M, N, splitk)
        ws = workspace(nws, A.device, "gemm")
        nws = ws.numel()
    if lp_mode():
        done = _gemm_lp(lib, A, B, a_kcontig, b_kcontig, C, pre, M, N, K, bias, residual, act, gradz,
                        gact, splitk, accumulate, a_rowsum)
        if done:
            return (C, pre) if want_preact else C
    elif PRECISION != "f32":
        raise ValueError(f"qarig.ops.PRECISION must be one of {PRECISIONS}, not {PRECISION!r}")
    fn, fn_name = lib.qarig_gemm_f32, "qarig_gemm_f32"
    if GEMM_EVENTS is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(fn(
        ptr(A), A.stride(0), int(a_kcontig), ptr(B), B.stride(0), int(b_kcontig),
        ptr(C), C.stride(0), M, N, K, ptr(bias),
        ptr(residual), residual.stride(0) if residual is not None else 0,
        ptr(pre), pre.stride(0) if pre is not None else 0, act,
        ptr(gradz), gradz.stride(0) if gradz is not None else 0, gact,
        splitk, int(accumulate), ptr(a_rowsum), ptr(ws), nws, stream()), fn_name)
    if GEMM_EVENTS is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        outs = "+".join(n for n, t in (("C", C), ("P", pre)) if t is not None)
        ins = "".join(n for n, t in (("b", bias), ("r", residual), ("z", gradz), ("s", a_rowsum)) if t is not None)
        hbm = 4.0 * (M * K + N * K + M * N * ((2 if accumulate and splitk == 1 else 1) + (pre is not None) +
                                              (residual is not None) + (gradz is not None)))
        GEMM_EVENTS.append((2.0 * M * N * K * flop_frac, ev0, ev1, "",
                            f"f32 {'N' if a_kcontig else 'T'}{'T' if b_kcontig else 'N'} {M}x{N}x{K} sk{splitk} "
                            f"out {outs} in {ins or '-'} act{act}{' acc' if accumulate else ''}", hbm))
    return (C, pre) if want_preact else C


# ---- reduced precision (BASELINE config 5): bf16 operands in HBM -------------------------------
# bf16 shadows of tensors that several GEMMs read (weights, in both layouts): keyed by storage,
# shape, torch's in-place version counter and LP_EPOCH, which qarig.optim bumps whenever its
# Adam kernel rewrites the parameters behind torch's back.
LP_EPOCH = 0
_lp_cache = {}


def lp_invalidate():
    """Every key carries LP_EPOCH, so all entries are dead after the bump: drop them (and the device
    memory they hold) right away."""
    global LP_EPOCH
    LP_EPOCH += 1
    _lp_cache.clear()


def _lp_get(key, src):
    """Cached shadow of `src` under `key`, or None.  Keys name a tensor by address / shape / version
    only, and an address can be handed to another tensor of the same shape once its owner is freed
    (a second model built in the same process): an entry therefore also holds a weak reference to
    the tensor it was made from and only serves that very tensor."""
    hit = _lp_cache.get(key)
    if hit is None:
        return None
    if hit[0]() is not src:
        del _lp_cache[key]
        return None
    return hit[1]


_LP_CACHE_MAX = 1024     # entries; a process that keeps meeting new weights / geometries starts over


def _lp_put(key, src, value):
    import weakref
    if len(_lp_cache) >= _LP_CACHE_MAX:
        _lp_cache.clear()
    _lp_cache[key] = (weakref.ref(src), value)


def cast_bf16(x, cache=False):
    """bf16 copy (torch.bfloat16, same shape) of a dense fp32 tensor, rounded to nearest even."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    key = None
    if cache and not torch.cuda.is_current_stream_capturing():
        src = getattr(x, "_qarig_src", x)
        key = ("n", x.data_ptr(), tuple(x.shape), x._version, LP_EPOCH)
        hit = _lp_get(key, src)
        if hit is not None:
            return hit
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(_lib.load().qarig_cast_bf16(ptr(x), ptr(out), x.numel(), stream()), "qarig_cast_bf16")
    if key is not None:
        _lp_put(key, src, out)
    return out


def cast_fp8(x, cache=False, want_bf16=False):
    """(e4m3 bytes as torch.uint8 of x's shape, dequantisation factor as a 1-element fp32 tensor
    [, bf16 copy of x from the same pass]) of a dense fp32 tensor: one scale for the tensor, 448 / max|x| (include/qarig.h
    qarig_cast_fp8).  The factor stays on the device; the GEMM epilogue multiplies by it."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() % 4 == 0
    key = None
    if cache and not torch.cuda.is_current_stream_capturing():
        src = getattr(x, "_qarig_src", x)
        key = ("f8", x.data_ptr(), tuple(x.shape), x._version, LP_EPOCH)
        hit = _lp_get(key, src)
        if hit is not None:
            return hit
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    aux = torch.empty(2, dtype=torch.float32, device=x.device)     # [0] inv_scale, [1] amax bits
    xb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    check(_lib.load().qarig_cast_fp8(ptr(x), x.numel(), ptr(out), ptr(aux), aux.data_ptr() + 4, ptr(xb),
                                     stream()), "qarig_cast_fp8")
    res = (out, aux[:1], xb) if want_bf16 else (out, aux[:1])
    if key is not None:
        _lp_put(key, src, res)
    return res


def f8_supported(M, N, K):
    return bool(_lib.load().qarig_gemm_f8_supported(M, N, K))


def gemm_f8(A8, inv_a, B8, inv_b, M, N, K, C=None, bias=None, residual=None, preact=None, act=0,
            Cb=None, Pb=None):
    """Forward product on e4m3 operands (include/qarig.h qarig_gemm_f8): C = epilogue(inv_a * inv_b *
    A8 B8^T), A8 (M,K) and B8 (N,K) uint8 from cast_fp8."""
    if GEMM_EVENTS is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.load().qarig_gemm_f8(
        ptr(A8), A8.stride(0), ptr(B8), B8.stride(0), ptr(inv_a), ptr(inv_b),
        ptr(C), C.stride(0) if C is not None else 0, M, N, K, ptr(bias),
        ptr(residual), residual.stride(0) if residual is not None else 0,
        ptr(preact), preact.stride(0) if preact is not None else 0, act,
        ptr(Cb), Cb.stride(0) if Cb is not None else 0, ptr(Pb), Pb.stride(0) if Pb is not None else 0,
        stream()), "qarig_gemm_f8")
    if GEMM_EVENTS is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        GEMM_EVENTS.append((2.0 * M * N * K, ev0, ev1, "f8"))


def cast_transpose_bf16(x, cache=False):
    """bf16 (C, R) transpose of a row-contiguous fp32 (R, C) matrix."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    key = None
    if cache and not torch.cuda.is_current_stream_capturing():
        src = getattr(x, "_qarig_src", x)
        key = ("t", x.data_ptr(), tuple(x.shape), x.stride(0), x._version, LP_EPOCH)
        hit = _lp_get(key, src)
        if hit is not None:
            return hit
    R, Cc = x.shape
    out = torch.empty((Cc, R), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().qarig_cast_transpose_bf16(ptr(x), x.stride(0), R, Cc, ptr(out), stream()),
          "qarig_cast_transpose_bf16")
    if key is not None:
        _lp_put(key, src, out)
    return out


def cast_colsum(x, colsum_out, accumulate=False, want_cast=True):
    """One pass over x (M,N) (fp32, or bf16 with want_cast False): column sums (+)= into
    colsum_out (fp32 (N,)) -- the bias gradient -- and, for fp32 input, the bf16 copy of x."""
    M, N = x.shape
    assert x.stride(1) == 1 and colsum_out.shape == (N,) and colsum_out.is_contiguous()
    is_bf = x.dtype == torch.bfloat16
    out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device) if (want_cast and not is_bf) else None
    lib = _lib.load()
    ws = workspace(lib.qarig_cast_colsum_workspace_bytes(M, N), x.device, "colsum")
    check(lib.qarig_cast_colsum(ptr(x), x.stride(0), int(is_bf), M, N, ptr(out), ptr(colsum_out),
                                int(accumulate), ptr(ws), ws.numel(), stream()), "qarig_cast_colsum")
    return out


def _is_param_like(t):
    """Weights (cached shadows) vs activations (cast per call)."""
    return isinstance(t, torch.nn.Parameter) or getattr(t, "_qarig_weight", False)


def gemm_lp(A_bf, B_bf, layout, M, N, K, C=None, bias=None, residual=None, preact=None, act=0, gradz=None,
            gact=0, splitk=1, accumulate=False, Cb=None, Pb=None):
    """Raw reduced-precision GEMM on bf16 operands (include/qarig.h qarig_gemm_lp)."""
    lib = _lib.load()
    ws, nws = None, 0
    if splitk > 1:
        nws = lib.qarig_gemm_lp_workspace_bytes(M, N, splitk)
        ws = workspace(nws, A_bf.device, "gemm")
        nws = ws.numel()
    if GEMM_EVENTS is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib.qarig_gemm_lp(
        ptr(A_bf), A_bf.stride(0), ptr(B_bf), B_bf.stride(0), layout,
        ptr(C), C.stride(0) if C is not None else 0, M, N, K, ptr(bias),
        ptr(residual), residual.stride(0) if residual is not None else 0,
        ptr(preact), preact.stride(0) if preact is not None else 0, act,
        ptr(gradz), gradz.stride(0) if gradz is not None else 0,
        int(gradz is not None and gradz.dtype == torch.bfloat16), gact, splitk, int(accumulate),
        ptr(Cb), Cb.stride(0) if Cb is not None else 0, ptr(Pb), Pb.stride(0) if Pb is not None else 0,
        ptr(ws), nws, stream()), "qarig_gemm_lp")
    if GEMM_EVENTS is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        outs = "+".join(n for n, t in (("C", C), ("P", preact), ("Cb", Cb), ("Pb", Pb)) if t is not None)
        ins = "".join(n for n, t in (("b", bias), ("r", residual), ("z", gradz)) if t is not None)
        hbm = 2.0 * (M * K + N * K) + M * N * (4.0 * (C is not None) * (2 if accumulate and splitk == 1 else 1) +
                                               4.0 * (preact is not None) + 2.0 * (Cb is not None) +
                                               2.0 * (Pb is not None) + 4.0 * (residual is not None) +
                                               (0 if gradz is None else gradz.element_size()))
        GEMM_EVENTS.append((2.0 * M * N * K, ev0, ev1, "",
                            f"lp {('NT', 'TN', 'NN')[layout]} {M}x{N}x{K} sk{splitk} out {outs} in {ins or '-'} act{act}", hbm))


def _gemm_lp(lib, A, B, a_kcontig, b_kcontig, C, pre, M, N, K, bias, residual, act, gradz, gact, splitk,
             accumulate, a_rowsum):
    """The fp32-tensor GEMM contract on the bf16 MFMA: operands are cast to bf16 in HBM (weights
    once per optimiser step, in the layout the product needs; activations per call) and
    multiplied by qarig_gemm_lp.  Returns False when the shape is not an interior one (the
    caller then runs the fp32 kernels).  The three Linear products map to
        forward   (kc, kc): NT on  x_bf (M,K)    and W_bf (N,K)
        d-input   (kc, xc): NN on  dT_bf (M,N')  and the weight shadow as stored (N',K')
        d-weight  (xc, xc): TN on  dT_bf, x_bf as they lie (row-major, reduction-major)."""
    if a_kcontig == (not b_kcontig) and not a_kcontig:
        return False                       # (xc, kc) never occurs on the hot path
    if splitk > 1 and (K % splitk or (K // splitk) % 64):
        s = splitk
        while s > 1 and (K % s or (K // s) % 64):
            s -= 1
        splitk = s
    plain = bias is None and residual is None and pre is None and gradz is None and act == 0
    if splitk > 1 and not plain:
        splitk = 1
    if not lib.qarig_gemm_lp_supported(M, N, K, splitk):
        return False
    for t in (A, B):
        if t.data_ptr() % 16 or t.stride(0) % 8:
            return False
    if a_rowsum is not None:
        if a_kcontig:
            return False
        colsum(A, out=a_rowsum, accumulate=accumulate)      # bias gradient: its own fp32 pass
    if a_kcontig and b_kcontig:            # forward
        A_bf = cast_bf16(A if A.is_contiguous() else A.contiguous())
        B_bf = cast_bf16(B if B.is_contiguous() else B.contiguous(), cache=_is_param_like(B))
        layout = 0
    elif a_kcontig:                        # d-input: B is (K_red, N_out) = the weight as stored
        if B.stride(0) % 8 or not B.is_contiguous():
            return False
        A_bf = cast_bf16(A if A.is_contiguous() else A.contiguous())
        B_bf = cast_bf16(B, cache=_is_param_like(B))
        layout = 2
    else:                                  # d-weight
        A_bf = cast_bf16(A if A.is_contiguous() else A.contiguous())
        B_bf = cast_bf16(B if B.is_contiguous() else B.contiguous())
        layout = 1
    gemm_lp(A_bf, B_bf, layout, M, N, K, C=C, bias=bias, residual=residual, preact=pre, act=act,
            gradz=gradz, gact=gact, splitk=splitk, accumulate=accumulate)
    return True


def gemm_grouped_skinny(A, W, bias=None, act=0, shared_a=False):
    """out[g] = act(A[g] @ W[g].T + bias[g]).  A: (G, M, K) or (M, K) with shared_a;
    W: (G, N, K); bias: (G, N) or None.  M <= 512, K % 256 == 0.  Returns (G, M, N)."""
    require_cuda(A, W, bias)
    G, N, K = W.shape
    assert W.is_contiguous() and A.is_contiguous() and A.dtype == W.dtype == torch.float32
    if shared_a:
        M, Ka = A.shape
        a_gs = 0
    else:
        Ga, M, Ka = A.shape
        assert Ga == G
        a_gs = M * K
    assert Ka == K
    if bias is not None:
        assert bias.shape == (G, N) and bias.is_contiguous()
    C = torch.empty((G, M, N), dtype=torch.float32, device=A.device)
    check(_lib.load().qarig_gemm_grouped_skinny_f32(
        ptr(A), K, a_gs, ptr(W), K, N * K, ptr(C), N, M * N, ptr(bias), N, G, M, N, K, act,
        stream()), "qarig_gemm_grouped_skinny_f32")
    return C


def gemm_skinny_ln(x, W, bias=None, act=0, gamma=None, beta=None, scale=None, shift=None, eps=1e-5, mul=None):
    """out[g] = act(LN(x) @ W[g].T + bias[g]) [* mul]: the decode step's Linear with the LayerNorm in
    front of it (affine gamma/beta (K,) or AdaLN scale/shift (M, K); neither: none) and the gate
    multiply behind it (mul (M, N)) inside the same launch.  x: (M, K); W: (G, N, K) or (N, K);
    bias: like W without K.  Returns (G, M, N), or (M, N) for a 2-D W."""
    require_cuda(x, W, bias, gamma, beta, scale, shift, mul)
    single = W.dim() == 2
    G, (N, K) = (1 if single else W.shape[0]), W.shape[-2:]
    M = x.shape[0]
    assert x.shape == (M, K) and x.is_contiguous() and W.is_contiguous() and x.dtype == W.dtype == torch.float32
    if bias is not None:
        assert bias.numel() == G * N and bias.is_contiguous()
    for t in (gamma, beta):
        assert t is None or (t.shape == (K,) and t.is_contiguous())
    for t in (scale, shift):
        assert t is None or (t.shape == (M, K) and t.is_contiguous())
    assert mul is None or (mul.shape == (M, N) and mul.is_contiguous() and G == 1)
    C = torch.empty((G, M, N), dtype=torch.float32, device=x.device)
    check(_lib.load().qarig_gemm_skinny_ln_f32(
        ptr(x), K, float(eps), ptr(gamma), ptr(beta), ptr(scale), ptr(shift), K, ptr(W), K, N * K, ptr(C), N,
        M * N, ptr(bias), N, ptr(mul), N, G, M, N, K, act, stream()), "qarig_gemm_skinny_ln_f32")
    return C[0] if single else C


def decode_linear_supported(M, N, K, ln=False):
    return bool(_lib.load().qarig_decode_linear_supported(int(M), int(N), int(K), int(bool(ln))))


def decode_linear(x, W, bias=None, act=0, gamma=None, beta=None, scale=None, shift=None, eps=1e-5,
                  residual=None, mul=None, out=None):
    """out[g] = act(LN(x[g]) @ W[g].T + bias[g] + residual) * mul -- the Linear of a single-token decode
    step on the weight-streaming kernel (csrc/decode.hip; M <= 16 rows).  x: (M, K) shared by the
    groups, or (G, M, K); W: (G, N, K) or (N, K); bias like W without K.  LN: gamma/beta (K,), or
    scale/shift (M, K) rows or ONE (K,) row for every activation row.  residual (M, N) (one group);
    mul (M, N) or (N,).  Returns (G, M, N), or (M, N) for a 2-D W."""
    require_cuda(x, W, bias, gamma, beta, scale, shift, residual, mul)
    single = W.dim() == 2
    G, (N, K) = (1 if single else W.shape[0]), W.shape[-2:]
    if x.dim() == 3:
        assert x.shape[0] == G
        M, x_gs = x.shape[1], x.shape[1] * K
    else:
        M, x_gs = x.shape[0], 0
    assert x.shape[-1] == K and x.is_contiguous() and W.is_contiguous() and x.dtype == W.dtype == torch.float32
    if bias is not None:
        assert bias.numel() == G * N and bias.is_contiguous()
    for t in (gamma, beta):
        assert t is None or (t.shape == (K,) and t.is_contiguous())
    ldmod = 0
    for t in (scale, shift):
        assert t is None or (t.shape in ((M, K), (K,)) and t.is_contiguous())
    if scale is not None:
        assert scale.shape == shift.shape
        ldmod = K if scale.dim() == 2 else 0
    assert residual is None or (residual.shape == (M, N) and residual.is_contiguous() and G == 1)
    ldmul = 0
    if mul is not None:
        assert mul.shape in ((M, N), (N,)) and mul.is_contiguous()
        ldmul = N if mul.dim() == 2 else 0
    C = out if out is not None else torch.empty((G, M, N), dtype=torch.float32, device=x.device)
    assert C.numel() == G * M * N and C.is_contiguous() and C.dtype == torch.float32
    check(_lib.load().qarig_decode_linear_f32(
        ptr(x), K, x_gs, float(eps), ptr(gamma), ptr(beta), ptr(scale), ptr(shift), ldmod, ptr(W), K, N * K,
        ptr(bias), N, ptr(residual), N, ptr(mul), ldmul, ptr(C), N, M * N, G, M, N, K, act, stream()),
        "qarig_decode_linear_f32")
    return C.view(M, N) if single else C.view(G, M, N)


GEMM_MAX_GROUPS = 16          # csrc/gemm.hip GEMM_MAX_GROUPS


def grouped_splitk(groups, M, N, K):
    """Reduction split of a grouped launch of `groups` (M, N, K) products: 1 when the groups alone
    fill the chip twice over, else the smallest split whose workgroup count is at least 512 and a
    (near) multiple of the 256 CUs -- co-resident workgroups share a CU's matrix pipes, so the
    launch lasts as long as its busiest CU."""
    tiles = groups * ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 512:
        return 1
    best, best_eff = 1, 0.0
    for s in (1, 2, 3, 4, 6, 8, 12, 16):
        if K % (16 * s) or K // s < 128:
            continue
        wg = tiles * s
        eff = wg / (-(-wg // 256) * 256)
        if wg >= 512 and eff >= 0.95:
            return s
        if eff > best_eff + 1e-9:
            best, best_eff = s, eff
    return best


def gemm_grouped_supported(M, N, K, splitk=1, any_precision=False):
    """any_precision: callers whose products stay on the fp32 grouped kernel in the reduced-precision mode
    too (the conditioning projections of a position table: a few thousand rows, where 63 separate bf16
    nodes -- a cast, a product, two gradient products, a reduce and an accumulation add each -- cost twice
    what the twelve fp32 grouped launches do)."""
    return (any_precision or not lp_mode()) and bool(_lib.load().qarig_gemm_grouped_supported(M, N, K, splitk))


def _ptr_table(tensors, G):
    if tensors is None:
        return None
    assert len(tensors) == G
    return (_lib.P * G)(*[t.data_ptr() if t is not None else None for t in tensors])


def gemm_grouped(A, B, C, M, N, K, a_kcontig=True, b_kcontig=True, bias=None, residual=None, preact=None,
                 act=0, gradz=None, gact=0, splitk=1, accumulate=False, sum_groups=False, a_rowsum=None):
    """`len(A)` products of one shape in one launch (include/qarig.h qarig_gemm_f32_grouped): lists
    of fp32 tensors per group (A / B entries may repeat), outputs in the tensors of C (one tensor
    when sum_groups).  Operands must be dense row-major with a common row stride."""
    G = len(A)
    assert 1 <= G <= GEMM_MAX_GROUPS and len(B) == G
    require_cuda(*A, *B, *C)
    lda, ldb, ldc = A[0].stride(0), B[0].stride(0), C[0].stride(0)
    for t in (*A, *B, *C, *(bias or ()), *(residual or ()), *(preact or ()), *(gradz or ()), *(a_rowsum or ())):
        assert t.dtype == torch.float32 and t.stride(-1) == 1
    assert all(t.stride(0) == lda for t in A) and all(t.stride(0) == ldb for t in B)
    assert all(t.stride(0) == ldc and t.shape == (M, N) for t in C) and len(C) == (1 if sum_groups else G)
    ld = lambda ts: ts[0].stride(0) if ts else 0   # noqa: E731
    for ts in (residual, preact, gradz):
        if ts:
            assert all(t.stride(0) == ts[0].stride(0) and t.shape == (M, N) for t in ts)
    lib = _lib.load()
    ws, nws = None, 0
    if splitk > 1 or sum_groups or a_rowsum is not None:
        nws = lib.qarig_gemm_grouped_workspace_bytes(G, M, N, splitk, int(sum_groups))
        ws = workspace(nws, A[0].device, "gemm")
        nws = ws.numel()
    if GEMM_EVENTS is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    Ct = list(C) + [C[0]] * (G - len(C))
    check(lib.qarig_gemm_f32_grouped(
        G, _ptr_table(A, G), lda, int(a_kcontig), _ptr_table(B, G), ldb, int(b_kcontig), _ptr_table(Ct, G), ldc,
        M, N, K, _ptr_table(bias, G), _ptr_table(residual, G), ld(residual), _ptr_table(preact, G), ld(preact),
        act, _ptr_table(gradz, G), ld(gradz), gact, splitk, int(accumulate), int(sum_groups),
        _ptr_table(a_rowsum, G), ptr(ws), nws, stream()), "qarig_gemm_f32_grouped")
    if GEMM_EVENTS is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        GEMM_EVENTS.append((2.0 * G * M * N * K, ev0, ev1, "f32g" if lp_mode() else ""))


def colsum(X, out=None, accumulate=False):
    """(N,) column sums of X (M,N) in a fixed order (optionally added into `out`)."""
    require_cuda(X)
    assert X.dim() == 2 and X.stride(1) == 1 and X.dtype == torch.float32
    M, N = X.shape
    if out is None:
        assert not accumulate
        out = torch.empty(N, dtype=torch.float32, device=X.device)
    lib = _lib.load()
    nb = lib.qarig_colsum_workspace_bytes(M, N)
    ws = workspace(nb, X.device, "colsum")
    check(lib.qarig_colsum_f32(ptr(X), X.stride(0), M, N, ptr(out), int(accumulate), ptr(ws),
                               ws.numel(), stream()), "qarig_colsum_f32")
    return out


def auto_splitk(M, N, K):
    """GEMMs with fewer output tiles than CUs (the M = batch*beams rows of autoregressive
    decode, small per-GPU batches) spread their reduction over the chip; the epilogue then
    runs in the reduce pass."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 192 or K < 256 or K % 64:
        return 1
    s64 = _tile64_splitk(M, N, K)
    if s64:
        return s64
    if tiles < 64:
        return max(1, min(256 // tiles, K // 64))
    return max(1, min(512 // tiles, K // 128))


def _tile64_splitk(M, N, K):
    """Reduction split of a product the library runs on 64 x 64 tiles (qarig_gemm_tile64): none when the tiles
    fill the chip, else the smallest split of whole 16-deep k-tiles, at least 128 deep, that gives >= 256
    workgroups.  0: not a 64-tile shape."""
    if not _lib.load().qarig_gemm_tile64(int(M), int(N), int(K)):
        return 0
    tiles = (M // 64) * (N // 64)
    s = 1
    while tiles * s < 256 and K % (2 * s) == 0 and (K // (2 * s)) % 16 == 0 and K // (2 * s) >= 128:
        s *= 2
    return s


def pick_splitk(M, N, K):
    """Reduction split for weight-gradient shaped GEMMs (small M x N, long K)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 256:
        return 1
    s64 = _tile64_splitk(M, N, K)
    if s64:
        return s64
    if K < 2048:
        # short reductions over few tiles (weight gradients of the position-table layers,
        # K = a few hundred positions): a handful of workgroups would each walk the whole
        # reduction at DMA latency; spread it in 64-deep slices that divide K exactly
        if tiles > 64 or K % 64:
            return 1
        s = K // 64
        while s > 1 and (K % s or (K // s) % 16 or s * tiles > 512):
            s -= 1
        return max(1, s)
    # (a 512 x 512 gradient over 2,048 rows -- 16 tiles -- measured 40 us in 4 slices of 512 rows and
    # 29 us in 8 of 256: with few tiles the slices may be as short as 16 k-tiles)
    s = max(1, min(max(1, 512 // tiles), K // (256 if tiles <= 32 else 512), 32))
    # slices that divide K into whole 16-deep tiles keep the launch on the interior kernels
    # (the padded classifier's 640 x 2048 output over K = 16384 asked for 6: 2736-deep slices
    # sent it to the guarded kernel at half the rate)
    while s > 1 and (K % s or (K // s) % 16):
        s -= 1
    return s


# ------------------------------------------------------------ shared small helpers
_flag_cache = {}


def _bad_flag(device):
    """Device int the kernels set when they meet an out-of-range index."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    f = _flag_cache.get(key)
    if f is None:
        f = torch.zeros(1, dtype=torch.int32, device=device)
        _flag_cache[key] = f
    return f


def check_index_flag(device, what):
    """Raises IndexError (what torch's embedding / nll_loss would do) if a kernel
    flagged an out-of-range index.  Synchronises; callers use it where the reference
    already synchronises (loss.item()) or in tests/debug mode."""
    f = _bad_flag(device)
    if int(f.item()) != 0:
        f.zero_()
        raise IndexError(f"{what}: index out of range")


def _i64c(t):
    if t.dtype != torch.int64:
        t = t.long()
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------- codebook
def patchify(image, patch_dim):
    """(N,C,H,W) -> (N,Seq,D).  reference models/layers.py:8-34"""
    require_cuda(image)
    image = f32c(image)
    N, C, H, W = image.shape
    pH, pW = patch_dim
    if H % pH or W % pW:  # the reference truncates silently (`//`)
        image = image[:, :, :H // pH * pH, :W // pW * pW].contiguous()
        N, C, H, W = image.shape
    out = torch.empty((N, (H // pH) * (W // pW), C * pH * pW), dtype=torch.float32,
                      device=image.device)
    check(_lib.load().qarig_patchify_fwd(ptr(image), N, C, H, W, pH, pW, ptr(out), stream()),
          "qarig_patchify_fwd")
    return out


def unpatchify(patches, image_dim, patch_dim):
    """(N,Seq,D) -> (N,C,H,W).  reference models/layers.py:37-71"""
    require_cuda(patches)
    patches = f32c(patches)
    H, W = image_dim
    pH, pW = patch_dim
    N, Seq, D = patches.shape
    C = D // (pH * pW)
    assert Seq == (H // pH) * (W // pW) and H % pH == 0 and W % pW == 0
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=patches.device)
    check(_lib.load().qarig_unpatchify_fwd(ptr(patches), N, C, H, W, pH, pW, ptr(out), stream()),
          "qarig_unpatchify_fwd")
    return out


def codebook_gather_image(ids, codebook, image_dim, patch_dim):
    """unpatchify(codebook[ids]).  reference models/Codebook.py:138-154"""
    require_cuda(ids, codebook)
    ids = _i64c(ids)
    codebook = f32c(codebook)
    N, Seq = ids.shape
    H, W = image_dim
    pH, pW = patch_dim
    K, D = codebook.shape
    C = D // (pH * pW)
    assert Seq == (H // pH) * (W // pW)
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=ids.device)
    check(_lib.load().qarig_codebook_gather_image(ptr(ids), N, C, H, W, pH, pW, ptr(codebook), K,
                                                  ptr(out), ptr(_bad_flag(ids.device)), stream()),
          "qarig_codebook_gather_image")
    return out


def gather_rows(ids, table):
    require_cuda(ids, table)
    ids = _i64c(ids).reshape(-1)
    table = f32c(table)
    K, D = table.shape
    out = torch.empty((ids.numel(), D), dtype=torch.float32, device=table.device)
    check(_lib.load().qarig_gather_rows(ptr(ids), ids.numel(), D, K, ptr(table), ptr(out),
                                        ptr(_bad_flag(table.device)), stream()), "qarig_gather_rows")
    return out


def som_weights(bmu_idx, K, two_var):
    require_cuda(bmu_idx)
    bmu_idx = _i64c(bmu_idx).reshape(-1)
    g = torch.empty((bmu_idx.numel(), K), dtype=torch.float32, device=bmu_idx.device)
    check(_lib.load().qarig_som_weights_fwd(ptr(bmu_idx), bmu_idx.numel(), K, float(two_var), ptr(g),
                                            stream()), "qarig_som_weights_fwd")
    return g


def som_reach(two_var):
    """Index distance beyond which exp(-d^2 / two_var) < 2^-40: what som_band drops."""
    import math
    return int(math.ceil(math.sqrt(40.0 * math.log(2.0) * float(two_var))))


def som_band(table, two_var, reach=None):
    """out[j] = sum_{|j-b| <= reach} exp(-(j-b)^2 / two_var) table[b] (reference Codebook.py:112-130
    folded onto the code axis: see csrc/codebook.hip som_band_kernel)."""
    require_cuda(table)
    table = f32c(table)
    K, D = table.shape
    out = torch.empty_like(table)
    reach = som_reach(two_var) if reach is None else int(reach)
    check(_lib.load().qarig_som_band(ptr(table), K, D, float(two_var), min(reach, K), ptr(out), stream()),
          "qarig_som_band")
    return out


# ------------------------------------------------------------------- transformer
_freq_cache = {}


def pos_frequencies(D, device):
    """exp(arange(D/2) * -ln(1e4)/(D/2-1)) computed on the host with the same torch
    ops as the reference (layers.py:84-91), cached per (D, device)."""
    import math
    key = (D, str(device))
    f = _freq_cache.get(key)
    if f is None:
        half = D // 2
        c = math.log(10_000) / (half - 1)
        f = torch.exp(torch.arange(half, dtype=torch.float32) * -c).to(device)
        _freq_cache[key] = f
    return f


def posemb(pos, D):
    """get_positional_embeddings(D, pos) -> (len(pos), D).  pos: int or float tensor."""
    require_cuda(pos)
    pos = pos.reshape(-1).to(torch.float32).contiguous()
    out = torch.empty((pos.numel(), D), dtype=torch.float32, device=pos.device)
    check(_lib.load().qarig_posemb_fwd(ptr(pos), pos.numel(), D, ptr(pos_frequencies(D, pos.device)),
                                       ptr(out), stream()), "qarig_posemb_fwd")
    return out


def assemble_tokens(lr_idx, hr_idx, base, k_lr, k_hr, offs=None, window=None):
    """(hr_in, hr_tg, pos) int64 (N,W) from the BMU indices: token assembly + window slicing of
    the training loop (reference train_quantized_transformer.py:423-484) in one launch.
    offs: int64 (N,) window starts on the device, or None for the whole sequence (pos None)."""
    require_cuda(hr_idx, lr_idx, offs)
    hr_idx = _i64c(hr_idx)
    N, S_hr = hr_idx.shape
    S_lr = 0
    if base:
        lr_idx = _i64c(lr_idx)
        S_lr = lr_idx.shape[1]
    s_in = (S_lr if base else 1) + S_hr
    W = s_in if offs is None else int(window)
    dev = hr_idx.device
    hr_in = torch.empty((N, W), dtype=torch.int64, device=dev)
    hr_tg = torch.empty((N, W), dtype=torch.int64, device=dev)
    pos = torch.empty((N, W), dtype=torch.int64, device=dev) if offs is not None else None
    if offs is not None:
        offs = _i64c(offs)
        assert offs.shape == (N,)
    check(_lib.load().qarig_assemble_tokens(ptr(lr_idx) if base else None, S_lr, ptr(hr_idx), S_hr, N,
                                            int(base), int(k_lr), int(k_hr), ptr(offs), W, ptr(hr_in),
                                            ptr(hr_tg), ptr(pos), ptr(_bad_flag(dev)), stream()),
          "qarig_assemble_tokens")
    return hr_in, hr_tg, pos


def embedding_fwd(ids, table, pe=None):
    """table[ids] (+ pe[s]) -> (N,S,D)."""
    require_cuda(ids, table, pe)
    ids = _i64c(ids)
    N, S = ids.shape
    V, D = table.shape
    out = torch.empty((N, S, D), dtype=torch.float32, device=table.device)
    check(_lib.load().qarig_embedding_fwd(ptr(ids), N * S, S, D, V, ptr(table), ptr(pe), ptr(out),
                                          ptr(_bad_flag(table.device)), stream()),
          "qarig_embedding_fwd")
    return out


def embedding_bwd(ids, dy, V):
    ids = _i64c(ids).reshape(-1)
    dy = f32c(dy).reshape(ids.numel(), -1)
    D = dy.shape[1]
    out = torch.empty((V, D), dtype=torch.float32, device=dy.device)
    check(_lib.load().qarig_embedding_bwd(ptr(ids), ids.numel(), D, V, ptr(dy), ptr(out), stream()),
          "qarig_embedding_bwd")
    return out


def layernorm_fwd(x2d, gamma=None, beta=None, scale=None, shift=None, eps=1e-5, mod_idx=None):
    """mod_idx (int32 (M,)): scale/shift are (P,D) position tables read at row mod_idx[token]."""
    M, D = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(M, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x2d.device)
    if mod_idx is not None:
        assert mod_idx.dtype == torch.int32 and mod_idx.shape == (M,) and mod_idx.is_contiguous()
    check(_lib.load().qarig_layernorm_fwd(ptr(x2d), M, D, eps, ptr(gamma), ptr(beta), ptr(scale),
                                          ptr(shift), ptr(mod_idx), ptr(y), ptr(mean), ptr(rstd),
                                          stream()), "qarig_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x2d, mean, rstd, gamma=None, scale=None, want_dy_xhat=False, mod_idx=None,
                  dx_add=None):
    """dx_add (M,D): the skip connection's gradient of the same x, added into dx by the kernel."""
    M, D = x2d.shape
    dx = torch.empty_like(x2d)
    dyx = torch.empty_like(x2d) if want_dy_xhat else None
    if dx_add is not None:
        assert dx_add.shape == x2d.shape and dx_add.dtype == torch.float32 and dx_add.is_contiguous()
    check(_lib.load().qarig_layernorm_bwd(ptr(dy), ptr(x2d), ptr(mean), ptr(rstd), ptr(gamma),
                                          ptr(scale), ptr(mod_idx), M, D, ptr(dx), ptr(dyx), ptr(dx_add),
                                          stream()), "qarig_layernorm_bwd")
    return dx, dyx


def rowmap_build(idx, P):
    """(offsets (P+1,), rows (M,)) int32: rows[offsets[p]:offsets[p+1]] = ascending tokens with
    idx == p.  Out-of-range indices set the device index flag (see check_index_flag)."""
    require_cuda(idx)
    assert idx.dtype == torch.int32 and idx.dim() == 1 and idx.is_contiguous()
    M = idx.numel()
    counts = torch.empty(P, dtype=torch.int32, device=idx.device)
    offsets = torch.empty(P + 1, dtype=torch.int32, device=idx.device)
    rows = torch.empty(M, dtype=torch.int32, device=idx.device)
    check(_lib.load().qarig_rowmap_build(ptr(idx), M, P, ptr(counts), ptr(offsets), ptr(rows),
                                         ptr(_bad_flag(idx.device)), stream()), "qarig_rowmap_build")
    return offsets, rows


def segment_sum(src, offsets, rows):
    """(P,D) sums of the rows of src (M,D) that the row map assigns to each table row."""
    P = offsets.numel() - 1
    M, D = src.shape
    assert src.is_contiguous() and rows.numel() == M
    out = torch.empty((P, D), dtype=torch.float32, device=src.device)
    check(_lib.load().qarig_segment_sum(ptr(src), ptr(offsets), ptr(rows), P, D, ptr(out), stream()),
          "qarig_segment_sum")
    return out


def mul_rows_fwd(a, tab, idx):
    M, D = a.shape
    y = torch.empty_like(a)
    check(_lib.load().qarig_mul_rows_fwd(ptr(a), ptr(tab), ptr(idx), ptr(y), M, D, stream()),
          "qarig_mul_rows_fwd")
    return y


def mul_rows_bwd(dy, a, tab, idx):
    M, D = a.shape
    da = torch.empty_like(a)
    db = torch.empty_like(a)
    check(_lib.load().qarig_mul_rows_bwd(ptr(dy), ptr(a), ptr(tab), ptr(idx), ptr(da), ptr(db), M, D,
                                         stream()), "qarig_mul_rows_bwd")
    return da, db


ATTENTION_HEAD_DIMS = (4, 8, 16, 32, 64, 128)   # csrc/attention.hip instantiations; 128: csrc/attention_wide.hip
DECODE_HEAD_DIMS = (4, 8, 16, 32, 64)           # csrc/decode.hip decode_attention_kernel (key/value cache)


def attention_head_dim(d):
    """The kernel head dim that serves a model head dim d: d itself, or the next instantiated one (the
    caller zero-pads every head: extra zero columns add nothing to q.k and yield zero output columns),
    or None above the widest."""
    for hd in ATTENTION_HEAD_DIMS:
        if hd >= d:
            return hd
    return None


def attention_fwd(q, k, v, heads, causal, scale_dim=None):
    """scale_dim: the model's head dim when the tensors carry zero-padded heads (softmax scale 1/sqrt(scale_dim))."""
    N, Sq, D = q.shape
    Sk = k.shape[1]
    d = D // heads
    o = torch.empty_like(q)
    lse = torch.empty((N, heads, Sq), dtype=torch.float32, device=q.device)
    lib = _lib.load()
    fn = lib.qarig_attention_lp_fwd if lp_mode() else lib.qarig_attention_fwd
    check(fn(ptr(q), ptr(k), ptr(v), N, Sq, Sk, heads, d, int(causal), float((scale_dim or d) ** 0.5), ptr(o),
             ptr(lse), stream()), "qarig_attention_fwd")
    return o, lse


def attention_decode(q, k_new, v_new, kcache, vcache, length, heads, len_dev=None, o_mul=None):
    """One decode step: q (B,D) against cache rows [0,length) (+ the appended new row).
    kcache/vcache: row-major (B, max_len, D) views with unit row stride D (batch stride free), or
    head-major (B, H, max_len, d) -- DecodeCache's layout.  o_mul: (B, D), or one (D,) row for every sequence."""
    B, D = q.shape
    d = D // heads
    assert kcache.shape == vcache.shape and kcache.shape[0] == B and vcache.stride() == kcache.stride()
    if kcache.dim() == 4:
        assert kcache.shape[1] == heads and kcache.shape[3] == d and kcache.stride(3) == 1 and kcache.stride(2) == d
        max_len, hstride, rstride = kcache.shape[2], kcache.stride(1), d
    else:
        assert kcache.shape[2] == D and kcache.stride(2) == 1 and kcache.stride(1) == D
        max_len, hstride, rstride = kcache.shape[1], d, D
    assert q.is_contiguous()
    o = torch.empty_like(q)
    ldmul = 0
    if o_mul is not None:
        assert o_mul.shape in (q.shape, (D,)) and o_mul.is_contiguous()
        ldmul = D if o_mul.dim() == 2 else 0
    check(_lib.load().qarig_decode_attention(
        ptr(q), ptr(k_new) if k_new is not None else None,
        ptr(v_new) if v_new is not None else None, ptr(kcache), ptr(vcache), B, heads, d,
        int(length), ptr(len_dev) if len_dev is not None else None, max_len,
        kcache.stride(0), hstride, rstride, float(d ** 0.5), ptr(o_mul), ldmul, ptr(o), stream()),
        "qarig_decode_attention")
    return o


DECODE_CTL_WORDS = 8          # include/qarig.h QARIG_DECODE_CTL_WORDS: [0] len, [1] chunk start, [2] draws, [3] candidate


def decode_embed(ids, table, pe, ctl=None, length=0, proj_table=None, proj_row=None):
    """x[b] = table[ids[b]] + pe[len] (len = ctl[0], or `length` without ctl) and proj_row <- row len of
    proj_table (max_len, PD), the stage's per-position projections of `cond`.  Returns x (B, D)."""
    require_cuda(ids, table, pe, ctl, proj_table, proj_row)
    ids = _i64c(ids).reshape(-1)
    B, (V, D) = ids.numel(), table.shape
    assert table.is_contiguous() and table.dtype == torch.float32
    max_len = pe.shape[0] if pe is not None else (proj_table.shape[0] if proj_table is not None else int(length) + 1)
    assert pe is None or (pe.shape == (max_len, D) and pe.is_contiguous())
    pd = 0
    if proj_table is not None:
        assert proj_table.dim() == 2 and proj_table.shape[0] == max_len and proj_table.is_contiguous()
        pd = proj_table.shape[1]
        assert proj_row is not None and proj_row.numel() == pd and proj_row.is_contiguous()
    assert ctl is None or (ctl.dtype == torch.int32 and ctl.numel() >= DECODE_CTL_WORDS)
    x = torch.empty((B, D), dtype=torch.float32, device=table.device)
    check(_lib.load().qarig_decode_embed(ptr(ids), B, D, V, ptr(table), ptr(pe), ptr(ctl), int(length), max_len,
                                         ptr(proj_table), pd, ptr(x), ptr(proj_row), ptr(_bad_flag(table.device)),
                                         stream()), "qarig_decode_embed")
    return x


def decode_sample(logits, temperature, end_token, generate_mode, shift, uniforms, ctl, slot, beam_width, ids,
                  chunk, comb, forced=None, probs_log=None, inc_len=False, beams=0):
    """One draw per row of logits (B, V): see include/qarig.h qarig_decode_sample.  beams > 0: candidate-major
    draw numbering, uniforms / forced / probs_log have B // beams columns."""
    require_cuda(logits, uniforms, ctl, ids, chunk, comb, forced, probs_log)
    B, V = logits.shape
    max_draws = uniforms.shape[0]
    cols = B // beams if beams else B
    assert logits.stride(1) == 1 and uniforms.shape == (max_draws, cols) and uniforms.is_contiguous()
    assert ids.dtype == chunk.dtype == torch.int64 and ids.numel() == B and chunk.shape == (B, beam_width)
    assert comb.shape == (B,) and comb.dtype == torch.float32 and ctl.dtype == torch.int32
    assert forced is None or (forced.shape == (max_draws, cols) and forced.dtype == torch.int64 and forced.is_contiguous())
    assert probs_log is None or (probs_log.shape == (max_draws, cols, V) and probs_log.is_contiguous())
    check(_lib.load().qarig_decode_sample(ptr(logits), logits.stride(0), B, V, float(temperature), int(end_token),
                                          int(bool(generate_mode)), int(shift), ptr(uniforms), ptr(forced), ptr(ctl),
                                          int(slot), int(beam_width), max_draws, int(bool(inc_len)), int(beams),
                                          ptr(ids), ptr(chunk), ptr(comb), ptr(probs_log), stream()),
          "qarig_decode_sample")


def decode_decide(ctl, N, NB, beam_width, comb, chunk, best_p, best_chunk, take, draws=None):
    """draws: draw rows the candidate set consumed (default beam_width: one candidate chunk per row)."""
    require_cuda(ctl, comb, chunk, best_p, best_chunk, take)
    assert comb.numel() == N * NB and chunk.shape == (N * NB, beam_width) and best_chunk.shape == (N, beam_width)
    assert best_p.numel() == N and take.numel() == N and take.dtype == torch.int32
    check(_lib.load().qarig_decode_decide(ptr(ctl), N, NB, beam_width, int(beam_width if draws is None else draws),
                                          ptr(comb), ptr(chunk), ptr(best_p), ptr(best_chunk), ptr(take), stream()),
          "qarig_decode_decide")


def decode_rows(ctl, kv, staged, take, N, NB, restore):
    """kv (layers, 2, N * NB, H, max_len, d) <-> staged (layers, 2, N, H, R, d) at rows [ctl[1], ctl[1] + R)."""
    require_cuda(ctl, kv, staged, take)
    layers, two, B, H, max_len, d = kv.shape
    R = staged.shape[4]
    assert two == 2 and B == N * NB and staged.shape == (layers, 2, N, H, R, d)
    assert kv.is_contiguous() and staged.is_contiguous()
    check(_lib.load().qarig_decode_rows(ptr(ctl), ptr(kv), ptr(staged), ptr(take), layers * 2, N, NB, H, R, d,
                                        max_len, int(bool(restore)), stream()), "qarig_decode_rows")


def decode_commit(ctl, N, NB, beam_width, best_chunk, tokens, ids):
    require_cuda(ctl, best_chunk, tokens, ids)
    assert tokens.dtype == torch.int64 and tokens.dim() == 2 and tokens.shape[0] == N and tokens.stride(1) == 1
    assert ids.numel() == N * NB and best_chunk.shape == (N, beam_width)
    check(_lib.load().qarig_decode_commit(ptr(ctl), N, NB, beam_width, ptr(best_chunk), ptr(tokens),
                                          tokens.stride(0), ptr(ids), stream()), "qarig_decode_commit")


def decode_advance(ctl, beam_width):
    require_cuda(ctl)
    check(_lib.load().qarig_decode_advance(ptr(ctl), int(beam_width), stream()), "qarig_decode_advance")


def attention_bwd(q, k, v, o, dO, lse, heads, causal, scale_dim=None):
    N, Sq, D = q.shape
    Sk = k.shape[1]
    d = D // heads
    dq = torch.empty_like(q)
    dk = torch.empty_like(k)
    dv = torch.empty_like(v)
    delta = torch.empty_like(lse)
    lib = _lib.load()
    fn = lib.qarig_attention_lp_bwd if lp_mode() else lib.qarig_attention_bwd
    check(fn(ptr(q), ptr(k), ptr(v), ptr(o), ptr(dO), ptr(lse), N, Sq, Sk, heads, d, int(causal),
             float((scale_dim or d) ** 0.5), ptr(dq), ptr(dk), ptr(dv), ptr(delta), stream()), "qarig_attention_bwd")
    return dq, dk, dv


def cross_entropy_fwd(logits2d, target, want_grad=True):
    M, C = logits2d.shape
    loss = torch.empty((), dtype=torch.float32, device=logits2d.device)
    dl = torch.empty_like(logits2d) if want_grad else None
    rows = torch.empty(M, dtype=torch.float32, device=logits2d.device)
    check(_lib.load().qarig_cross_entropy_fwd(ptr(logits2d), ptr(target), M, C, ptr(loss), ptr(dl),
                                              ptr(rows), ptr(_bad_flag(logits2d.device)), stream()),
          "qarig_cross_entropy_fwd")
    return loss, dl


def adam_step(p, g, m, v, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale=1.0, dev_step=None, shadow=None):
    """shadow: optional bf16 tensor of p's numel that receives the updated parameters (reduced precision)."""
    assert shadow is None or (shadow.dtype == torch.bfloat16 and shadow.numel() == p.numel())
    check(_lib.load().qarig_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), beta1, beta2, eps,
                                      step_size, bc2_sqrt, grad_scale, ptr(dev_step), ptr(shadow), stream()),
          "qarig_adam_step")


def mul_fwd(a, b):
    y = torch.empty_like(a)
    check(_lib.load().qarig_mul_fwd(ptr(a), ptr(b), ptr(y), a.numel(), stream()), "qarig_mul_fwd")
    return y


def mul_bwd(dy, a, b):
    da = torch.empty_like(a)
    db = torch.empty_like(a)
    check(_lib.load().qarig_mul_bwd(ptr(dy), ptr(a), ptr(b), ptr(da), ptr(db), a.numel(), stream()),
          "qarig_mul_bwd")
    return da, db


def act_fwd(x, act):
    y = torch.empty_like(x)
    check(_lib.load().qarig_act_fwd(ptr(x), ptr(y), x.numel(), act, stream()), "qarig_act_fwd")
    return y


def act_bwd(dy, z, act):
    dz = torch.empty_like(z)
    check(_lib.load().qarig_act_bwd(ptr(dy), ptr(z), ptr(dz), z.numel(), act, stream()),
          "qarig_act_bwd")
    return dz


def scale_by(x, s):
    y = torch.empty_like(x)
    check(_lib.load().qarig_scale_by(ptr(x), ptr(s), ptr(y), x.numel(), stream()), "qarig_scale_by")
    return y


# -------------------------------------------------------------------------- conv
def _conv_workspace(weight, nbytes, geom, tag, inference):
    """(scratch tensor, flags) of a conv forward.  Training (`inference` False -- the caller decides, outside
    its autograd node: torch.is_grad_enabled() is always False inside an autograd.Function.forward, and
    functional.conv2d_act samples it before .apply): the shared scratch, weights re-ordered in every call.  Inference (the
    weights are constants between optimiser steps): one scratch per weight and geometry, kept while the
    weight is unchanged (address / version / LP_EPOCH, as the bf16 shadows), so the re-ordering launch runs
    once (QARIG_CONV_PACKED_VALID).  Inside a stream capture always the shared scratch: a captured launch
    must not hold the address of a cache entry that an invalidation can free."""
    if not inference or torch.cuda.is_current_stream_capturing():
        return workspace(nbytes, weight.device, tag), 0
    key = (tag, weight.data_ptr(), tuple(weight.shape), weight._version, LP_EPOCH, _lib.OPTION_EPOCH, geom)
    hit = _lp_get(key, weight)
    if hit is not None:
        return hit, 1
    ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device=weight.device)
    _lp_put(key, weight, ws)
    return ws, 0


def conv2d_fwd(x, weight, bias, stride, pad, act, want_preact=False, inference=False):
    """nn.Conv2d + bias + activation (reference models/layers.py:157-184, 211-230).
    inference: no gradient will be asked of this call (cached re-ordered weights, _conv_workspace)."""
    require_cuda(x, weight, bias)
    x = f32c(x)
    N, Cin, H, W = x.shape
    Cout, Cin2, k, k2 = weight.shape
    assert Cin == Cin2 and k == k2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    pre = torch.empty_like(y) if want_preact else None
    lib = _lib.load()
    ws, flags = _conv_workspace(weight, lib.qarig_conv2d_fwd_workspace_bytes_n(N, Cin, H, W, Cout, k, stride),
                                (N, H, W, stride, pad, x.data_ptr() % 16), "convfwd", inference)   # (alignment picks the kernel family)
    check(lib.qarig_conv2d_fwd_ws(ptr(x), N, Cin, H, W, ptr(weight), ptr(bias), Cout, k, stride,
                                  pad, act, ptr(y), ptr(pre), ptr(ws), ws.numel(), flags, stream()),
          "qarig_conv2d_fwd_ws")
    return (y, pre) if want_preact else y


def conv_transpose2d_fwd(x, weight, bias, act, want_preact=False, inference=False):
    """nn.ConvTranspose2d(4, 2, 1) + bias + activation (reference layers.py:188-207)."""
    require_cuda(x, weight, bias)
    x = f32c(x)
    N, Cin, H, W = x.shape
    Cin2, Cout, k, k2 = weight.shape
    assert Cin == Cin2 and k == 4 and k2 == 4
    y = torch.empty((N, Cout, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    pre = torch.empty_like(y) if want_preact else None
    lib = _lib.load()
    ws, flags = _conv_workspace(weight, lib.qarig_conv_transpose2d_workspace_bytes_n(N, Cin, H, W, Cout),
                                (N, H, W, bool(want_preact), x.data_ptr() % 16), "convt", inference)
    check(lib.qarig_conv_transpose2d_fwd(ptr(x), N, Cin, H, W, ptr(weight), ptr(bias), Cout, act,
                                         ptr(y), ptr(pre), ptr(ws), ws.numel(), flags, stream()),
          "qarig_conv_transpose2d_fwd")
    return (y, pre) if want_preact else y


def mse_fwd(pred, target, want_grad=True):
    """mean((pred-target)^2) and its gradient w.r.t. pred."""
    require_cuda(pred, target)
    assert pred.shape == target.shape
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    dp = torch.empty_like(pred) if want_grad else None
    lib = _lib.load()
    part = torch.empty(lib.qarig_mse_workspace_bytes() // 4, dtype=torch.float32, device=pred.device)
    check(lib.qarig_mse_fwd(ptr(pred), ptr(target), pred.numel(), ptr(loss), ptr(dp), ptr(part),
                            stream()), "qarig_mse_fwd")
    return loss, dp


def index_histogram(ids, counts):
    """counts (int64 [K], on the device) += histogram of ids."""
    require_cuda(ids, counts)
    ids = _i64c(ids).reshape(-1)
    assert counts.dtype == torch.int64 and counts.is_contiguous()
    check(_lib.load().qarig_index_histogram(ptr(ids), ids.numel(), counts.numel(), ptr(counts),
                                            ptr(_bad_flag(ids.device)), stream()),
          "qarig_index_histogram")
    return counts
