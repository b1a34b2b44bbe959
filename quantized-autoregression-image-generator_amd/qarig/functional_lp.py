"""Reduced-precision (bf16) forms of the Linear building blocks -- BASELINE config 5's
"bf16 MFMA attn/FFN" mode; opt-in (qarig.ops.set_precision("bf16")), never the fp32 parity path.

What changes against qarig.functional's fp32 nodes (same math, same parameters, same gradients
up to bf16 rounding of the GEMM operands):
 * every GEMM runs on csrc/gemm_lp.hip with bf16 operands IN HBM;
 * tensors that only GEMMs read are STORED in bf16 and never exist in fp32: the MLP hidden
   activation h, its pre-activation t1 (read back only for act'), and the hidden gradient dT1 --
   the 2048-wide tensors that dominate the step's HBM traffic.  They are written by the producing
   GEMM's epilogue (Cb / Pb outputs), not by a cast pass;
 * the narrow (512-wide) GEMM inputs that other kernels produce in fp32 (LayerNorm output,
   attention output, incoming gradients) are cast once per node; the cast of an incoming gradient
   also yields the bias gradient (column sums ride on the same pass: qarig_cast_colsum);
 * weights keep fp32 masters; ONE bf16 shadow per weight, as stored (N,K): a view of the flat bf16 image the
   Adam kernel writes beside the fp32 update (optim.FlatAdam.flat_shadow; before the first step, and for weights
   no FlatAdam owns, a cast cached until the next optimiser step): the forward reads it reduction-contiguous,
   the input gradient reduction-major (transposed on the LDS read), so no W^T copy exists;
 * weight gradients are TN products of the row-major bf16 activations as they lie (transposed on
   the LDS read), accumulated in fp32 straight into the parameter's .grad.
"""
import torch

from . import ops
from ._lib import f32c, require_cuda
from .functional import _2d, _grad_slot, _report_done


def _pad128(n):
    return (n + 127) // 128 * 128


def _shadow(w, transpose=False, pad_rows=0):
    """bf16 shadow of a weight (N,K): as stored, or transposed (K,N); optionally zero-padded to
    pad_rows output rows (ragged classifier widths).  Cached until the next optimiser step."""
    if pad_rows and pad_rows != w.shape[0]:
        key = ("p", w.data_ptr(), tuple(w.shape), pad_rows, transpose, w._version, ops.LP_EPOCH)
        hit = None if torch.cuda.is_current_stream_capturing() else ops._lp_get(key, w)
        if hit is not None:
            return hit
        wp = torch.zeros((pad_rows, w.shape[1]), dtype=torch.float32, device=w.device)
        wp[:w.shape[0]].copy_(w.detach())
        out = ops.cast_transpose_bf16(wp) if transpose else ops.cast_bf16(wp)
        if not torch.cuda.is_current_stream_capturing():
            ops._lp_put(key, w, out)
        return out
    if not transpose:
        owner = getattr(w, "_qarig_owner", None)
        if owner is not None and hasattr(owner, "shadow_of"):
            v = owner.shadow_of(w)           # written by the optimiser's own pass (optim.FlatAdam)
            if v is not None:
                return v
    wd = w.detach()
    wd._qarig_weight = True
    wd._qarig_src = w            # the cache entry is tied to the parameter, not to this temporary
    return ops.cast_transpose_bf16(wd, cache=True) if transpose else ops.cast_bf16(wd, cache=True)


def _cast_in(x2, shapes):
    """bf16 copy of a node's fp32 input (backward reads it) and, in "fp8" mode when every forward
    product of the node that reads x ((M, N, K) in `shapes`) fits the e4m3 kernel, its per-tensor
    e4m3 quantisation from the same pass: (xb, (x8, inv_scale) or None)."""
    if ops.PRECISION == "fp8" and all(ops.f8_supported(*s) for s in shapes):
        x8, sx, xb = ops.cast_fp8(x2, want_bf16=True)
        return xb, (x8, sx)
    return ops.cast_bf16(x2), None


def _fwd_nt(xq, xb, w, M, N, K, **epi):
    """The forward product x W^T of a node: bf16 operands, or -- with xq from _cast_in -- x and W
    quantised per tensor (W once per optimiser step) and multiplied on the fp8 MFMA.  Backward is
    the bf16 one either way."""
    if xq is not None:
        wd = w.detach()
        wd._qarig_weight = True
        wd._qarig_src = w
        w8, sw = ops.cast_fp8(wd, cache=True)
        ops.gemm_f8(xq[0], xq[1], w8, sw, M, N, K, **epi)
    else:
        ops.gemm_lp(xb, _shadow(w), 0, M, N, K, **epi)


def _ok(M, N, K):
    return bool(ops._lib.load().qarig_gemm_lp_supported(M, N, K, 1))


def _splitk(tiles, K):
    """K slices for a weight-gradient TN product: fill the chip, whole 64-deep tiles."""
    s = max(1, min(512 // max(1, tiles), K // 512, 32))
    while s > 1 and (K % s or (K // s) % 64):
        s -= 1
    return s


def _wgrad(dTb, xb, w, n_rows):
    """dW (n_rows, K) = dT^T x over the token rows, fp32, accumulated into w.grad when it is a
    FlatAdam slot; dTb (M, Np >= n_rows) and xb (M, K) are the row-major bf16 activations."""
    M, Np = dTb.shape
    K = xb.shape[1]
    sk = _splitk((Np // 128) * (K // 128), M)
    if Np % 256 == 0 and K % 256 == 0:
        # 256 x 256-tile kernel (csrc/gemm_lp.hip: taken from 224 workgroups up): 2048 x 512 outputs
        # over 32768 rows measured 95 us there with 16 slices against 101 us on the 128-tiles with 8
        bt = (Np // 256) * (K // 256)
        s = 1
        while bt * s < 224:
            s *= 2
        if bt >= 16 and M % s == 0 and (M // s) % 64 == 0 and M // s >= 1024:
            sk = s
    slot = _grad_slot(w)
    if slot is not None and Np == n_rows:
        ops.gemm_lp(dTb, xb, 1, Np, K, M, C=slot, splitk=sk, accumulate=True)
        _report_done(w)
        return None
    dw = torch.empty((Np, K), dtype=torch.float32, device=xb.device)
    ops.gemm_lp(dTb, xb, 1, Np, K, M, C=dw, splitk=sk)
    dw = dw[:n_rows]
    if slot is not None:
        slot.add_(dw)
        _report_done(w)
        return None
    return dw


def _bias_grad(src, b, n, want_cast):
    """Column sums of src (M, Np) -> bias gradient (first n columns); returns (db or None,
    bf16 copy of src or None)."""
    Np = src.shape[1]
    slot = _grad_slot(b) if b is not None else None
    if b is None:
        return None, (ops.cast_bf16(src) if want_cast else None)
    if slot is not None and Np == n:
        cb = ops.cast_colsum(src, slot, accumulate=True, want_cast=want_cast)
        _report_done(b)
        return None, cb
    tmp = torch.zeros(Np, dtype=torch.float32, device=src.device)
    cb = ops.cast_colsum(src, tmp, accumulate=False, want_cast=want_cast)
    if slot is not None:
        slot.add_(tmp[:n])
        _report_done(b)
        return None, cb
    return tmp[:n], cb


class _MLP2LP(torch.autograd.Function):
    """y = act2(act1(x W1^T + b1) W2^T + b2), reduced precision (see module docstring)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act1, act2):
        require_cuda(x, w1, w2)
        shp = x.shape
        x2 = _2d(f32c(x))
        M, K = x2.shape
        H, N = w1.shape[0], w2.shape[0]
        Np = _pad128(N)
        xb, xq = _cast_in(x2, [(M, H, K)])
        hb = torch.empty((M, H), dtype=torch.bfloat16, device=x2.device)
        t1b = torch.empty((M, H), dtype=torch.bfloat16, device=x2.device) if act1 else None
        _fwd_nt(xq, xb, w1, M, H, K, bias=b1, act=act1, Cb=hb, Pb=t1b)
        y = torch.empty((M, Np), dtype=torch.float32, device=x2.device)
        t2 = torch.empty((M, Np), dtype=torch.float32, device=x2.device) if act2 else None
        b2p = b2
        if Np != N and b2 is not None:
            b2p = torch.zeros(Np, dtype=torch.float32, device=x2.device)
            b2p[:N].copy_(b2.detach())
        ops.gemm_lp(hb, _shadow(w2, pad_rows=Np), 0, M, Np, H, C=y, bias=b2p, preact=t2, act=act2)
        ctx.save_for_backward(xb, t1b if t1b is not None else hb, hb, t2 if t2 is not None else hb)
        ctx.cfg = (act1, act2, N, Np, shp)
        ctx.params = (w1, b1, w2, b2)
        if Np != N:
            y = y[:, :N].contiguous()
        return y.reshape(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xb, t1b, hb, t2 = ctx.saved_tensors
        act1, act2, N, Np, shp = ctx.cfg
        w1, b1, w2, b2 = ctx.params
        M, K = xb.shape
        H = hb.shape[1]
        dy2 = _2d(f32c(dy))
        if Np != N:
            pad = torch.zeros((M, Np), dtype=torch.float32, device=dy2.device)
            pad[:, :N].copy_(dy2)
            dy2 = pad
        dT2 = ops.act_bwd(dy2, t2, act2) if act2 else dy2
        db2, dT2b = _bias_grad(dT2, b2 if ctx.needs_input_grad[4] else None, N, True)
        dT1b = torch.empty((M, H), dtype=torch.bfloat16, device=dy2.device)
        ops.gemm_lp(dT2b, _shadow(w2, pad_rows=Np), 2, M, H, Np,
                    gradz=t1b if act1 else None, gact=act1, Cb=dT1b)
        dw2 = _wgrad(dT2b, hb, w2, N) if ctx.needs_input_grad[3] else None
        db1, _ = _bias_grad(dT1b, b1 if ctx.needs_input_grad[2] else None, H, False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy2.device)
            ops.gemm_lp(dT1b, _shadow(w1), 2, M, K, H, C=dx)
            dx = dx.reshape(shp)
        dw1 = _wgrad(dT1b, xb, w1, H) if ctx.needs_input_grad[1] else None
        return dx, dw1, db1, dw2, db2, None, None


class _MLP2x3LP(torch.autograd.Function):
    """The q, k, v MLPs of a self-attention layer on their common input: one bf16 cast of the
    input, the three input gradients accumulated by the GEMM epilogues into one tensor."""

    @staticmethod
    def forward(ctx, x, act1, act2, *params):
        require_cuda(x, *params)
        shp = x.shape
        x2 = _2d(f32c(x))
        M, K = x2.shape
        xb, xq = _cast_in(x2, [(M, params[4 * i].shape[0], K) for i in range(3)])
        outs, saved = [], [xb]
        for i in range(3):
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            H, N = w1.shape[0], w2.shape[0]
            hb = torch.empty((M, H), dtype=torch.bfloat16, device=x2.device)
            t1b = torch.empty((M, H), dtype=torch.bfloat16, device=x2.device) if act1 else hb
            _fwd_nt(xq, xb, w1, M, H, K, bias=b1, act=act1, Cb=hb, Pb=t1b if act1 else None)
            y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
            t2 = torch.empty((M, N), dtype=torch.float32, device=x2.device) if act2 else None
            ops.gemm_lp(hb, _shadow(w2), 0, M, N, H, C=y, bias=b2, preact=t2, act=act2)
            outs.append(y.reshape(*shp[:-1], N))
            saved += [t1b, hb, t2 if t2 is not None else hb]
        ctx.save_for_backward(*saved)
        ctx.cfg = (act1, act2, shp)
        ctx.params = params
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors
        act1, act2, shp = ctx.cfg
        xb = saved[0]
        M, K = xb.shape
        grads = []
        dx = None
        for i in range(3):
            w1, b1, w2, b2 = ctx.params[4 * i:4 * i + 4]
            t1b, hb, t2 = saved[1 + 3 * i:4 + 3 * i]
            H, N = w1.shape[0], w2.shape[0]
            ni = 3 + 4 * i
            dy2 = _2d(f32c(dys[i]))
            dT2 = ops.act_bwd(dy2, t2, act2) if act2 else dy2
            db2, dT2b = _bias_grad(dT2, b2 if ctx.needs_input_grad[ni + 3] else None, N, True)
            dT1b = torch.empty((M, H), dtype=torch.bfloat16, device=dy2.device)
            ops.gemm_lp(dT2b, _shadow(w2), 2, M, H, N, gradz=t1b if act1 else None,
                        gact=act1, Cb=dT1b)
            dw2 = _wgrad(dT2b, hb, w2, N) if ctx.needs_input_grad[ni + 2] else None
            db1, _ = _bias_grad(dT1b, b1 if ctx.needs_input_grad[ni + 1] else None, H, False)
            if ctx.needs_input_grad[0]:
                if dx is None:
                    dx = torch.empty((M, K), dtype=torch.float32, device=dy2.device)
                    ops.gemm_lp(dT1b, _shadow(w1), 2, M, K, H, C=dx)
                else:
                    ops.gemm_lp(dT1b, _shadow(w1), 2, M, K, H, C=dx, accumulate=True)
            dw1 = _wgrad(dT1b, xb, w1, H) if ctx.needs_input_grad[ni] else None
            grads += [dw1, db1, dw2, db2]
        if dx is not None:
            dx = dx.reshape(shp)
        return (dx, None, None, *grads)


class _LinearActLP(torch.autograd.Function):
    """y = act(x W^T + b [+ residual]), reduced precision."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act):
        require_cuda(x, weight)
        shp = x.shape
        x2 = _2d(f32c(x))
        M, K = x2.shape
        N = weight.shape[0]
        xb, xq = _cast_in(x2, [(M, N, K)])
        r2 = _2d(f32c(residual)) if residual is not None else None
        y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
        t = torch.empty((M, N), dtype=torch.float32, device=x2.device) if act else None
        _fwd_nt(xq, xb, weight, M, N, K, C=y, bias=bias, residual=r2, preact=t, act=act)
        ctx.save_for_backward(xb, t if t is not None else xb)
        ctx.cfg = (act, residual is not None, shp)
        ctx.params = (weight, bias)
        return y.reshape(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xb, t = ctx.saved_tensors
        act, has_res, shp = ctx.cfg
        weight, bias = ctx.params
        M, K = xb.shape
        N = weight.shape[0]
        dy2 = _2d(f32c(dy))
        dT = ops.act_bwd(dy2, t, act) if act else dy2
        db, dTb = _bias_grad(dT, bias if (bias is not None and ctx.needs_input_grad[2]) else None, N, True)
        dx = dw = dr = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy2.device)
            ops.gemm_lp(dTb, _shadow(weight), 2, M, K, N, C=dx)
            dx = dx.reshape(shp)
        if ctx.needs_input_grad[1]:
            dw = _wgrad(dTb, xb, weight, N)
        if has_res and ctx.needs_input_grad[3]:
            dr = dT.reshape(dy.shape)
        return dx, dw, db, dr, None


def mlp2_supported(x, w1, w2):
    M = x.numel() // x.shape[-1]
    K, H, N = w1.shape[1], w1.shape[0], w2.shape[0]
    Np = _pad128(N)
    # forward (M,H,K), (M,Np,H); d-input (M,H,Np), (M,K,H); d-weight (Np,H,M), (H,K,M)
    return (ops.lp_mode() and M >= 1024 and _ok(M, H, K) and _ok(M, Np, H) and _ok(M, H, Np)
            and _ok(M, K, H) and _ok(Np, H, M) and _ok(H, K, M))


def linear_supported(x, weight):
    M = x.numel() // x.shape[-1]
    N, K = weight.shape
    return (ops.lp_mode() and M >= 1024 and _ok(M, N, K) and _ok(M, K, N) and _ok(N, K, M))
