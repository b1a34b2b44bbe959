"""Differentiable building blocks: torch.autograd.Function shells whose forward and
backward are the HIP kernels behind the C ABI (qarig.ops).  torch owns the tensors and
the graph; every FLOP and every byte moved is ours.

Saved-tensor policy: HBM is 288 GB per GPU, so every pre-activation that a backward
needs is kept (no recompute); compatible with torch.utils.checkpoint(use_reentrant=
False) for the reference's --use-activation-checkpoint flag.
"""
import os

import torch

from . import ops
from ._lib import f32c, require_cuda


def _2d(t):
    return t.reshape(-1, t.shape[-1])


# When a parameter already owns a contiguous .grad (FlatAdam points every .grad into its
# flat buffer), weight/bias gradients are accumulated there by the kernel itself and the
# autograd node returns None for them: no temporary, no elementwise add pass.
FUSED_GRAD_ACCUMULATE = True


def _grad_slot(param):
    if not FUSED_GRAD_ACCUMULATE or not isinstance(param, torch.nn.Parameter):
        return None
    g = param.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.shape != param.shape:
        return None
    return g


def _report_done(param):
    """Tells an overlapping optimiser (FlatAdam.enable_allreduce_overlap) that this
    parameter's gradient is final for the step."""
    cb = getattr(param, "_qarig_grad_done", None)
    if cb is not None:
        cb(param)


def _wgrad(dT, X, param=None, bias_param=None, want_bias=False, flop_frac=1.0):
    """dW[N,K] = dT^T X over the row dimension (split-K through fp32 slabs) and, riding on
    the same launch, db[N] = column sums of dT (the row sums of the A operand dT^T).
    Returns (dW, db); an entry is None when it was accumulated into the parameter's .grad
    (or not requested)."""
    M, N = dT.shape
    K = X.shape[1]
    sk = ops.pick_splitk(N, K, M)
    wslot = _grad_slot(param)
    bslot = _grad_slot(bias_param) if want_bias else None
    if wslot is not None and (not want_bias or bslot is not None):
        ops.gemm(dT, X, a_kcontig=False, b_kcontig=False, splitk=sk, out=wslot, accumulate=True,
                 a_rowsum=bslot)
        _report_done(param)
        if want_bias:
            _report_done(bias_param)
        return None, None
    db = torch.zeros(N, dtype=torch.float32, device=dT.device) if want_bias else None
    dw = ops.gemm(dT, X, a_kcontig=False, b_kcontig=False, splitk=sk, a_rowsum=db, flop_frac=flop_frac)
    return dw, db


def _bgrad(dT, param=None):
    slot = _grad_slot(param)
    if slot is not None:
        ops.colsum(dT, out=slot, accumulate=True)
        _report_done(param)
        return None
    return ops.colsum(dT)


class _LinearAct(torch.autograd.Function):
    """y = act(x W^T + b [+ residual]) -- LinearLayer / ResidualLinearLayer core
    (reference models/layers.py:234-254, 297-303)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act):
        require_cuda(x, weight)
        shp = x.shape
        x2 = _2d(f32c(x))
        r2 = _2d(f32c(residual)) if residual is not None else None
        if act:
            y, t = ops.gemm(x2, weight, bias=bias, residual=r2, want_preact=True, act=act)
        else:
            y, t = ops.gemm(x2, weight, bias=bias, residual=r2), None
        ctx.save_for_backward(x2, weight, t)
        ctx.act = act
        ctx.has_res = residual is not None
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        return y.reshape(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, t = ctx.saved_tensors
        dy2 = _2d(f32c(dy))
        dT = ops.act_bwd(dy2, t, ctx.act) if ctx.act else dy2
        dx = dw = db = dr = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dT, weight, a_kcontig=True, b_kcontig=False).reshape(
                *dy.shape[:-1], weight.shape[1])
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = _wgrad(dT, x2, ctx.params[0], ctx.params[1], want_b)
        elif want_b:
            db = _bgrad(dT, ctx.params[1])
        if ctx.has_res and ctx.needs_input_grad[3]:
            dr = dT.reshape(dy.shape)
        return dx, dw, db, dr, None


def _no_grad():
    return not torch.is_grad_enabled()


def _lp():
    """Reduced-precision nodes (qarig.functional_lp), imported on first use."""
    from . import functional_lp
    return functional_lp


def linear_act(x, weight, bias=None, residual=None, act=0):
    if ops.lp_mode() and not _no_grad() and _lp().linear_supported(x, weight):
        return _lp()._LinearActLP.apply(x, weight, bias, residual, act)
    if _no_grad():   # inference: straight to the kernel, no autograd node, no saved tensors
        require_cuda(x, weight)
        r2 = _2d(f32c(residual)) if residual is not None else None
        y = ops.gemm(_2d(f32c(x)), weight, bias=bias, residual=r2, act=act)
        return y.reshape(*x.shape[:-1], weight.shape[0])
    return _LinearAct.apply(x, weight, bias, residual, act)


class _MLP2(torch.autograd.Function):
    """y = act2(act1(x W1^T + b1) W2^T + b2): the two-layer MLPs that make up q/k/v
    blocks, the FFN, the pos-cond MLP and the classifier (reference
    models/layers.py:330-340, 389-418; models/Transformer.py:82-102).  Backward fuses
    act1' into the epilogue of the dH GEMM."""

    @staticmethod
    def _padded_out(M, N, K, act2, b2):
        """A wide, ragged output layer (the classifier: 513 = K_hr + 1 columns over N*S rows) runs
        on zero-padded copies of its weight -- whole 128-column tiles, hence the interior GEMM
        kernels forward, d-input and d-weight -- instead of the guarded kernels (575 -> ~360 us
        per GEMM at 16384 x 513 x 2048); the padding columns carry zero gradients."""
        return N % 128 != 0 and N > 256 and M >= 4096 and K % 16 == 0 and not act2 and b2 is not None

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act1, act2):
        require_cuda(x, w1, w2)
        shp = x.shape
        x2 = _2d(f32c(x))
        h, t1 = ops.gemm(x2, w1, bias=b1, want_preact=True, act=act1)
        N, K = w2.shape
        ctx.pad = _MLP2._padded_out(x2.shape[0], N, K, act2, b2)
        if ctx.pad:
            Np = (N + 127) // 128 * 128
            w2p = torch.zeros((Np, K), dtype=torch.float32, device=w2.device)
            w2p[:N].copy_(w2.detach())
            b2p = torch.zeros(Np, dtype=torch.float32, device=w2.device)
            b2p[:N].copy_(b2.detach())
            y, t2 = ops.gemm(h, w2p, bias=b2p, flop_frac=N / Np)[:, :N].contiguous(), None
            ctx.save_for_backward(x2, w1, w2p, t1, h, t2)
        else:
            if act2:
                y, t2 = ops.gemm(h, w2, bias=b2, want_preact=True, act=act2)
            else:
                y, t2 = ops.gemm(h, w2, bias=b2), None
            ctx.save_for_backward(x2, w1, w2, t1, h, t2)
        ctx.act1, ctx.act2 = act1, act2
        ctx.params = (w1, b1, w2, b2)
        return y.reshape(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, t1, h, t2 = ctx.saved_tensors
        dy2 = _2d(f32c(dy))
        dx = dw1 = db1 = dw2 = db2 = None
        if ctx.pad:
            N = ctx.params[2].shape[0]
            Np = w2.shape[0]                      # w2 here is the zero-padded copy
            dT2 = torch.zeros((dy2.shape[0], Np), dtype=torch.float32, device=dy2.device)
            dT2[:, :N].copy_(dy2)
            dT1 = ops.gemm(dT2, w2, a_kcontig=True, b_kcontig=False, gradz=t1, gact=ctx.act1, flop_frac=N / Np)
            if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
                dwp, dbp = _wgrad(dT2, h, None, None, True, flop_frac=N / Np)
                wslot, bslot = _grad_slot(ctx.params[2]), _grad_slot(ctx.params[3])
                if ctx.needs_input_grad[3]:
                    if wslot is not None:
                        wslot.add_(dwp[:N])
                        _report_done(ctx.params[2])
                    else:
                        dw2 = dwp[:N]
                if ctx.needs_input_grad[4]:
                    if bslot is not None:
                        bslot.add_(dbp[:N])
                        _report_done(ctx.params[3])
                    else:
                        db2 = dbp[:N]
        else:
            dT2 = ops.act_bwd(dy2, t2, ctx.act2) if ctx.act2 else dy2
            dT1 = ops.gemm(dT2, w2, a_kcontig=True, b_kcontig=False, gradz=t1, gact=ctx.act1)
            if ctx.needs_input_grad[3]:
                dw2, db2 = _wgrad(dT2, h, ctx.params[2], ctx.params[3], ctx.needs_input_grad[4])
            elif ctx.needs_input_grad[4]:
                db2 = _bgrad(dT2, ctx.params[3])
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dT1, w1, a_kcontig=True, b_kcontig=False).reshape(
                *dy.shape[:-1], w1.shape[1])
        if ctx.needs_input_grad[1]:
            dw1, db1 = _wgrad(dT1, x2, ctx.params[0], ctx.params[1], ctx.needs_input_grad[2])
        elif ctx.needs_input_grad[2]:
            db1 = _bgrad(dT1, ctx.params[1])
        return dx, dw1, db1, dw2, db2, None, None


def mlp2(x, w1, b1, w2, b2, act1, act2=0):
    if ops.lp_mode() and not _no_grad() and _lp().mlp2_supported(x, w1, w2):
        return _lp()._MLP2LP.apply(x, w1, b1, w2, b2, act1, act2)
    if _no_grad():
        require_cuda(x, w1, w2)
        h = ops.gemm(_2d(f32c(x)), w1, bias=b1, act=act1)
        return ops.gemm(h, w2, bias=b2, act=act2).reshape(*x.shape[:-1], w2.shape[0])
    return _MLP2.apply(x, w1, b1, w2, b2, act1, act2)


class _MLP2x3(torch.autograd.Function):
    """The q, k and v two-layer MLPs of a SELF-attention layer (reference models/layers.py:389-418)
    on their common input: same six GEMMs forward as three `_MLP2`s; in backward the three
    input gradients are accumulated by the GEMM epilogues into one tensor (dx = dT1_q W1_q,
    then += dT1_k W1_k, += dT1_v W1_v) instead of three tensors and two elementwise adds."""

    @staticmethod
    def forward(ctx, x, act1, act2, *params):
        require_cuda(x, *params)
        shp = x.shape
        x2 = _2d(f32c(x))
        outs, saved = [], [x2]
        for i in range(3):
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            h, t1 = ops.gemm(x2, w1, bias=b1, want_preact=True, act=act1)
            if act2:
                y, t2 = ops.gemm(h, w2, bias=b2, want_preact=True, act=act2)
            else:
                y, t2 = ops.gemm(h, w2, bias=b2), None
            outs.append(y.reshape(*shp[:-1], w2.shape[0]))
            saved += [t1, h, t2 if t2 is not None else x2.new_empty(0)]
        ctx.save_for_backward(*saved)
        ctx.act1, ctx.act2 = act1, act2
        ctx.params = params
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors
        x2 = saved[0]
        grads = []
        dx = None
        for i in range(3):
            w1, b1, w2, b2 = ctx.params[4 * i:4 * i + 4]
            t1, h, t2 = saved[1 + 3 * i:4 + 3 * i]
            dy2 = _2d(f32c(dys[i]))
            dT2 = ops.act_bwd(dy2, t2, ctx.act2) if ctx.act2 else dy2
            dT1 = ops.gemm(dT2, w2, a_kcontig=True, b_kcontig=False, gradz=t1, gact=ctx.act1)
            ni = 3 + 4 * i          # needs_input_grad index of w1
            dw1 = db1 = dw2 = db2 = None
            if ctx.needs_input_grad[ni + 2]:
                dw2, db2 = _wgrad(dT2, h, w2, b2, ctx.needs_input_grad[ni + 3])
            elif ctx.needs_input_grad[ni + 3]:
                db2 = _bgrad(dT2, b2)
            if ctx.needs_input_grad[0]:
                if dx is None:
                    dx = ops.gemm(dT1, w1, a_kcontig=True, b_kcontig=False)
                else:
                    ops.gemm(dT1, w1, a_kcontig=True, b_kcontig=False, out=dx, accumulate=True, splitk=1)
            if ctx.needs_input_grad[ni]:
                dw1, db1 = _wgrad(dT1, x2, w1, b1, ctx.needs_input_grad[ni + 1])
            elif ctx.needs_input_grad[ni + 1]:
                db1 = _bgrad(dT1, b1)
            grads += [dw1, db1, dw2, db2]
        if dx is not None:
            dx = dx.reshape(*dys[0].shape[:-1], x2.shape[1])
        return (dx, None, None, *grads)


# Grouped launches for MLPs that share their input (csrc/gemm.hip gemm_dma_pf_grouped_kernel): taken when
# the per-product tile count leaves the chip under-filled (M = a few thousand rows: the per-GPU shard
# of a small global batch).  QARIG_MLP_GROUPED=0 disables, =1 forces it on every supported shape.
MLP_GROUPED = os.environ.get("QARIG_MLP_GROUPED")
# cross-attention k / v MLPs: "all" (every decoder layer's pair in one grouped launch, evaluated
# before the layer loop), "layer" (each layer its own pair) or None = "layer" under torch.distributed
# with more than one rank, else "all" (models/Transformer.py _cross_kv_all_layers)
CROSS_KV_GROUPING = os.environ.get("QARIG_CROSS_KV_GROUPING") or None
MLP_GROUPED_MAX_ROWS = 8192


class _MLP2xG(torch.autograd.Function):
    """G two-layer MLPs y_g = act2(act1(x W1_g^T + b1_g) W2_g^T + b2_g) on ONE input x, all of one
    shape: the q / k / v blocks of a self-attention layer, or the cross-attention k / v blocks of
    one or of every decoder layer (reference models/layers.py:389-418, :581-599).  Every product --
    both forward layers, dT1, dx (the sum over the G blocks, in block order), both weight gradients
    with their bias gradients -- is ONE grouped launch (+ one grouped reduce where the reduction is
    split) instead of G; the hidden tensors of the G blocks are slices of one allocation."""

    @staticmethod
    def forward(ctx, x, act1, act2, *params):
        require_cuda(x, *params)
        G = len(params) // 4
        shp = x.shape
        x2 = _2d(f32c(x))
        M, D = x2.shape
        w1s, b1s, w2s, b2s = params[0::4], params[1::4], params[2::4], params[3::4]
        H, O = w1s[0].shape[0], w2s[0].shape[0]
        dev = x2.device
        hs = torch.empty((G, M, H), dtype=torch.float32, device=dev)
        t1s = torch.empty((G, M, H), dtype=torch.float32, device=dev)
        ops.gemm_grouped([x2] * G, w1s, list(hs.unbind(0)), M, H, D, bias=b1s, preact=list(t1s.unbind(0)),
                         act=act1, splitk=ops.grouped_splitk(G, M, H, D))
        ys = torch.empty((G, M, O), dtype=torch.float32, device=dev)
        t2s = torch.empty((G, M, O), dtype=torch.float32, device=dev) if act2 else None
        ops.gemm_grouped(list(hs.unbind(0)), w2s, list(ys.unbind(0)), M, O, H, bias=b2s,
                         preact=list(t2s.unbind(0)) if act2 else None, act=act2,
                         splitk=ops.grouped_splitk(G, M, O, H))
        ctx.save_for_backward(x2, hs, t1s, t2s if act2 else x2.new_empty(0))
        ctx.act1, ctx.act2, ctx.G = act1, act2, G
        ctx.params = params
        return tuple(y.reshape(*shp[:-1], O) for y in ys.unbind(0))

    @staticmethod
    def backward(ctx, *dys):
        x2, hs, t1s, t2s = ctx.saved_tensors
        G = ctx.G
        params = ctx.params
        w1s, b1s, w2s, b2s = params[0::4], params[1::4], params[2::4], params[3::4]
        M, D = x2.shape
        H, O = w1s[0].shape[0], w2s[0].shape[0]
        dev = x2.device
        dT2 = [_2d(f32c(d)) for d in dys]
        if ctx.act2:
            dT2 = [ops.act_bwd(d, t2s[g], ctx.act2) for g, d in enumerate(dT2)]
        h = list(hs.unbind(0))
        # dT1_g = (dT2_g W2_g) * act1'(t1_g)
        dT1s = torch.empty((G, M, H), dtype=torch.float32, device=dev)
        dT1 = list(dT1s.unbind(0))
        ops.gemm_grouped(dT2, w2s, dT1, M, H, O, a_kcontig=True, b_kcontig=False, gradz=list(t1s.unbind(0)),
                         gact=ctx.act1, splitk=ops.grouped_splitk(G, M, H, O))

        def wgrads(dT, X, ws, bs, No, Ko):
            """dW_g (No, Ko) = dT_g^T X_g over the M rows, db_g = column sums of dT_g."""
            wslots, bslots = [_grad_slot(p) for p in ws], [_grad_slot(p) for p in bs]
            inplace = all(s is not None for s in wslots) and all(s is not None for s in bslots)
            if inplace:
                outs, rsum = wslots, bslots
            else:
                dw = torch.empty((G, No, Ko), dtype=torch.float32, device=dev)
                db = torch.empty((G, No), dtype=torch.float32, device=dev)
                outs, rsum = list(dw.unbind(0)), list(db.unbind(0))
            ops.gemm_grouped(dT, X, outs, No, Ko, M, a_kcontig=False, b_kcontig=False, accumulate=inplace,
                             a_rowsum=rsum, splitk=ops.grouped_splitk(G, No, Ko, M))
            if inplace:
                for p in (*ws, *bs):
                    _report_done(p)
                return [None] * G, [None] * G
            return outs, rsum

        dw2, db2 = wgrads(dT2, h, w2s, b2s, O, H)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, D), dtype=torch.float32, device=dev)
            ops.gemm_grouped(dT1, w1s, [dx], M, D, H, a_kcontig=True, b_kcontig=False, sum_groups=True,
                             splitk=ops.grouped_splitk(G, M, D, H))
            dx = dx.reshape(*dys[0].shape[:-1], D)
        dw1, db1 = wgrads(dT1, [x2] * G, w1s, b1s, H, D)
        grads = []
        for g in range(G):
            grads += [dw1[g], db1[g], dw2[g], db2[g]]
        return (dx, None, None, *grads)


def _mlp_group_ok(x, blocks_params, M):
    """Same-shape blocks on dense 16-B aligned fp32 parameters that all want gradients, at a row
    count where one product alone under-fills the chip, on shapes the grouped kernel takes."""
    if MLP_GROUPED == "0" or ops.lp_mode() or not 2 <= len(blocks_params) <= ops.GEMM_MAX_GROUPS:
        return False
    if MLP_GROUPED != "1" and M > MLP_GROUPED_MAX_ROWS:
        return False
    w1, b1, w2, b2 = blocks_params[0]
    if b1 is None or b2 is None or w1.dim() != 2 or w2.dim() != 2 or w2.shape[1] != w1.shape[0]:
        return False
    for p in blocks_params:
        if (p[0].shape != w1.shape or p[2].shape != w2.shape or p[1] is None or p[3] is None
                or p[1].shape != (w1.shape[0],) or p[3].shape != (w2.shape[0],)):
            return False
        for t in p:
            if (t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16
                    or not t.requires_grad or not t.is_cuda):
                return False
    H, D = w1.shape
    O = w2.shape[0]
    sup = ops.gemm_grouped_supported
    G = len(blocks_params)
    return (x.shape[-1] == D and sup(M, H, D, ops.grouped_splitk(G, M, H, D))
            and sup(M, O, H, ops.grouped_splitk(G, M, O, H)) and sup(M, H, O, ops.grouped_splitk(G, M, H, O))
            and sup(M, D, H, ops.grouped_splitk(G, M, D, H)) and sup(O, H, M, ops.grouped_splitk(G, O, H, M))
            and sup(H, D, M, ops.grouped_splitk(G, H, D, M)))


def mlp2xg(x, blocks_params, act1, act2=0):
    """blocks_params: G (w1, b1, w2, b2) tuples of one shape, all applied to x.  Returns G tensors."""
    M = x.numel() // x.shape[-1]
    if not _no_grad() and _mlp_group_ok(x, blocks_params, M):
        return _MLP2xG.apply(x, act1, act2, *[t for p in blocks_params for t in p])
    if len(blocks_params) == 3:
        return mlp2x3(x, blocks_params, act1, act2, _grouped=False)
    return tuple(mlp2(x, *p, act1, act2) for p in blocks_params)


def mlp2x3(x, blocks_params, act1, act2=0, _grouped=True):
    """blocks_params: three (w1, b1, w2, b2) tuples (q, k, v).  Returns (q, k, v)."""
    if _no_grad():
        return tuple(mlp2(x, *p, act1, act2) for p in blocks_params)
    if _grouped and _mlp_group_ok(x, blocks_params, x.numel() // x.shape[-1]):
        return _MLP2xG.apply(x, act1, act2, *[t for p in blocks_params for t in p])
    if ops.lp_mode() and all(_lp().mlp2_supported(x, p[0], p[2]) and p[2].shape[0] % 128 == 0
                                       for p in blocks_params):
        return _lp()._MLP2x3LP.apply(x, act1, act2, *[t for p in blocks_params for t in p])
    return _MLP2x3.apply(x, act1, act2, *[t for p in blocks_params for t in p])


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        require_cuda(a, b)
        a, b = f32c(a), f32c(b)
        ctx.save_for_backward(a, b)
        return ops.mul_fwd(a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        da, db = ops.mul_bwd(f32c(dy), a, b)
        return da, db


def mul(a, b):
    if _no_grad():
        require_cuda(a, b)
        return ops.mul_fwd(f32c(a), f32c(b))
    return _Mul.apply(a, b)


def _with_skip(y, x, with_skip):
    """(y, alias of x) for a normalisation node that also hands x to the block's skip connection."""
    return (y, x.view_as(x)) if with_skip else y


def _skip_grads(dy, dskip, x2):
    """Upstream gradients of a with_skip node -> (dy (M,D), skip gradient (M,D) or None).  Either
    may be missing (set_materialize_grads(False)): a consumer that ignores one of the outputs."""
    if dy is None:
        dy2 = torch.zeros_like(x2)
    else:
        dy2 = _2d(f32c(dy))
    add = _2d(f32c(dskip)) if dskip is not None else None
    return dy2, add


class _LayerNormAffine(torch.autograd.Function):
    """nn.LayerNorm(D) with gamma/beta (encoder blocks)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, with_skip=False):
        require_cuda(x, gamma, beta)
        x2 = _2d(f32c(x))
        y, mean, rstd = ops.layernorm_fwd(x2, gamma=gamma, beta=beta, eps=eps)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shape = x.shape
        ctx.params = (gamma, beta)
        ctx.set_materialize_grads(False)
        return _with_skip(y.reshape(x.shape), x, with_skip)

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x2, gamma, mean, rstd = ctx.saved_tensors
        dy2, add = _skip_grads(dy, dskip, x2)
        dx, dyx = ops.layernorm_bwd(dy2, x2, mean, rstd, gamma=gamma, want_dy_xhat=True, dx_add=add)
        gslot, bslot = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
        if gslot is not None and bslot is not None:     # column sums straight into the .grad slots
            ops.colsum(dyx, out=gslot, accumulate=True)
            ops.colsum(dy2, out=bslot, accumulate=True)
            _report_done(ctx.params[0])
            _report_done(ctx.params[1])
            return dx.reshape(ctx.shape), None, None, None, None
        return dx.reshape(ctx.shape), ops.colsum(dyx), ops.colsum(dy2), None, None


def layernorm_affine(x, gamma, beta, eps=1e-5, with_skip=False):
    """with_skip: returns (y, x_alias); the block's skip connection takes x_alias, so that both
    gradients of x arrive in this node's backward and are added by the LayerNorm kernel."""
    if _no_grad():
        require_cuda(x, gamma, beta)
        y = ops.layernorm_fwd(_2d(f32c(x)), gamma=gamma, beta=beta, eps=eps)[0].reshape(x.shape)
        return (y, x) if with_skip else y
    return _LayerNormAffine.apply(x, gamma, beta, eps, with_skip)


class _LayerNormMod(torch.autograd.Function):
    """AdaLNZero: scale * LayerNorm_noaffine(x) + shift, scale/shift per token
    (reference models/layers.py:146-153)."""

    @staticmethod
    def forward(ctx, x, scale, shift, eps, with_skip=False):
        require_cuda(x, scale, shift)
        x2 = _2d(f32c(x))
        s2 = _2d(f32c(scale))
        y, mean, rstd = ops.layernorm_fwd(x2, scale=s2, shift=_2d(f32c(shift)), eps=eps)
        ctx.save_for_backward(x2, s2, mean, rstd)
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)
        return _with_skip(y.reshape(x.shape), x, with_skip)

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x2, s2, mean, rstd = ctx.saved_tensors
        dy2, add = _skip_grads(dy, dskip, x2)
        dx, dscale = ops.layernorm_bwd(dy2, x2, mean, rstd, scale=s2, want_dy_xhat=True, dx_add=add)
        return dx.reshape(ctx.shape), dscale.reshape(ctx.shape), dy2.reshape(ctx.shape), None, None


def layernorm_mod(x, scale, shift, eps=1e-5, with_skip=False):
    if _no_grad():
        require_cuda(x, scale, shift)
        y = ops.layernorm_fwd(_2d(f32c(x)), scale=_2d(f32c(scale)), shift=_2d(f32c(shift)),
                              eps=eps)[0].reshape(x.shape)
        return (y, x) if with_skip else y
    return _LayerNormMod.apply(x, scale, shift, eps, with_skip)


class _Attention(torch.autograd.Function):
    """softmax(QK^T/sqrt(d) [causal]) V per head on (N,S,H*d) tensors
    (reference models/layers.py:433-474).  scale_dim: the model's head dim when the heads arrive
    zero-padded to a kernel head dim (see `attention`)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, causal, scale_dim=None):
        require_cuda(q, k, v)
        q, k, v = f32c(q), f32c(k), f32c(v)
        o, lse = ops.attention_fwd(q, k, v, heads, causal, scale_dim)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.heads, ctx.causal, ctx.scale_dim = heads, causal, scale_dim
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = ops.attention_bwd(q, k, v, o, f32c(do), lse, ctx.heads, ctx.causal, ctx.scale_dim)
        return dq, dk, dv, None, None, None


def _pad_heads(t, heads, d, dp):
    """(N,S,H*d) -> (N,S,H*dp), every head zero-padded from d to dp columns."""
    N, S, _ = t.shape
    return torch.nn.functional.pad(t.reshape(N, S, heads, d), (0, dp - d)).reshape(N, S, heads * dp)


def attention(q, k, v, heads, causal):
    """The reference accepts any `heads` that divides the model width (models/layers.py:433); the kernels
    are instantiated for head dims 4, 8, 16, 32, 64 (MFMA, csrc/attention.hip) and 128 (vector-ALU fma chains,
    csrc/attention_wide.hip: the rare few-heads model).  Any other head dim up to 128 runs on the next
    instantiated one with every head zero-padded: the extra columns add exact zeros to q.k and produce
    output columns that are dropped again (softmax scale from the model's head dim), so the result is the
    reference's; pad and slice are torch ops (a rare shape: not a hot path)."""
    d = q.shape[-1] // heads
    dp = ops.attention_head_dim(d)
    if dp is None:
        raise ValueError(f"attention: head dim {d} (= {q.shape[-1]} / {heads} heads) is wider than the widest "
                         f"MI355X kernel ({ops.ATTENTION_HEAD_DIMS[-1]})")
    if dp != d:
        qp, kp, vp = (_pad_heads(t, heads, d, dp) for t in (q, k, v))
        if _no_grad():
            require_cuda(q, k, v)
            o = ops.attention_fwd(f32c(qp), f32c(kp), f32c(vp), heads, causal, d)[0]
        else:
            o = _Attention.apply(qp, kp, vp, heads, causal, d)
        N, S, _ = o.shape
        return o.reshape(N, S, heads, dp)[..., :d].reshape(N, S, heads * d)
    if _no_grad():
        require_cuda(q, k, v)
        return ops.attention_fwd(f32c(q), f32c(k), f32c(v), heads, causal)[0]
    return _Attention.apply(q, k, v, heads, causal)


class _EmbeddingPos(torch.autograd.Function):
    """table[ids] + pe[s] (reference models/Transformer.py:127-139, 154-167)."""

    @staticmethod
    def forward(ctx, ids, table, pe):
        require_cuda(ids, table)
        ctx.save_for_backward(ids)
        ctx.V = table.shape[0]
        return ops.embedding_fwd(ids, table, pe)

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        D = dy.shape[-1]
        if ids.numel() >= 4096 and D % 4 == 0:
            # many tokens per vocabulary row: row map once, then each table row sums its own
            # tokens (reads dy once) instead of every row scanning the whole id stream
            off, rows = ops.rowmap_build(ids.reshape(-1).to(torch.int32), ctx.V)
            return None, ops.segment_sum(_2d(f32c(dy)), off, rows), None
        return None, ops.embedding_bwd(ids, dy, ctx.V), None


def embedding_pos(ids, table, pe=None):
    if _no_grad():
        require_cuda(ids, table)
        return ops.embedding_fwd(ids, table, pe)
    return _EmbeddingPos.apply(ids, table, pe)


class _CrossEntropy(torch.autograd.Function):
    """mean CE over rows; the logits gradient is produced in the forward pass."""

    @staticmethod
    def forward(ctx, logits, target):
        require_cuda(logits, target)
        l2 = _2d(f32c(logits))
        t = target.reshape(-1)
        if t.dtype != torch.int64:
            t = t.long()
        loss, dl = ops.cross_entropy_fwd(l2, t.contiguous(), want_grad=True)
        ctx.save_for_backward(dl)
        ctx.shape = logits.shape
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dl,) = ctx.saved_tensors
        return ops.scale_by(dl, f32c(dloss).reshape(1)).reshape(ctx.shape), None


def cross_entropy(logits, target):
    return _CrossEntropy.apply(logits, target)


class _Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        require_cuda(x)
        x = f32c(x)
        ctx.save_for_backward(x)
        ctx.act = act
        return ops.act_fwd(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.act_bwd(f32c(dy), x, ctx.act), None


def activation(x, act):
    return _Activation.apply(x, act) if act else x


class _Conv2dAct(torch.autograd.Function):
    """Conv2d + bias + activation (reference models/layers.py:157-184, 211-230)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act):
        require_cuda(x, weight, bias)
        assert weight.is_contiguous() and weight.dtype == torch.float32, "conv weight must be dense fp32"
        x = f32c(x)     # the kernels (forward AND both gradients) read dense NCHW fp32: save this one
        need = any(ctx.needs_input_grad)
        if need and act:
            y, pre = ops.conv2d_fwd(x, weight, bias, stride, pad, act, want_preact=True)
        else:
            y, pre = ops.conv2d_fwd(x, weight, bias, stride, pad, act), None
        ctx.save_for_backward(x, weight, pre)
        ctx.cfg = (stride, pad, act, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import conv_backward
        x, weight, pre = ctx.saved_tensors
        stride, pad, act, has_bias = ctx.cfg
        return conv_backward.conv2d_bwd(ctx, dy, x, weight, pre, stride, pad, act, has_bias) + \
            (None, None, None)


def conv2d_act(x, weight, bias, stride=1, pad=1, act=0):
    # decided HERE: inside an autograd.Function.forward grad mode is always off.  No grad: the weights are
    # constants between optimiser steps, their re-ordered copies are cached per weight (ops._conv_workspace);
    # a training forward re-orders into the shared scratch on every call
    if _no_grad():
        require_cuda(x, weight, bias)
        assert weight.is_contiguous() and weight.dtype == torch.float32, "conv weight must be dense fp32"
        return ops.conv2d_fwd(f32c(x), weight, bias, stride, pad, act, inference=True)
    return _Conv2dAct.apply(x, weight, bias, stride, pad, act)


class _ConvT2dAct(torch.autograd.Function):
    """ConvTranspose2d(4,2,1) + bias + activation (reference models/layers.py:188-207)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        require_cuda(x, weight, bias)
        assert weight.is_contiguous() and weight.dtype == torch.float32, "conv weight must be dense fp32"
        x = f32c(x)
        need = any(ctx.needs_input_grad)
        if need and act:
            y, pre = ops.conv_transpose2d_fwd(x, weight, bias, act, want_preact=True)
        else:
            y, pre = ops.conv_transpose2d_fwd(x, weight, bias, act), None
        ctx.save_for_backward(x, weight, pre)
        ctx.cfg = (act, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import conv_backward
        x, weight, pre = ctx.saved_tensors
        act, has_bias = ctx.cfg
        return conv_backward.conv_transpose2d_bwd(ctx, dy, x, weight, pre, act, has_bias) + (None,)


def conv_transpose2d_act(x, weight, bias, act=0):
    if _no_grad():          # (see conv2d_act)
        require_cuda(x, weight, bias)
        assert weight.is_contiguous() and weight.dtype == torch.float32, "conv weight must be dense fp32"
        return ops.conv_transpose2d_fwd(f32c(x), weight, bias, act, inference=True)
    return _ConvT2dAct.apply(x, weight, bias, act)


class _MSELoss(torch.autograd.Function):
    """F.mse_loss(pred, target) with gradient to pred (the target is data)."""

    @staticmethod
    def forward(ctx, pred, target):
        require_cuda(pred, target)
        loss, dp = ops.mse_fwd(f32c(pred), f32c(target.detach()), want_grad=True)
        ctx.save_for_backward(dp)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dp,) = ctx.saved_tensors
        return ops.scale_by(dp, f32c(dloss).reshape(1)), None


def mse_loss(pred, target):
    return _MSELoss.apply(pred, target)


# ---------------------------------------------------------------- position-table conditioning
# `cond` is a function of the token's integer position alone, so the conditioning MLP and
# every ScaleLayer/ShiftLayer projection are evaluated once per distinct position (a (P,D)
# table) and the per-token consumers index the tables; see csrc/condtable.hip.  False
# re-enables the reference's per-token evaluation (tests compare the two).
USE_COND_TABLE = True
# the table form is taken when it has at least this many times fewer rows than there are
# tokens (the table is padded to whole 128-row tiles); tests set 0 to force it on tiny shapes
COND_TABLE_MIN_RATIO = 4
# grouped evaluation of the table projections: "all" (one launch for every decoder layer),
# "layer" (one per layer: gradients final in layer order, for the overlapped data-parallel
# all-reduce) or None = "layer" under torch.distributed with more than one rank, else "all"
COND_TABLE_GROUPING = os.environ.get("QARIG_COND_GROUPING") or None
# per-token cond (no table): its projections as grouped launches (CondTokens); QARIG_COND_GROUPS=0 disables
USE_COND_GROUPS = os.environ.get("QARIG_COND_GROUPS", "1") != "0"


class CondTable:
    """What the decoder blocks receive as `cond` in table form: `table` (P,D) = cond of
    position p (differentiable), `idx` int32 (M,) = position of token m, and the row map used
    to sum per-token gradients back into table rows.  `groups`: lists of nn.Linear modules
    (ScaleLayer/ShiftLayer of the decoder blocks) whose projections of the table are evaluated
    together as one grouped launch per list when the shapes allow."""

    def __init__(self, table, idx, shape, groups=()):
        self.table = table
        self.shape = shape                      # (N, S) of the token grid
        # positions outside [0, P) raise the device index flag here (the host turns it into an
        # IndexError at its next check, as for embedding ids); the consumers read through a
        # clamped copy so that such a batch cannot address memory outside the tables
        self.offsets, self.rows = ops.rowmap_build(idx, table.shape[0])
        self.idx = idx.clamp(0, table.shape[0] - 1)
        self._proj = {}
        self._group_of = {}
        P, D = table.shape
        for group in groups:
            group = list(group)
            # tables of up to 512 rows: the weight-streaming grouped kernel; longer ones (sequences of
            # more than ~500 tokens: BASELINE config 4's 1,025) the grouped 128 x 128-tile launches
            fits = (P <= 512 and D % 256 == 0) or ops.gemm_grouped_supported(P, D, D, any_precision=True)
            if (group and fits
                    and all(l.weight.shape == (D, D) and l.bias is not None for l in group)):
                for l in group:
                    self._group_of[id(l)] = group

    def ensure(self, linears):
        """Evaluates (and caches) the grouped projections that `linears` belong to.  Callers that
        wrap a layer in torch.utils.checkpoint call this BEFORE entering the checkpointed region:
        a projection first evaluated inside it would be computed (and its inputs saved) in the
        original forward but found cached by the recomputation, and non-reentrant checkpointing
        rejects a recomputation that saves a different number of tensors."""
        for l in linears:
            if id(l) in self._group_of:
                self.projection(l)

    def projection(self, linear):
        """linear(table) (P, out).  Linears registered in a group are evaluated together, as one
        grouped launch, the first time any of them is asked for -- i.e. inside the forward of the
        layer that owns them, so that in backward the group's gradient node runs right after that
        layer's and its parameters' gradients are final in layer order (what the overlapped
        data-parallel all-reduce buckets rely on)."""
        got = self._proj.get(id(linear))
        if got is not None:
            return got
        group = self._group_of.get(id(linear))
        if group is None:
            return linear_act(self.table, linear.weight, linear.bias)
        params = [t for l in group for t in (l.weight, l.bias)]
        outs = _TableProjections.apply(self.table, *params)
        for l, o in zip(group, outs):
            self._proj[id(l)] = o
        return self._proj[id(linear)]


class CondTokens:
    """Per-token `cond` (N,S,D) whose ScaleLayer / ShiftLayer projections are evaluated as grouped
    launches: every projection of a decoder (3 per block: AdaLN scale, shift and the residual gate,
    reference models/layers.py:100-153, 258-304) reads the same tensor, so up to 16 of them are one
    128 x 128-tile launch forward, one for the summed gradient of `cond` and one for their weight
    gradients (_TableProjections on the token rows) -- instead of one product, two gradient products
    and their split-K reduces per projection and an accumulation add per extra consumer of `cond`.
    Taken when the position table is not worth it (fewer than COND_TABLE_MIN_RATIO times fewer
    positions than tokens: BASELINE config 4's 1,025-token sequences at 8 x 256 tokens per GPU).
    Same `projection` / `ensure` interface as CondTable; the consumers are the per-token kernels."""

    def __init__(self, cond, groups=()):
        self.cond = cond
        self._flat = cond.reshape(-1, cond.shape[-1])
        self._proj = {}
        self._group_of = {}
        M, D = self._flat.shape
        for group in groups:
            group = list(group)
            if (group and ops.gemm_grouped_supported(M, D, D)
                    and all(l.weight.shape == (D, D) and l.bias is not None for l in group)):
                for l in group:
                    self._group_of[id(l)] = group

    def ensure(self, linears):
        for l in linears:
            if id(l) in self._group_of:
                self.projection(l)

    def projection(self, linear):
        got = self._proj.get(id(linear))
        if got is not None:
            return got
        group = self._group_of.get(id(linear))
        if group is None:
            return linear_act(self.cond, linear.weight, linear.bias)
        params = [t for l in group for t in (l.weight, l.bias)]
        outs = _TableProjections.apply(self._flat, *params)
        for l, o in zip(group, outs):
            self._proj[id(l)] = o.reshape(self.cond.shape)
        return self._proj[id(linear)]


class _TableProjections(torch.autograd.Function):
    """G projections of the same (P,D) table, out_g = table W_g^T + b_g.  Up to 512 rows: one grouped
    skinny launch; backward as two GEMMs over the concatenated gradients (d-table with K = G*D,
    d-weights as one (G*D, D) product) instead of 2G small ones.  Longer tables: grouped 128 x 128-tile
    launches of up to 16 members on the parameters where they lie (no stacked copy) -- forward, the
    d-table sum over the members, and the weight gradients with their bias row sums accumulated
    straight into the .grad slots."""

    @staticmethod
    def forward(ctx, table, *params):
        require_cuda(table)
        weights, biases = params[0::2], params[1::2]
        tab = f32c(table)
        P, D = tab.shape
        ctx.params = params
        ctx.tiled = not (P <= 512 and D % 256 == 0)
        if ctx.tiled:
            G = len(weights)
            out = torch.empty((G, P, D), dtype=torch.float32, device=tab.device)
            for c in range(0, G, ops.GEMM_MAX_GROUPS):
                n = min(ops.GEMM_MAX_GROUPS, G - c)
                ops.gemm_grouped([tab] * n, list(weights[c:c + n]), list(out[c:c + n].unbind(0)), P, D, D,
                                 bias=list(biases[c:c + n]), splitk=ops.grouped_splitk(n, P, D, D))
            ctx.save_for_backward(tab)
            return tuple(out.unbind(0))
        W = torch.stack([w.detach() for w in weights])             # (G, D, D)
        b = torch.stack([v.detach() for v in biases])               # (G, D)
        out = ops.gemm_grouped_skinny(tab, W, b, shared_a=True)     # (G, P, D)
        ctx.save_for_backward(tab, W)
        return tuple(out.unbind(0))

    @staticmethod
    def _backward_tiled(ctx, douts):
        (tab,) = ctx.saved_tensors
        P, D = tab.shape
        wparams, bparams = ctx.params[0::2], ctx.params[1::2]
        G = len(wparams)
        dev = tab.device
        cols = [f32c(d) if d is not None else torch.zeros((P, D), dtype=torch.float32, device=dev) for d in douts]
        wslots = [_grad_slot(p) for p in wparams]
        bslots = [_grad_slot(p) for p in bparams]
        inplace = all(s is not None for s in wslots) and all(s is not None for s in bslots)
        if not inplace:
            dW = torch.empty((G, D, D), dtype=torch.float32, device=dev)
            db = torch.empty((G, D), dtype=torch.float32, device=dev)
            wslots, bslots = list(dW.unbind(0)), list(db.unbind(0))
        dtab = torch.empty((P, D), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        for c in range(0, G, ops.GEMM_MAX_GROUPS):
            n = min(ops.GEMM_MAX_GROUPS, G - c)
            if dtab is not None:
                ops.gemm_grouped(cols[c:c + n], list(wparams[c:c + n]), [dtab], P, D, D, a_kcontig=True,
                                 b_kcontig=False, sum_groups=True, accumulate=c > 0,
                                 splitk=ops.grouped_splitk(n, P, D, D))
            ops.gemm_grouped(cols[c:c + n], [tab] * n, wslots[c:c + n], D, D, P, a_kcontig=False, b_kcontig=False,
                             accumulate=inplace, a_rowsum=bslots[c:c + n], splitk=ops.grouped_splitk(n, D, D, P))
        if inplace:
            for p in ctx.params:
                _report_done(p)
            return (dtab, *([None] * len(ctx.params)))
        grads = [dtab]
        for g in range(G):
            grads += [wslots[g], bslots[g]]
        return tuple(grads)

    @staticmethod
    def backward(ctx, *douts):
        if ctx.tiled:
            return _TableProjections._backward_tiled(ctx, douts)
        tab, W = ctx.saved_tensors
        G, D, _ = W.shape
        P = tab.shape[0]
        cols = [f32c(d) if d is not None else torch.zeros((P, D), dtype=torch.float32, device=tab.device)
                for d in douts]
        dcat = torch.cat(cols, dim=1)                               # (P, G*D)
        W2 = W.reshape(G * D, D)
        dtab = ops.gemm(dcat, W2, a_kcontig=True, b_kcontig=False) if ctx.needs_input_grad[0] else None
        dW = ops.gemm(dcat, tab, a_kcontig=False, b_kcontig=False,
                      splitk=ops.pick_splitk(G * D, D, P)).reshape(G, D, D)
        db = ops.colsum(dcat).reshape(G, D)
        grads = [dtab]
        wparams, bparams = ctx.params[0::2], ctx.params[1::2]
        wslots = [_grad_slot(p) for p in wparams]
        bslots = [_grad_slot(p) for p in bparams]
        if all(s is not None for s in wslots) and all(s is not None for s in bslots):
            torch._foreach_add_(wslots, list(dW.unbind(0)))
            torch._foreach_add_(bslots, list(db.unbind(0)))
            for p in ctx.params:
                _report_done(p)
            grads += [None] * len(ctx.params)
        else:
            for g in range(G):
                grads += [dW[g], db[g]]
        return tuple(grads)


class _LayerNormModTable(torch.autograd.Function):
    """AdaLNZero with scale/shift given per POSITION: y = scale_tab[idx] * LN(x) + shift_tab[idx]."""

    @staticmethod
    def forward(ctx, x, scale_tab, shift_tab, idx, offsets, rows, eps, with_skip=False):
        require_cuda(x, scale_tab, shift_tab)
        x2 = _2d(f32c(x))
        st = f32c(scale_tab)
        y, mean, rstd = ops.layernorm_fwd(x2, scale=st, shift=f32c(shift_tab), eps=eps, mod_idx=idx)
        ctx.save_for_backward(x2, st, mean, rstd, idx, offsets, rows)
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)
        return _with_skip(y.reshape(x.shape), x, with_skip)

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x2, st, mean, rstd, idx, offsets, rows = ctx.saved_tensors
        dy2, add = _skip_grads(dy, dskip, x2)
        dx, dscale_tok = ops.layernorm_bwd(dy2, x2, mean, rstd, scale=st, want_dy_xhat=True,
                                           mod_idx=idx, dx_add=add)
        dscale = ops.segment_sum(dscale_tok, offsets, rows)
        dshift = ops.segment_sum(dy2, offsets, rows)
        return dx.reshape(ctx.shape), dscale, dshift, None, None, None, None, None


def layernorm_mod_table(x, scale_tab, shift_tab, cond, eps=1e-5, with_skip=False):
    if _no_grad():
        require_cuda(x, scale_tab, shift_tab)
        y = ops.layernorm_fwd(_2d(f32c(x)), scale=f32c(scale_tab), shift=f32c(shift_tab), eps=eps,
                              mod_idx=cond.idx)[0].reshape(x.shape)
        return (y, x) if with_skip else y
    return _LayerNormModTable.apply(x, scale_tab, shift_tab, cond.idx, cond.offsets, cond.rows, eps, with_skip)


class _MulTable(torch.autograd.Function):
    """x * gate_tab[idx] (ResidualLinearLayer's scale multiply, gate given per position)."""

    @staticmethod
    def forward(ctx, x, gate_tab, idx, offsets, rows):
        require_cuda(x, gate_tab)
        x2 = _2d(f32c(x))
        gt = f32c(gate_tab)
        ctx.save_for_backward(x2, gt, idx, offsets, rows)
        return ops.mul_rows_fwd(x2, gt, idx).reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gt, idx, offsets, rows = ctx.saved_tensors
        dx, dg_tok = ops.mul_rows_bwd(_2d(f32c(dy)), x2, gt, idx)
        return dx.reshape(dy.shape), ops.segment_sum(dg_tok, offsets, rows), None, None, None


def mul_table(x, gate_tab, cond):
    if _no_grad():
        require_cuda(x, gate_tab)
        return ops.mul_rows_fwd(_2d(f32c(x)), f32c(gate_tab), cond.idx).reshape(x.shape)
    return _MulTable.apply(x, gate_tab, cond.idx, cond.offsets, cond.rows)
