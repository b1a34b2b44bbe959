"""Backward of the conv layers (autograd of nn.Conv2d / nn.ConvTranspose2d as the
reference's autoencoder trains them, train_autoencoder.py:203-226), on the same
implicit-GEMM MFMA core as the forward:
  d(input)  = a stride-1 conv per output-parity class (Conv2d), or a k=4/s=2 Conv2d
              over the gradient whose weight matrix is the ConvTranspose weight as stored;
  d(weight) = pixels-reduced correlation, split over the pixel axis (fp32 slabs,
              fixed summation order);
  d(bias)   = per-channel sum.
"""
import torch

from . import _lib, ops
from ._lib import check, f32c, ptr, stream, workspace


def _bias_grad(dT):
    N, C, H, W = dT.shape
    db = torch.empty(C, dtype=torch.float32, device=dT.device)
    check(_lib.load().qarig_conv_bias_grad(ptr(dT), N, C, H * W, ptr(db), stream()),
          "qarig_conv_bias_grad")
    return db


def _wgrad(G, X, k, stride, pad, out_shape):
    N, Cg, Gh, Gw = G.shape
    _, Cx, H, W = X.shape
    dw = torch.empty(out_shape, dtype=torch.float32, device=G.device)
    lib = _lib.load()
    ws = workspace(lib.qarig_conv_wgrad_workspace_bytes(Cg, Cx * k * k, N * Gh * Gw), G.device, "wgrad")
    check(lib.qarig_conv_wgrad(ptr(G), N, Cg, Gh, Gw, ptr(X), Cx, H, W, k, stride, pad, ptr(dw),
                               ptr(ws), ws.numel(), stream()), "qarig_conv_wgrad")
    return dw


def conv2d_bwd(ctx, dy, x, weight, pre, stride, pad, act, has_bias):
    dT = f32c(dy)
    if act:
        dT = ops.act_bwd(dT, pre, act)
    N, Cin, H, W = x.shape
    Cout, _, k, _ = weight.shape
    _, _, Ho, Wo = dT.shape
    dx = dw = db = None
    lib = _lib.load()
    if ctx.needs_input_grad[0]:
        dx = torch.empty_like(x)
        ws = workspace(lib.qarig_conv2d_bwd_data_workspace_bytes_n(N, Cin, H, W, Cout, k, stride), x.device, "convbwd")
        check(lib.qarig_conv2d_bwd_data(ptr(dT), N, Cout, Ho, Wo, ptr(weight), Cin, k, stride, pad, H,
                                        W, ptr(dx), ptr(ws), ws.numel(), stream()),
              "qarig_conv2d_bwd_data")
    if ctx.needs_input_grad[1]:
        dw = _wgrad(dT, x, k, stride, pad, weight.shape)
    if has_bias and ctx.needs_input_grad[2]:
        db = _bias_grad(dT)
    return dx, dw, db


def conv_transpose2d_bwd(ctx, dy, x, weight, pre, act, has_bias):
    dT = f32c(dy)
    if act:
        dT = ops.act_bwd(dT, pre, act)
    N, Cin, H, W = x.shape
    _, Cout, _, _ = weight.shape
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = torch.empty_like(x)
        lib = _lib.load()
        ws = workspace(lib.qarig_conv_transpose2d_bwd_data_workspace_bytes_n(N, Cin, H, W, Cout), x.device, "convbwd")
        check(lib.qarig_conv_transpose2d_bwd_data_ws(ptr(dT), N, Cout, H, W, ptr(weight), Cin, ptr(dx), ptr(ws),
                                                     ws.numel(), stream()),
              "qarig_conv_transpose2d_bwd_data_ws")
    if ctx.needs_input_grad[1]:
        dw = _wgrad(x, dT, 4, 2, 1, weight.shape)
    if has_bias and ctx.needs_input_grad[2]:
        db = _bias_grad(dT)
    return dx, dw, db
