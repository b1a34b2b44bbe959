"""The decode step's own kernels (csrc/decode.hip): the weight-streaming Linear against an fp64
contraction in every form the decoder blocks use it (reference models/layers.py:130-153 AdaLN /
nn.LayerNorm in front, :258-304 residual + gate behind, :389-418 grouped q/k/v MLPs)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ref(x, W, b, form, gam, bet, sc, sh, res, mul, act):
    xd = x.double()
    if xd.dim() == 2:
        xd = xd[None].expand(W.shape[0], -1, -1)
    if form != "none":
        h = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-5)
        h = h * gam.double() + bet.double() if form == "affine" else sc.double() * h + sh.double()
    else:
        h = xd
    y = torch.einsum("gmk,gnk->gmn", h, W.double())
    if b is not None:
        y = y + b.double()[:, None]
    if res is not None:
        y = y + res.double()
    if act == 1:
        y = torch.nn.functional.silu(y)
    if mul is not None:
        y = y * mul.double()
    return y


# every (rows, kernel shape) the launcher can pick: K/4 in {64, 128, 256} with 1/2/4 loads per thread,
# K = 2048 (two chunks per row) and 4096 (four), 4-row and 16-row register tiles, ragged N
@pytest.mark.parametrize("M,G,N,K", [
    (4, 1, 2048, 512), (4, 3, 2048, 512), (4, 1, 512, 2048), (4, 3, 512, 2048), (4, 1, 512, 512),
    (1, 1, 513, 2048), (3, 1, 513, 256), (2, 2, 77, 1024), (16, 1, 2048, 512), (16, 3, 512, 2048),
    (9, 1, 512, 512), (16, 1, 513, 2048), (5, 2, 40, 4096), (4, 1, 96, 4096), (16, 1, 8193, 2048),
    (4, 1, 8, 256), (13, 1, 1030, 1024)])
@pytest.mark.parametrize("form", ["none", "affine", "adaln", "adaln_row"])
def test_decode_linear_vs_fp64(M, G, N, K, form):
    from qarig import ops
    if form != "none" and K > 1024:
        assert not ops.decode_linear_supported(M, N, K, True)
        pytest.skip("LayerNorm prologue needs a workgroup to cover whole rows")
    assert ops.decode_linear_supported(M, N, K, form != "none")
    g = torch.Generator().manual_seed(M * 131 + N + K)
    x = (torch.randn((M, K), generator=g) * 1.5 + 0.3).cuda()
    W = (torch.randn((G, N, K), generator=g) * 0.05).cuda()
    b = torch.randn((G, N), generator=g).cuda()
    gam, bet = torch.randn(K, generator=g).cuda(), torch.randn(K, generator=g).cuda()
    rows = (K,) if form == "adaln_row" else (M, K)
    sc, sh = torch.randn(rows, generator=g).cuda(), torch.randn(rows, generator=g).cuda()
    kw = {"none": {}, "affine": dict(gamma=gam, beta=bet), "adaln": dict(scale=sc, shift=sh),
          "adaln_row": dict(scale=sc, shift=sh)}[form]
    f = "adaln" if form == "adaln_row" else form
    got = ops.decode_linear(x, W, b, act=1, **kw)
    assert got.shape == (G, M, N)
    assert rel_err(got, _ref(x, W, b, f, gam, bet, sc, sh, None, None, 1)) < 5e-6
    if G == 1:      # the residual layer's form: skip input added before the activation, gate behind
        res = torch.randn((M, N), generator=g).cuda()
        for mul in (torch.randn((M, N), generator=g).cuda(), torch.randn(N, generator=g).cuda()):
            got = ops.decode_linear(x, W[0], b[0], act=1, residual=res, mul=mul, **kw)
            assert got.shape == (M, N)
            assert rel_err(got, _ref(x, W, b, f, gam, bet, sc, sh, res, mul, 1)[0]) < 5e-6
        got = ops.decode_linear(x, W[0], None, act=0, **kw)
        assert rel_err(got, _ref(x, W, None, f, gam, bet, sc, sh, None, None, 0)[0]) < 5e-6
    else:           # per-group activations (the second layer of the q/k/v MLPs)
        xg = torch.randn((G, M, K), generator=g).cuda()
        if form == "none":
            got = ops.decode_linear(xg, W, b, act=1)
            assert rel_err(got, _ref(xg, W, b, "none", None, None, None, None, None, None, 1)) < 5e-6
    a = ops.decode_linear(x, W, b, act=1, **kw)
    assert torch.equal(a, ops.decode_linear(x, W, b, act=1, **kw)), "not run-to-run reproducible"


def test_decode_linear_rejects_what_it_cannot_run():
    from qarig import ops
    x = torch.randn(4, 512).cuda()
    W = torch.randn(8, 512).cuda()
    with pytest.raises(RuntimeError, match="M <= 16"):
        ops.decode_linear(torch.randn(17, 512).cuda(), W)
    with pytest.raises(RuntimeError, match="M <= 16"):
        ops.decode_linear(torch.randn(4, 768).cuda(), torch.randn(8, 768).cuda())
    with pytest.raises(RuntimeError, match="pair"):
        ops.decode_linear(x, W, gamma=torch.ones(512).cuda())
    with pytest.raises(RuntimeError, match="exclusive"):
        ops.decode_linear(x, W, gamma=torch.ones(512).cuda(), beta=torch.ones(512).cuda(),
                          scale=torch.ones(512).cuda(), shift=torch.ones(512).cuda())
    with pytest.raises(RuntimeError, match="M <= 16"):
        ops.decode_linear(torch.randn(4, 2048).cuda(), torch.randn(8, 2048).cuda(),
                          gamma=torch.ones(2048).cuda(), beta=torch.ones(2048).cuda())


def test_skinny_entry_points_route_small_batches_to_the_streaming_kernel():
    """qarig_gemm_f32 / _grouped_skinny / _skinny_ln hand their <= 16-row calls to decode_linear_kernel
    (option decode_stream); both kernels must agree to fp32 summation order."""
    from qarig import ops, _lib
    g = torch.Generator().manual_seed(5)
    x = torch.randn((4, 512), generator=g).cuda()
    W = (torch.randn((3, 2048, 512), generator=g) * 0.05).cuda()
    b = torch.randn((3, 2048), generator=g).cuda()
    sc, sh = torch.randn((4, 512), generator=g).cuda(), torch.randn((4, 512), generator=g).cuda()
    res = torch.randn((4, 2048), generator=g).cuda()
    lib = _lib.load()
    outs = {}
    for v in (1, 0):
        old = lib.qarig_set_option(b"decode_stream", v)
        try:
            outs[v] = (ops.gemm_grouped_skinny(x, W, b, act=1, shared_a=True),
                       ops.gemm_skinny_ln(x, W, b, act=1, scale=sc, shift=sh),
                       ops.gemm(x, W[0], bias=b[0], residual=res, act=1))
        finally:
            lib.qarig_set_option(b"decode_stream", old)
    for a, c in zip(outs[1], outs[0]):
        assert rel_err(a, c) < 2e-6
    assert torch.equal(outs[1][0], ops.decode_linear(x, W, b, act=1))
