"""Opt-in reduced-precision mode (qarig.ops.PRECISION = "bf16"; BASELINE config 5).
NOT the parity mode: the fp32 tests elsewhere hold the reference bar.  Here the bf16-MFMA
GEMM (csrc/gemm_lp.hip: bf16 operands in HBM, fp32 accumulation) is pinned against its own
definition - operands rounded to bf16 (round-to-nearest-even), exact products, fp32
accumulation - for which an fp64 contraction of the rounded operands is the reference
(tolerance 3e-6 * sqrt(K/512), accumulation order only); the fragment maps of both layouts are
checked on exact integer data; and a training step is checked to track the fp32 step within
the bf16 rounding budget (tolerances below)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16_mode():
    from qarig import ops
    old = ops.PRECISION
    ops.PRECISION = "bf16"
    yield ops
    ops.PRECISION = old


def _rounded(t):
    return t.bfloat16().double()


def test_cast_kernels_round_to_nearest_even():
    from qarig import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn((300, 264), generator=g).cuda()
    x[0, :4] = torch.tensor([1.0 + 2 ** -8, 1.0 + 3 * 2 ** -8, -1.0 - 2 ** -8, 65504.0])   # ties -> even
    assert torch.equal(ops.cast_bf16(x), x.bfloat16())
    assert torch.equal(ops.cast_transpose_bf16(x), x.bfloat16().t().contiguous())
    y = torch.randn((130, 70), generator=g).cuda()                # ragged tiles, tail kernel
    assert torch.equal(ops.cast_bf16(y), y.bfloat16())
    assert torch.equal(ops.cast_transpose_bf16(y), y.bfloat16().t().contiguous())
    v = ops.cast_transpose_bf16(x[:, :128])                       # strided source
    assert torch.equal(v, x[:, :128].bfloat16().t().contiguous())


@pytest.mark.parametrize("M,N,K", [(256, 384, 192), (4096, 4096, 192), (2048, 512, 2048)])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_lp_fragment_maps_on_exact_integer_data(layout, M, N, K):
    """A = a permutation-like integer matrix, B asymmetric small integers: every product and sum
    is exact in bf16/fp32, so any wrong lane map, swizzle or transposed read shows as a wrong
    integer (guide: 'A = I check with ASYMMETRIC B').  4096 x 4096 reaches the 256 x 256-tile
    kernel (>= 224 of its tiles), 2048 x 512 x 2048 reaches it through split-K."""
    from qarig import ops
    g = torch.Generator().manual_seed(3)
    A = torch.zeros((M, K))
    A[torch.arange(M), torch.randint(0, K, (M,), generator=g)] = 1.0
    A[torch.arange(M), torch.randint(0, K, (M,), generator=g)] += 2.0
    B = torch.randint(-8, 9, (N, K), generator=g).float() + torch.arange(N)[:, None] % 5
    ref = A.double() @ B.double().t()
    if layout == 0:      # NT
        Ab, Bb = A.cuda().bfloat16(), B.cuda().bfloat16()
    elif layout == 1:    # TN
        Ab, Bb = A.t().contiguous().cuda().bfloat16(), B.t().contiguous().cuda().bfloat16()
    else:                # NN: A reduction-contiguous, B reduction-major
        Ab, Bb = A.cuda().bfloat16(), B.t().contiguous().cuda().bfloat16()
    C = torch.empty((M, N), device="cuda")
    Cb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ops.gemm_lp(Ab, Bb, layout, M, N, K, C=C, Cb=Cb)
    assert torch.equal(C.double().cpu(), ref)
    assert torch.equal(Cb.double().cpu(), ref.bfloat16().double())
    C2 = torch.empty((M, N), device="cuda")
    ops.gemm_lp(Ab, Bb, layout, M, N, K, C=C2, splitk=32 if K == 2048 else 3)
    assert torch.equal(C2.double().cpu(), ref)


def test_lp_epilogue_bf16_copies():
    from qarig import ops
    from oracle import ref_models as rm
    g = torch.Generator().manual_seed(4)
    M, N, K = 128, 256, 128
    A, W = torch.randn((M, K), generator=g).cuda(), (torch.randn((N, K), generator=g) * 0.1).cuda()
    b = torch.randn(N, generator=g).cuda()
    C = torch.empty((M, N), device="cuda")
    P = torch.empty((M, N), device="cuda")
    Cb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    Pb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ops.gemm_lp(ops.cast_bf16(A), ops.cast_bf16(W), 0, M, N, K, C=C, bias=b, preact=P, act=1, Cb=Cb, Pb=Pb)
    t = _rounded(A.cpu()) @ _rounded(W.cpu()).t() + b.double().cpu()
    assert rel_err(P, t) < 3e-6 and rel_err(C, rm.activation(t, "silu")) < 5e-6
    assert torch.equal(Cb, C.bfloat16()) and torch.equal(Pb, P.bfloat16())
    only = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)     # bf16-only output
    ops.gemm_lp(ops.cast_bf16(A), ops.cast_bf16(W), 0, M, N, K, bias=b, act=1, Cb=only)
    assert torch.equal(only, Cb)


@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("M,N,K,splitk", [(128, 128, 64, 1), (256, 384, 512, 1), (128, 256, 2048, 4),
                                          (384, 128, 4096, 8)])
def test_bf16_gemm_is_exact_on_rounded_operands(bf16_mode, M, N, K, splitk, ak, bk):
    ops = bf16_mode
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((M, K) if ak else (K, M), generator=g).cuda()
    B = torch.randn((N, K) if bk else (K, N), generator=g).cuda()
    C = ops.gemm(A, B, ak, bk, splitk=splitk)
    Ar, Br = _rounded(A.cpu()), _rounded(B.cpu())
    ref = (Ar if ak else Ar.t()) @ (Br.t() if bk else Br)
    tol = 3e-6 * max(1.0, K / 512) ** 0.5
    assert rel_err(C, ref) < tol
    exact = (A.double().cpu() if ak else A.double().cpu().t()) @ \
            (B.double().cpu().t() if bk else B.double().cpu())
    e = rel_err(C, exact)
    assert 1e-5 < e < 2e-2, e            # the rounding is really there, and is bf16-sized
    assert torch.equal(C, ops.gemm(A, B, ak, bk, splitk=splitk))      # deterministic


def test_bf16_classifier_products_at_config5_shapes(bf16_mode):
    """BASELINE configs[4]'s largest products at their own extents -- the classifier over the 8192-entry
    codebook (+ <end>, padded to 8320 columns) on a shard of 8 x 4096 tokens: forward x W^T
    (32768 x 8320 x 2048), input gradient dY W (reduction 8320) and weight gradient dY^T x (reduction 32768) --
    against the fp64 contraction of the bf16-rounded operands (reference models/Transformer.py:193-200)."""
    ops = bf16_mode
    g = torch.Generator(device="cuda").manual_seed(11)
    M, N, K = 32768, 8320, 2048
    x = torch.randn((M, K), generator=g, device="cuda")
    W = torch.randn((N, K), generator=g, device="cuda") * 0.05
    dY = torch.randn((M, N), generator=g, device="cuda")
    r = lambda t: t.bfloat16().double()
    cases = ((x, W, True, True, K, lambda: r(x) @ r(W).t()),                    # forward
             (dY, W, True, False, N, lambda: r(dY) @ r(W)),                      # input gradient
             (dY, x, False, False, M, lambda: r(dY).t() @ r(x)))                 # weight gradient
    for A, B, ak, bk, red, ref in cases:
        C = ops.gemm(A, B, ak, bk)
        want = ref()
        assert C.shape == want.shape
        err = float((C.double() - want).abs().max() / want.abs().max())
        assert err < 3e-6 * (red / 512) ** 0.5, (ak, bk, err)
        del C, want


def test_bf16_gemm_epilogues_and_accumulate(bf16_mode):
    ops = bf16_mode
    from oracle import ref_models as rm
    g = torch.Generator().manual_seed(0)
    M, N, K = 256, 256, 512
    A = torch.randn((M, K), generator=g).cuda()
    W = (torch.randn((N, K), generator=g) * 0.1).cuda()
    b = torch.randn((N,), generator=g).cuda()
    R = torch.randn((M, N), generator=g).cuda()
    Z = torch.randn((M, N), generator=g).cuda()
    t = _rounded(A.cpu()) @ _rounded(W.cpu()).t()
    C, pre = ops.gemm(A, W, bias=b, residual=R, want_preact=True, act=1)
    tt = t + b.double().cpu() + R.double().cpu()
    assert rel_err(pre, tt) < 3e-6
    assert rel_err(C, rm.activation(tt, "silu")) < 5e-6
    Zd = Z.double().cpu().requires_grad_(True)
    rm.activation(Zd, "silu").sum().backward()
    assert rel_err(ops.gemm(A, W, gradz=Z, gact=1), t * Zd.grad) < 5e-6
    acc = R.clone()
    ops.gemm(A, W, out=acc, accumulate=True)
    assert rel_err(acc, t + R.double().cpu()) < 3e-6
    acc = R.clone()
    ops.gemm(A, W, out=acc, accumulate=True, splitk=4)
    assert rel_err(acc, t + R.double().cpu()) < 3e-6
    # bias gradient riding on a weight-gradient GEMM: separate fp32 column sum in this mode
    dT = torch.randn((1024, 256), generator=g).cuda()
    X = torch.randn((1024, 128), generator=g).cuda()
    rs = torch.zeros(256, device="cuda")
    dW = ops.gemm(dT, X, False, False, a_rowsum=rs, splitk=2)
    assert rel_err(dW, _rounded(dT.cpu()).t() @ _rounded(X.cpu())) < 3e-6 * 2 ** 0.5
    assert rel_err(rs, dT.double().cpu().sum(0)) < 2e-6


def test_lp_big_tile_kernel_epilogues():
    """The 256 x 256-tile kernel (taken from 224 of its tiles up: 4096 x 4096 here) through every
    epilogue option, all three layouts: bias + residual + SiLU + saved pre-activation + bf16
    copies (NT), the act' fusion on a bf16 pre-activation (NN), accumulate (TN)."""
    from qarig import ops
    from oracle import ref_models as rm
    g = torch.Generator().manual_seed(5)
    M, N, K = 4096, 4096, 256
    A = torch.randn((M, K), generator=g)
    W = torch.randn((N, K), generator=g) * 0.1
    b = torch.randn((N,), generator=g)
    R = torch.randn((M, N), generator=g)
    Z = torch.randn((M, N), generator=g)
    Ab, Wb = A.cuda().bfloat16(), W.cuda().bfloat16()
    t = _rounded(A) @ _rounded(W).t()
    C = torch.empty((M, N), device="cuda")
    pre = torch.empty((M, N), device="cuda")
    Cb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    Pb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ops.gemm_lp(Ab, Wb, 0, M, N, K, C=C, bias=b.cuda(), residual=R.cuda(), preact=pre, act=1, Cb=Cb, Pb=Pb)
    tt = t + b.double() + R.double()
    assert rel_err(pre, tt) < 3e-6
    assert rel_err(C, rm.activation(tt, "silu")) < 5e-6
    assert torch.equal(Cb, C.bfloat16()) and torch.equal(Pb, pre.bfloat16())
    # NN: B reduction-major; C *= silu'(z) with z given in bf16
    Zb = Z.cuda().bfloat16()
    G = torch.empty((M, N), device="cuda")
    ops.gemm_lp(Ab, Wb.t().contiguous(), 2, M, N, K, C=G, gradz=Zb, gact=1)
    Zd = Zb.double().cpu().requires_grad_(True)
    rm.activation(Zd, "silu").sum().backward()
    assert rel_err(G, t * Zd.grad) < 5e-6
    # TN: both operands reduction-major, accumulated into an existing tensor
    acc = R.cuda().clone()
    ops.gemm_lp(Ab.t().contiguous(), Wb.t().contiguous(), 1, M, N, K, C=acc, accumulate=True)
    assert rel_err(acc, t + R.double()) < 3e-6


def test_bf16_mode_falls_back_to_fp32_kernels_off_the_interior(bf16_mode):
    ops = bf16_mode
    g = torch.Generator().manual_seed(1)
    A = torch.randn((100, 96), generator=g).cuda()      # ragged M, K % 32 == 0
    W = torch.randn((513, 96), generator=g).cuda()
    got = ops.gemm(A, W)
    ops.PRECISION = "f32"
    assert torch.equal(got, ops.gemm(A, W))
    ops.PRECISION = "bf16"
    with pytest.raises(ValueError):
        ops.PRECISION = "int4"
        ops.gemm(A, W)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm()))


@pytest.mark.parametrize("N_out,act2", [(512, 0), (512, 1), (513, 0)])
def test_lp_mlp_nodes_track_fp32_nodes(bf16_mode, N_out, act2):
    """The bf16-storage MLP node (h, t1, dT1 exist only in bf16; bias gradients from the fused
    cast + column-sum pass; padded ragged output width) against the fp32 node from the same
    weights: outputs and every gradient within the bf16 rounding budget (relative error of
    the tensor < 1e-2, cosine > 0.9999)."""
    ops = bf16_mode
    from qarig import functional as QF
    g = torch.Generator().manual_seed(7)
    M, K, H = 2048, 512, 2048
    x = torch.randn((4, M // 4, K), generator=g).cuda()
    w1, b1 = (torch.randn((H, K), generator=g) * 0.04).cuda(), (torch.randn(H, generator=g) * 0.1).cuda()
    w2, b2 = (torch.randn((N_out, H), generator=g) * 0.02).cuda(), (torch.randn(N_out, generator=g) * 0.1).cuda()
    dy = torch.randn((4, M // 4, N_out), generator=g).cuda()
    res = {}
    for mode in ("f32", "bf16"):
        ops.PRECISION = mode
        leaves = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        y = QF.mlp2(*leaves, 1, act2)
        if mode == "bf16":
            assert type(y.grad_fn).__name__ == "_MLP2LPBackward"
        (y * dy).sum().backward()
        res[mode] = [y.detach()] + [t.grad for t in leaves]
    ops.PRECISION = "bf16"
    for a, b in zip(res["bf16"], res["f32"]):
        assert rel_err(a, b) < 1.5e-2 and _cos(a, b) > 0.9999


def test_lp_qkv_and_residual_nodes_track_fp32_nodes(bf16_mode):
    ops = bf16_mode
    from qarig import functional as QF
    g = torch.Generator().manual_seed(8)
    M, K, H = 2048, 512, 1024
    x = torch.randn((2, M // 2, K), generator=g).cuda()
    params = []
    for _ in range(3):
        params.append(((torch.randn((H, K), generator=g) * 0.04).cuda(), (torch.randn(H, generator=g) * 0.1).cuda(),
                       (torch.randn((K, H), generator=g) * 0.03).cuda(), (torch.randn(K, generator=g) * 0.1).cuda()))
    dys = [torch.randn((2, M // 2, K), generator=g).cuda() for _ in range(3)]
    wr, br = (torch.randn((K, K), generator=g) * 0.04).cuda(), (torch.randn(K, generator=g) * 0.1).cuda()
    skip = torch.randn((2, M // 2, K), generator=g).cuda()
    res = {}
    for mode in ("f32", "bf16"):
        ops.PRECISION = mode
        xl = x.clone().requires_grad_(True)
        pl = [tuple(t.clone().requires_grad_(True) for t in p) for p in params]
        q, k, v = QF.mlp2x3(xl, pl, 1, 0)
        (q * dys[0] + k * dys[1] + v * dys[2]).sum().backward()
        out = [q.detach(), k.detach(), v.detach(), xl.grad] + [t.grad for p in pl for t in p]
        xr, sl = x.clone().requires_grad_(True), skip.clone().requires_grad_(True)
        wl, bl = wr.clone().requires_grad_(True), br.clone().requires_grad_(True)
        z = QF.linear_act(xr, wl, bl, residual=sl, act=1)
        (z * dys[0]).sum().backward()
        out += [z.detach(), xr.grad, sl.grad, wl.grad, bl.grad]
        res[mode] = out
    ops.PRECISION = "bf16"
    for a, b in zip(res["bf16"], res["f32"]):
        assert rel_err(a, b) < 1.5e-2 and _cos(a, b) > 0.9999


@pytest.mark.parametrize("N,Sq,Sk,H,d,causal", [(2, 300, 300, 6, 8, True), (1, 1024, 1024, 8, 8, True),
                                                (2, 200, 333, 5, 8, False), (1, 130, 130, 4, 16, True),
                                                (1, 96, 96, 2, 64, False),
                                                # BASELINE configs[4]: one 4096-token sequence x 64 heads of 8, and its
                                                # cross-attention over the previous stage's 1024 tokens
                                                (1, 4096, 4096, 64, 8, True), (1, 4096, 1024, 64, 8, False)])
def test_lp_attention_tracks_fp64(bf16_mode, N, Sq, Sk, H, d, causal):
    """Attention with the QK^T / PV (and backward) products on the bf16 MFMA, fp32 softmax and
    accumulation: against fp64 attention of the bf16-ROUNDED q, k, v the remaining error is the
    rounding of the probabilities / score gradients to bf16 (2^-9 relative each) -- output and
    all three gradients within 1e-2 of the tensor's largest value, cosine > 0.9995."""
    ops = bf16_mode
    from qarig import functional as QF
    g = torch.Generator().manual_seed(Sq + Sk + H + d)
    D = H * d
    q, k, v = (torch.randn((N, S, D), generator=g).cuda() for S in (Sq, Sk, Sk))
    do = torch.randn((N, Sq, D), generator=g).cuda()

    def ref(q, k, v):
        qh = q.reshape(N, Sq, H, d).permute(0, 2, 1, 3)
        kh = k.reshape(N, Sk, H, d).permute(0, 2, 1, 3)
        vh = v.reshape(N, Sk, H, d).permute(0, 2, 1, 3)
        s = qh @ kh.transpose(-1, -2) / (d ** 0.5)
        if causal:
            s = s.masked_fill(torch.triu(torch.ones(Sq, Sk, dtype=torch.bool, device=q.device), 1), float("-inf"))
        return (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(N, Sq, D)

    a = [t.bfloat16().double().requires_grad_(True) for t in (q, k, v)]
    oa = ref(*a)
    (oa * do.bfloat16().double()).sum().backward()
    b = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o = QF.attention(b[0], b[1], b[2], H, causal)
    (o * do).sum().backward()
    assert rel_err(o, oa) < 1e-2 and _cos(o, oa) > 0.9995
    for x, y in zip(b, a):
        assert rel_err(x.grad, y.grad) < 1.5e-2 and _cos(x.grad, y.grad) > 0.9995
    ops.PRECISION = "f32"            # and the fp32 kernels are untouched by the mode switch
    o32 = QF.attention(q, k, v, H, causal)
    ops.PRECISION = "bf16"
    assert not torch.equal(o32, o.detach()) and rel_err(o32, ref(q.double(), k.double(), v.double())) < 2e-6


def test_bf16_train_step_tracks_fp32_step():
    """One README-shaped (narrower/shallower) training step in both modes from the same
    weights: loss within 2e-3 relative, flat gradient cosine similarity > 0.999."""
    from models.Transformer import Transformer
    from qarig import ops, pipeline
    from qarig.optim import FlatAdam
    res = {}
    for mode in ("f32", "bf16"):
        torch.manual_seed(2)
        m = Transformer(use_encoder=False, use_pos_cond=True, num_enc_layers=None, num_dec_layers=2,
                        num_enc_embedding=None, num_dec_embedding=1024, self_attn_heads=16,
                        cross_attn_heads=None, transformer_in_dim=256, transformer_out_dim=513,
                        transformer_hidden_dim=1024).cuda()
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for p in m.parameters():
                if p.abs().max() == 0:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        opt = FlatAdam(m.parameters(), lr=1e-4, betas=(0.5, 0.999))
        x = torch.randint(0, 1024, (8, 128), generator=g).cuda()
        t = torch.randint(0, 513, (8, 128), generator=g).cuda()
        pos = torch.arange(128)[None].repeat(8, 1).cuda()
        old, ops.PRECISION = ops.PRECISION, mode
        try:
            opt.zero_grad()
            loss = pipeline.train_step(m, opt, x, None, t, pos, dp=False)
        finally:
            ops.PRECISION = old
        res[mode] = (float(loss), opt.flat_grad.detach().clone())
    lf, gf = res["f32"]
    lb, gb = res["bf16"]
    assert abs(lb - lf) < 2e-3 * abs(lf)
    cos = float((gf * gb).sum() / (gf.norm() * gb.norm()))
    assert cos > 0.999, cos
    assert not torch.equal(gf, gb)


def test_weight_shadows_follow_captured_graph_replays(bf16_mode):
    """Graph-replayed training (pipeline.GraphedTrainStep) rewrites the parameters from a HIP kernel
    without touching torch's version counters: an eager no_grad forward between replays (the
    per-checkpoint sampling of train_quantized_transformer.py) must see the CURRENT weights, not the
    bf16 shadows it cached at an earlier evaluation.  Reference of each evaluation: a second model
    loaded with the current fp32 weights (its own, fresh shadows)."""
    from models.Codebook import Codebook
    from models.Transformer import Transformer
    from qarig import pipeline
    from qarig.optim import FlatAdam

    def build_model():
        torch.manual_seed(3)
        m = Transformer(use_encoder=False, use_pos_cond=True, num_enc_layers=None, num_dec_layers=1,
                        num_enc_embedding=None, num_dec_embedding=128 + 128, self_attn_heads=16,
                        cross_attn_heads=None, transformer_in_dim=128, transformer_out_dim=129,
                        transformer_hidden_dim=256).cuda()
        g = torch.Generator().manual_seed(6)
        with torch.no_grad():
            for p in m.parameters():
                if p.abs().max() == 0:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        return m

    g = torch.Generator().manual_seed(4)
    lr_cb = Codebook(patch_dim=(16, 16), image_dim=(16, 16), image_channel=4, num_embeddings=128).cuda()
    hr_cb = Codebook(patch_dim=(1, 1), image_dim=(16, 16), image_channel=4, num_embeddings=128).cuda()
    with torch.no_grad():
        lr_cb.codebook.weight.copy_(torch.tanh(torch.randn((128, 1024), generator=g)))
        hr_cb.codebook.weight.copy_(torch.tanh(torch.randn((128, 4), generator=g)))
    m = build_model()
    opt = FlatAdam(m.parameters(), lr=1e-2, betas=(0.5, 0.999))
    step = pipeline.GraphedTrainStep(m, opt, lr_cb, hr_cb, True, 128, warmup=1)
    xe = torch.randint(0, 256, (4, 128), generator=g).cuda()
    pe = torch.arange(128)[None].repeat(4, 1).cuda()

    def evaluate(model):
        with torch.no_grad():
            return model(xe, None, pe, pos_bound=257).clone()

    def fresh_copy_eval():
        twin = build_model()
        twin.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
        return evaluate(twin)

    outs = []
    for it in range(4):
        z = torch.tanh(torch.randn((4, 4, 16, 16), generator=g)).cuda()
        rand = torch.randint(0, 257 - 128 + 1, (4,), generator=g)
        step(z, rand)
        got = evaluate(m)          # caches the weight shadows of THIS moment
        assert torch.equal(got, fresh_copy_eval()), f"stale shadows after step {it}"
        outs.append(got)
    assert step.graph is not None
    assert not torch.equal(outs[-1], outs[-2])      # the replays did move the weights


def test_adam_pass_writes_the_bf16_weight_image(bf16_mode):
    """In a reduced-precision mode the Adam kernel leaves the updated parameters rounded to bf16 in one flat image
    (FlatAdam.flat_shadow); the Linear nodes read their weights from it instead of casting every weight after
    every step.  The image equals round-to-nearest-even of the fp32 masters bit for bit; a parameter somebody
    else modifies in place, and a step taken outside the mode, fall back to the cast of the current weights."""
    from qarig import ops, functional_lp, _lib
    from qarig.optim import FlatAdam
    torch.manual_seed(2)
    lin = [torch.nn.Linear(256, 512), torch.nn.Linear(512, 129), torch.nn.Linear(129, 256)]    # a ragged bias in the middle
    m = torch.nn.Sequential(*lin).cuda()
    opt = FlatAdam(m.parameters(), lr=1e-2, betas=(0.5, 0.999))
    ps = list(m.parameters())
    assert all(o % 8 == 0 for o in opt.offsets)
    assert opt.shadow_of(ps[0]) is None                              # no pass yet
    for p in ps:
        p.grad.copy_(torch.randn_like(p))
    opt.step()
    for p in ps:
        v = opt.shadow_of(p)
        assert v is not None and v.dtype == torch.bfloat16 and v.shape == p.shape and v.data_ptr() % 16 == 0
        assert torch.equal(v, p.detach().to(torch.bfloat16))
    w = ps[0]
    n0 = _lib.N_CALLS
    sh = functional_lp._shadow(w)
    assert sh.data_ptr() == opt.shadow_of(w).data_ptr() and _lib.N_CALLS == n0        # no cast launch
    with torch.no_grad():
        w.mul_(2.0)                                                  # somebody else's in-place change
    assert opt.shadow_of(w) is None
    assert torch.equal(functional_lp._shadow(w), w.detach().to(torch.bfloat16))
    opt.step()                                                       # the next pass serves it again
    assert torch.equal(opt.shadow_of(w), w.detach().to(torch.bfloat16))
    ops.set_precision("f32")
    try:
        opt.step()
    finally:
        ops.set_precision("bf16")
    assert opt.shadow_of(w) is None                                  # image left behind by an fp32 step
    assert torch.equal(functional_lp._shadow(w), w.detach().to(torch.bfloat16))
