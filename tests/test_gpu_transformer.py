"""GPU parity of the Transformer building blocks and of the whole model (forward,
backward, Adam) against the oracle and the reference's golden vectors."""

import pytest
import torch

from conftest import grad_err, load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5  # max|err| / max|ref|: the fp32 tolerance SURVEY 8d states for logits/attention


def test_posemb_matches_reference_goldens():
    """vs the oracle on this host: sinf/cosf accuracy only (the frequency table comes
    from the same host torch ops).  vs the golden file (made on another CPU): the
    reference's own fp32 `exp` differs by an ulp between hosts, which the angle
    pos*f amplifies by |pos| -- so that bound scales with the position."""
    from qarig import ops
    from oracle import ref_models as rm
    g = load_golden("layers")
    cases = [(torch.arange(1, 18), 32, "pos_int"),
             (torch.arange(0, 300, 7, dtype=torch.float32), 32, "pos_float"),
             (torch.tensor([1, 2, 255, 256, 1023, 4096]), 512, "pos_int_512")]
    for pos, dim, key in cases:
        got = ops.posemb(pos.cuda(), dim).cpu()
        assert (got - rm.positional_embeddings(dim, pos)).abs().max() < 2e-7
        bound = 2e-7 + 1.2e-7 * pos.abs().max().item()
        assert (got - g[key]).abs().max() < bound


@pytest.mark.parametrize("M,D", [(7, 32), (300, 512), (64, 100)])
def test_layernorm_forms(M, D):
    from qarig import functional as QF
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn((M, D), generator=g) * 2 + 0.5
    gam, bet = torch.randn(D, generator=g), torch.randn(D, generator=g)
    sc, sh = torch.randn((M, D), generator=g), torch.randn((M, D), generator=g)
    dy = torch.randn((M, D), generator=g)

    def run(dev, dt):
        xs = [t.to(dev, dt).requires_grad_(True) for t in (x, gam, bet, sc, sh)]
        if dev == "cuda":
            ya = QF.layernorm_affine(xs[0], xs[1], xs[2])
            ym = QF.layernorm_mod(xs[0], xs[3], xs[4])
        else:
            ya = torch.nn.functional.layer_norm(xs[0], (D,), xs[1], xs[2])
            ym = xs[3] * torch.nn.functional.layer_norm(xs[0], (D,)) + xs[4]
        ((ya + 2 * ym) * dy.to(dev, dt)).sum().backward()
        return [ya, ym] + [t.grad for t in xs]

    got, ref = run("cuda", torch.float32), run("cpu", torch.float64)
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 5e-6


@pytest.mark.parametrize("M,D", [(7, 32), (300, 512), (64, 100)])
def test_layernorm_with_skip_adds_both_gradients_of_x(M, D):
    """with_skip nodes return (y, alias of x): the skip connection's gradient comes back into the
    node and the LayerNorm backward kernel adds it (qarig_layernorm_bwd dx_add) -- same dx as the
    two-consumer graph that autograd would accumulate itself; a consumer may ignore either output."""
    from qarig import functional as QF
    g = torch.Generator().manual_seed(M * 3 + D)
    x = (torch.randn((M, D), generator=g) * 2 + 0.5).cuda()
    gam, bet = torch.randn(D, generator=g).cuda(), torch.randn(D, generator=g).cuda()
    sc, sh = torch.randn((M, D), generator=g).cuda(), torch.randn((M, D), generator=g).cuda()
    dy, ds = torch.randn((M, D), generator=g).cuda(), torch.randn((M, D), generator=g).cuda()
    for form in ("affine", "mod"):
        def node(xx, with_skip):
            if form == "affine":
                return QF.layernorm_affine(xx, gam, bet, with_skip=with_skip)
            return QF.layernorm_mod(xx, sc, sh, with_skip=with_skip)
        xa = x.clone().requires_grad_(True)
        ya = node(xa, False)
        ((ya * dy).sum() + (xa * ds).sum()).backward()          # autograd adds the two gradients
        xb = x.clone().requires_grad_(True)
        yb, xs = node(xb, True)
        assert xs.data_ptr() == xb.data_ptr() and torch.equal(ya, yb)
        ((yb * dy).sum() + (xs * ds).sum()).backward()
        assert rel_err(xb.grad, xa.grad) < 1e-6
        xc = x.clone().requires_grad_(True)                     # skip output unused
        yc, _ = node(xc, True)
        (yc * dy).sum().backward()
        xd = x.clone().requires_grad_(True)
        (node(xd, False) * dy).sum().backward()
        assert torch.equal(xc.grad, xd.grad)
        xe = x.clone().requires_grad_(True)                     # only the skip output used
        _, xs = node(xe, True)
        (xs * ds).sum().backward()
        assert rel_err(xe.grad, ds) < 1e-6


@pytest.mark.parametrize("N,Sq,Sk,H,d,causal", [(2, 12, 12, 4, 8, True), (3, 70, 70, 2, 16, True),
                                                (2, 256, 256, 8, 8, True), (2, 33, 5, 4, 8, False),
                                                (1, 130, 130, 2, 64, False), (2, 9, 1, 2, 4, False),
                                                (1, 64, 200, 3, 32, False),
                                                # head dims without an instantiation run zero-padded on the next one
                                                (2, 70, 70, 4, 12, True), (1, 40, 90, 3, 24, False),
                                                (2, 33, 33, 2, 48, True), (2, 20, 20, 5, 2, True),
                                                (1, 17, 17, 6, 1, False), (1, 50, 50, 2, 60, True),
                                                # head dims 65 ... 128 (a 512-wide model with 4 heads): the wide-head
                                                # family, directly and zero-padded; tiles that are not whole, cross
                                                (2, 70, 70, 4, 128, True), (1, 200, 200, 2, 128, True),
                                                (2, 33, 90, 3, 128, False), (1, 130, 130, 2, 96, True),
                                                (1, 64, 5, 2, 72, False), (1, 65, 65, 1, 100, True)])
def test_attention_fwd_bwd(N, Sq, Sk, H, d, causal):
    """The reference's attention takes any heads | in_dim (models/layers.py:433-474)."""
    from qarig import functional as QF
    g = torch.Generator().manual_seed(Sq * 3 + Sk)
    D = H * d
    q, k, v = (torch.randn((N, S, D), generator=g) for S in (Sq, Sk, Sk))
    do = torch.randn((N, Sq, D), generator=g)

    def ref(q, k, v):
        qh = q.reshape(N, Sq, H, d).permute(0, 2, 1, 3)
        kh = k.reshape(N, Sk, H, d).permute(0, 2, 1, 3)
        vh = v.reshape(N, Sk, H, d).permute(0, 2, 1, 3)
        s = qh @ kh.transpose(-1, -2) / (d ** 0.5)
        if causal:
            s = s.masked_fill(torch.triu(torch.ones(Sq, Sk, dtype=torch.bool), 1), float("-inf"))
        return (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(N, Sq, D)

    a = [t.double().requires_grad_(True) for t in (q, k, v)]
    (ref(*a) * do.double()).sum().backward()
    b = [t.cuda().requires_grad_(True) for t in (q, k, v)]
    o = QF.attention(b[0], b[1], b[2], H, causal)
    (o * do.cuda()).sum().backward()
    assert rel_err(o, ref(*a)) < 2e-6
    for x, y in zip(b, a):
        assert rel_err(x.grad, y.grad) < 5e-6


def test_attention_layer_any_head_count_that_divides_the_width():
    """heads=4 on a 48-wide model (head dim 12) and heads=4 on a 512-wide one (head dim 128: what a README-width
    model with few heads has): the layer builds, trains and matches fp64 attention; heads wider than the widest
    kernel raise with the reason."""
    from models.layers import AttentionLayer
    torch.manual_seed(3)
    lay = AttentionLayer(heads=4, in_dim=48, hidden_dim=64, use_cross_attn=False, use_masked_attn=True).cuda()
    assert lay.head_dim == 12
    x = torch.randn(2, 19, 48).cuda().requires_grad_(True)
    y = lay(x)
    y.square().sum().backward()
    # fp64 restatement over the layer's own q/k/v MLP outputs
    with torch.no_grad():
        q, k, v = (blk(x.detach()) for blk in (lay.q_block, lay.k_block, lay.v_block))
    want = _attn_ref64(q.double(), k.double(), v.double(), 4, True)
    assert rel_err(y, want) < 2e-6
    assert x.grad is not None and torch.isfinite(x.grad).all()
    wide = AttentionLayer(heads=4, in_dim=512, hidden_dim=64, use_cross_attn=False, use_masked_attn=True).cuda()
    assert wide.head_dim == 128
    xw = torch.randn(2, 70, 512).cuda().requires_grad_(True)
    yw = wide(xw)
    yw.square().sum().backward()
    with torch.no_grad():
        q, k, v = (blk(xw.detach()) for blk in (wide.q_block, wide.k_block, wide.v_block))
    assert rel_err(yw, _attn_ref64(q.double(), k.double(), v.double(), 4, True)) < 2e-6
    assert torch.isfinite(xw.grad).all() and float(xw.grad.abs().max()) > 0
    with pytest.raises(ValueError, match="head dim 256"):
        AttentionLayer(heads=2, in_dim=512, hidden_dim=64, use_cross_attn=False)


def _attn_ref64(q, k, v, H, causal):
    """fp64 attention on the GPU with plain torch ops (the checker for the large shapes)."""
    N, Sq, D = q.shape
    Sk, d = k.shape[1], D // H
    qh = q.reshape(N, Sq, H, d).permute(0, 2, 1, 3)
    kh = k.reshape(N, Sk, H, d).permute(0, 2, 1, 3)
    vh = v.reshape(N, Sk, H, d).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / (d ** 0.5)
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(Sq, Sk, dtype=torch.bool, device=q.device), 1), float("-inf"))
    return (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(N, Sq, D)


@pytest.mark.parametrize("qw,bw", [(1, 1), (2, 2), (4, 2), (4, 4)])
@pytest.mark.parametrize("N,Sq,Sk,H,d,causal", [(2, 300, 300, 6, 8, True), (1, 1024, 1024, 8, 8, True),
                                                (2, 200, 333, 5, 8, False), (1, 520, 520, 4, 16, True)])
def test_attention_workgroup_shapes(N, Sq, Sk, H, d, causal, qw, bw, monkeypatch):
    """Every workgroup geometry of the MFMA attention kernels (1, 2 or 4 query / key slices per
    head, several K/V chunks per workgroup, ragged last chunk, a head count that is not a
    multiple of the 4 heads a workgroup covers) against fp64."""
    from qarig import functional as QF
    monkeypatch.setenv("QARIG_ATTN_QW", str(qw))
    monkeypatch.setenv("QARIG_ATTN_BW", str(bw))
    g = torch.Generator().manual_seed(Sq + Sk + H)
    D = H * d
    q, k, v = (torch.randn((N, S, D), generator=g).cuda() for S in (Sq, Sk, Sk))
    do = torch.randn((N, Sq, D), generator=g).cuda()
    a = [t.double().requires_grad_(True) for t in (q, k, v)]
    oa = _attn_ref64(*a, H, causal)
    (oa * do.double()).sum().backward()
    b = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o = QF.attention(b[0], b[1], b[2], H, causal)
    (o * do).sum().backward()
    assert rel_err(o, oa) < 2e-6
    for x, y in zip(b, a):
        assert rel_err(x.grad, y.grad) < 5e-6


def test_attention_4096_tokens_64_heads_vs_fp64():
    """BASELINE config 5's sequence: one 4096-token sequence, 64 heads of dim 8, causal; forward
    and all three gradients vs fp64 (computed head group by head group to bound memory)."""
    from qarig import functional as QF
    g = torch.Generator().manual_seed(4096)
    N, S, H, d = 1, 4096, 64, 8
    D = H * d
    q, k, v = (torch.randn((N, S, D), generator=g).cuda() for _ in range(3))
    do = torch.randn((N, S, D), generator=g).cuda()
    b = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o = QF.attention(b[0], b[1], b[2], H, True)
    (o * do).sum().backward()
    worst = 0.0
    for h0 in range(0, H, 8):                      # 8 heads at a time: 8 x 4096^2 fp64 scores = 1 GiB
        sl = slice(h0 * d, (h0 + 8) * d)
        a = [t[:, :, sl].double().requires_grad_(True) for t in (q, k, v)]
        oa = _attn_ref64(*a, 8, True)
        (oa * do[:, :, sl].double()).sum().backward()
        worst = max(worst, rel_err(o[:, :, sl], oa))
        for x, y in zip(b, a):
            worst = max(worst, rel_err(x.grad[:, :, sl], y.grad) / 2.5)
        del a, oa
    assert worst < 2e-6, worst


def test_attention_causal_invariance():
    from qarig import ops
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn((1, 40, 32), generator=g).cuda() for _ in range(3))
    o1, _ = ops.attention_fwd(q, k, v, 4, True)
    k2, v2 = k.clone(), v.clone()
    k2[:, -1] += 1.0
    v2[:, -1] -= 2.0
    o2, _ = ops.attention_fwd(q, k2, v2, 4, True)
    assert torch.equal(o1[:, :-1], o2[:, :-1]) and not torch.equal(o1[:, -1], o2[:, -1])


def test_cross_entropy_and_embedding():
    from qarig import functional as QF
    g = torch.Generator().manual_seed(3)
    M, C = 50, 513
    logits = torch.randn((M, C), generator=g) * 3
    tgt = torch.randint(0, C, (M,), generator=g)
    a = logits.double().requires_grad_(True)
    la = torch.nn.functional.cross_entropy(a, tgt)
    (la * 1.7).backward()
    b = logits.cuda().requires_grad_(True)
    lb = QF.cross_entropy(b, tgt.cuda())
    (lb * 1.7).backward()
    assert abs(float(lb.detach()) - float(la.detach())) < 1e-6 * max(1, abs(float(la)))
    assert rel_err(b.grad, a.grad) < 2e-6
    # embedding + positions
    V, D, N, S = 40, 32, 3, 12
    table = torch.randn((V, D), generator=g)
    ids = torch.randint(0, V, (N, S), generator=g)
    pe = torch.randn((S, D), generator=g)
    dy = torch.randn((N, S, D), generator=g)
    ta = table.double().requires_grad_(True)
    ((ta[ids] + pe.double()) * dy.double()).sum().backward()
    tb = table.cuda().requires_grad_(True)
    out = QF.embedding_pos(ids.cuda(), tb, pe.cuda())
    (out * dy.cuda()).sum().backward()
    assert rel_err(out, ta.detach()[ids] + pe.double()) < 1e-7
    assert rel_err(tb.grad, ta.grad) < 2e-6


def _mlp_ref(x, w1, b1, w2, b2, act2):
    h = torch.nn.functional.silu(x @ w1.t() + b1)
    y = h @ w2.t() + b2
    return torch.nn.functional.silu(y) if act2 else y


@pytest.mark.parametrize("act2", [0, 1])
def test_mlp2_and_residual_linear(act2):
    from qarig import functional as QF
    g = torch.Generator().manual_seed(5)
    M, D, Hd, O = 130, 32, 64, 33
    x = torch.randn((2, M // 2, D), generator=g)
    w1, b1 = torch.randn((Hd, D), generator=g) * 0.2, torch.randn(Hd, generator=g)
    w2, b2 = torch.randn((O, Hd), generator=g) * 0.2, torch.randn(O, generator=g)
    dy = torch.randn((2, M // 2, O), generator=g)
    a = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    (_mlp_ref(*a, act2) * dy.double()).sum().backward()
    b = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y = QF.mlp2(*b, 1, act2)
    (y * dy.cuda()).sum().backward()
    assert rel_err(y, _mlp_ref(*a, act2)) < 2e-6
    for p, q in zip(b, a):
        assert grad_err(p.grad, q.grad) < 5e-6
    # act(linear(x * s) + skip)
    s, skip = torch.randn((2, M // 2, D), generator=g), torch.randn((2, M // 2, Hd), generator=g)
    dz = torch.randn((2, M // 2, Hd), generator=g)
    a = [t.double().requires_grad_(True) for t in (x, s, skip, w1, b1)]
    (torch.nn.functional.silu((a[0] * a[1]) @ a[3].t() + a[4] + a[2]) * dz.double()).sum().backward()
    b = [t.cuda().requires_grad_(True) for t in (x, s, skip, w1, b1)]
    z = QF.linear_act(QF.mul(b[0], b[1]), b[3], b[4], residual=b[2], act=1)
    (z * dz.cuda()).sum().backward()
    for p, q in zip(b, a):
        assert grad_err(p.grad, q.grad) < 5e-6


TAGS = ["base", "base_pos", "encdec", "encdec_pos"]


def _build(tag, g):
    from models.Transformer import Transformer
    use_enc, use_pos = tag.startswith("encdec"), tag.endswith("pos")
    m = Transformer(use_encoder=use_enc, use_pos_cond=use_pos, num_enc_layers=2 if use_enc else None,
                    num_dec_layers=2, num_enc_embedding=24 if use_enc else None,
                    num_dec_embedding=40, self_attn_heads=4,
                    cross_attn_heads=2 if use_enc else None, transformer_in_dim=32,
                    transformer_out_dim=33, transformer_hidden_dim=64, hidden_activation="silu")
    m.custom_load_state_dict(g["sd"])
    return m.cuda()


@pytest.fixture(params=[False, True], ids=["per_token_cond", "position_table"])
def cond_table(request):
    """Both forms of the conditioning path: the reference's per-token evaluation, and the
    position table forced on (these shapes are too small for the default heuristic)."""
    from qarig import functional as QF
    old = (QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO)
    QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO = request.param, 0
    yield request.param
    QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO = old


@pytest.mark.parametrize("tag", TAGS)
def test_transformer_vs_reference_golden(tag, cond_table):
    """Reference weights + inputs -> logits, loss, every parameter gradient, and the
    weights after one Adam(0.5, 0.999) step, against what the reference produced."""
    from qarig import functional as QF
    from qarig.optim import FlatAdam
    g = load_golden("transformer_" + tag)
    m = _build(tag, g)
    opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
    x_enc = g["x_enc"].cuda() if "x_enc" in g else None
    pos = g["pos"].cuda() if "pos" in g else None
    opt.zero_grad()
    logits = m(g["x_dec"].cuda(), x_enc, pos)
    assert logits.shape == g["logits"].shape
    assert rel_err(logits, g["logits"]) < TOL
    loss = QF.cross_entropy(logits.view(-1, logits.shape[-1]), g["target"].cuda().flatten())
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for n, p in m.named_parameters():
        assert grad_err(p.grad, g["grads"][n]) < 5e-5, n
    if pos is not None:
        lf = m(g["x_dec"].cuda(), x_enc, pos.float())
        assert rel_err(lf, g["logits_float_pos"]) < TOL
    # Adam on the reference's own gradients (isolates the optimiser from grad noise)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.grad.copy_(g["grads"][n])
    opt.step()
    for n, p in m.state_dict().items():
        assert rel_err(p, g["sd_after_adam"][n]) < 1e-5, n


def test_transformer_readme_shape_vs_oracle(cond_table):
    """README-shaped block sizes (in 512 / hidden 2048 / 64 heads -> head dim 8), 2
    decoder layers, sliding-window conditioning, vs the CPU oracle in fp64."""
    from models.Transformer import Transformer
    from oracle import ref_models as rm
    torch.manual_seed(3)
    m = Transformer(use_encoder=True, use_pos_cond=True, num_enc_layers=1, num_dec_layers=2,
                    num_enc_embedding=512, num_dec_embedding=513, self_attn_heads=64,
                    cross_attn_heads=64, transformer_in_dim=512, transformer_out_dim=513,
                    transformer_hidden_dim=2048)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
    N, S, Se = 2, 256, 16
    x_dec = torch.randint(0, 513, (N, S), generator=gen)
    x_enc = torch.randint(0, 512, (N, Se), generator=gen)
    pos = torch.randint(0, 700, (N, 1), generator=gen) + torch.arange(S)[None]
    sd64 = {k: v.double() for k, v in m.state_dict().items()}
    cfg = dict(use_encoder=True, use_pos_cond=True, num_enc_layers=1, num_dec_layers=2,
               self_attn_heads=64, cross_attn_heads=64, hidden_activation="silu")
    ref = rm.transformer_forward(sd64, cfg, x_dec, x_enc, pos)
    m = m.cuda()
    with torch.no_grad():
        got = m(x_dec.cuda(), x_enc.cuda(), pos.cuda())
    span = float(ref.max() - ref.min())
    assert float((got.cpu().double() - ref).abs().max()) / span < TOL
    # causal invariance, exact: changing the last token leaves earlier logits bit-identical
    x2 = x_dec.clone()
    x2[:, -1] = (x2[:, -1] + 1) % 513
    with torch.no_grad():
        got2 = m(x2.cuda(), x_enc.cuda(), pos.cuda())
    assert torch.equal(got[:, :-1], got2[:, :-1])


def test_readme_block_sizes_train_step_grads_vs_oracle(cond_table):
    """README block sizes (512 / 2048 / 64 heads, window conditioning), 2 decoder layers,
    batch 2 x 256 tokens: loss and parameter gradients of the HIP path (interior GEMM
    kernels, LDS-broadcast attention, fused gradient accumulation) vs the CPU oracle."""
    from models.Transformer import Transformer
    from oracle import ref_models as rm
    from qarig import functional as QF
    from qarig.optim import FlatAdam
    torch.manual_seed(5)
    m = Transformer(use_encoder=False, use_pos_cond=True, num_enc_layers=None, num_dec_layers=2,
                    num_enc_embedding=None, num_dec_embedding=1024, self_attn_heads=64,
                    cross_attn_heads=None, transformer_in_dim=512, transformer_out_dim=513,
                    transformer_hidden_dim=2048)
    gen = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
    N, S = 2, 256
    x = torch.randint(0, 1024, (N, S), generator=gen)
    t = torch.randint(0, 513, (N, S), generator=gen)
    pos = torch.randint(0, 2, (N, 1), generator=gen) + torch.arange(S)[None]
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    cfg = dict(use_encoder=False, use_pos_cond=True, num_dec_layers=2, self_attn_heads=64,
               hidden_activation="silu")
    ref_loss = rm.cross_entropy(rm.transformer_forward(sd, cfg, x, None, pos), t)
    ref_loss.backward()
    m = m.cuda()
    opt = FlatAdam(m.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt.zero_grad()
    logits = m(x.cuda(), None, pos.cuda())
    loss = QF.cross_entropy(logits.view(-1, 513), t.cuda().flatten())
    loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 2e-5
    worst = 0.0
    for n, p in m.named_parameters():
        worst = max(worst, grad_err(p.grad, sd[n].grad, floor=1e-7))
    assert worst < 2e-4, worst   # fp32 vs fp32 with different summation orders, K up to 2048


@pytest.mark.parametrize("kv_grouping", ["all", "layer"])
def test_readme_block_sizes_encdec_train_step_grads_vs_oracle(cond_table, kv_grouping):
    """The enc-dec stage at README block sizes (512 / 2048 / 64 + 64 heads of 8), 2 decoder + 1 encoder
    layers, 2 x 256 decoder tokens against 256 encoder tokens (reference models/layers.py:538-599: cross
    attention dQ / dK / dV with Sk = S_enc, the k / v MLPs on the encoder output): loss and EVERY
    parameter gradient against the CPU oracle, with the cross-attention k / v MLPs grouped per layer and
    over all layers (2,048-row shapes: the grouped launches are on)."""
    from models.Transformer import Transformer
    from oracle import ref_models as rm
    from qarig import functional as QF
    from qarig.optim import FlatAdam
    torch.manual_seed(6)
    m = Transformer(use_encoder=True, use_pos_cond=True, num_enc_layers=1, num_dec_layers=2,
                    num_enc_embedding=512, num_dec_embedding=513, self_attn_heads=64,
                    cross_attn_heads=64, transformer_in_dim=512, transformer_out_dim=513,
                    transformer_hidden_dim=2048)
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
    N, S, S_enc = 2, 256, 256
    x = torch.randint(0, 513, (N, S), generator=gen)
    x_enc = torch.randint(0, 512, (N, S_enc), generator=gen)
    t = torch.randint(0, 513, (N, S), generator=gen)
    pos = torch.randint(0, 2, (N, 1), generator=gen) + torch.arange(S)[None]
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    cfg = dict(use_encoder=True, use_pos_cond=True, num_enc_layers=1, num_dec_layers=2, self_attn_heads=64,
               cross_attn_heads=64, hidden_activation="silu")
    ref_loss = rm.cross_entropy(rm.transformer_forward(sd, cfg, x, x_enc, pos), t)
    ref_loss.backward()
    m = m.cuda()
    opt = FlatAdam(m.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt.zero_grad()
    old = QF.CROSS_KV_GROUPING
    try:
        QF.CROSS_KV_GROUPING = kv_grouping
        logits = m(x.cuda(), x_enc.cuda(), pos.cuda())
        loss = QF.cross_entropy(logits.view(-1, 513), t.cuda().flatten())
        loss.backward()
    finally:
        QF.CROSS_KV_GROUPING = old
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 2e-5
    worst, worst_name = 0.0, None
    for n, p in m.named_parameters():
        e = grad_err(p.grad, sd[n].grad, floor=1e-7)
        if e > worst:
            worst, worst_name = e, n
    assert worst < 2e-4, (worst, worst_name)   # fp32 vs fp32 with different summation orders, K up to 2048


def test_mlp2_ragged_wide_output_runs_padded_and_matches_fp64():
    """The classifier shape class (many rows, 513 = K_hr + 1 output columns): _MLP2 runs its second
    layer on zero-padded weights so that forward, d-input and d-weight take the interior kernels;
    values and every gradient against an fp64 evaluation."""
    from qarig import functional as QF
    g = torch.Generator().manual_seed(11)
    M, Din, Hd, N = 4096, 256, 512, 513
    assert QF._MLP2._padded_out(M, N, Hd, 0, True)
    x = torch.randn((M, Din), generator=g).cuda().requires_grad_(True)
    w1 = (torch.randn((Hd, Din), generator=g) * 0.05).cuda().requires_grad_(True)
    b1 = torch.randn((Hd,), generator=g).cuda().requires_grad_(True)
    w2 = (torch.randn((N, Hd), generator=g) * 0.05).cuda().requires_grad_(True)
    b2 = torch.randn((N,), generator=g).cuda().requires_grad_(True)
    gy = torch.randn((M, N), generator=g).cuda()
    y = QF.mlp2(x, w1, b1, w2, b2, 1, 0)
    got = torch.autograd.grad(y, (x, w1, b1, w2, b2), gy)
    xd, w1d, b1d, w2d, b2d = (t.detach().double().cpu().requires_grad_(True) for t in (x, w1, b1, w2, b2))
    yd = torch.nn.functional.silu(xd @ w1d.t() + b1d) @ w2d.t() + b2d
    want = torch.autograd.grad(yd, (xd, w1d, b1d, w2d, b2d), gy.double().cpu())
    assert y.shape == (M, N) and rel_err(y, yd) < 5e-6
    for a, b in zip(got, want):
        assert rel_err(a, b) < 2e-5
