"""Grouped GEMM launches (qarig_gemm_f32_grouped) and the MLP nodes built on them, against fp64 and
against the ungrouped path.  The q / k / v two-layer MLPs of reference models/layers.py:389-418 and the
cross-attention k / v MLPs of every decoder layer (models/layers.py:538-599) run through them on
per-GPU shards of a few thousand rows."""
import pytest
import torch

from conftest import grad_err, rel_err

pytestmark = pytest.mark.gpu

GEMM_TOL = 2e-6


def _ref(A, B, ak, bk):
    Ad = A.double().cpu() if ak else A.double().cpu().t()
    Bd = B.double().cpu() if bk else B.double().cpu().t()
    return Ad @ Bd.t()


@pytest.mark.parametrize("G,M,N,K,ak,bk,splitk,shared_a", [
    (3, 256, 512, 512, True, True, 1, True),       # q/k/v first layer: one input, three weights
    (3, 256, 128, 2048, True, True, 4, False),     # second layer, split reduction + grouped reduce epilogue
    (2, 384, 256, 128, True, False, 1, False),     # dT1 = dT2 W2 * act'
    (1, 128, 128, 64, True, True, 1, False),       # a single member
    (5, 128, 256, 256, True, True, 2, True),       # odd member count
    (14, 128, 128, 192, True, False, 3, False),    # every decoder layer's k / v pair
    (16, 128, 128, 32, True, True, 1, True),       # the member limit
    (3, 128, 384, 640, False, False, 2, False),    # weight gradients (xc, xc)
])
def test_grouped_gemm_every_epilogue_vs_fp64(G, M, N, K, ak, bk, splitk, shared_a):
    from oracle import ref_models as rm
    from qarig import ops
    g = torch.Generator().manual_seed(G * 1000 + M + N + K + splitk)
    As = [torch.randn((M, K) if ak else (K, M), generator=g).cuda() for _ in range(1 if shared_a else G)]
    if shared_a:
        As = As * G
    Bs = [(torch.randn((N, K) if bk else (K, N), generator=g) * 0.1).cuda() for _ in range(G)]
    refs = [_ref(As[i], Bs[i], ak, bk) for i in range(G)]
    tol = GEMM_TOL * max(1, K / 512) ** 0.5
    assert ops.gemm_grouped_supported(M, N, K, splitk)
    # plain
    Cs = torch.full((G, M, N), float("nan"), device="cuda")
    ops.gemm_grouped(As, Bs, list(Cs.unbind(0)), M, N, K, ak, bk, splitk=splitk)
    for i in range(G):
        assert rel_err(Cs[i], refs[i]) < tol, i
    C2 = torch.empty_like(Cs)
    ops.gemm_grouped(As, Bs, list(C2.unbind(0)), M, N, K, ak, bk, splitk=splitk)
    assert torch.equal(Cs, C2)                      # deterministic slab order
    # agreement with the single-product entry up to the summation order of the split
    one = ops.gemm(As[G - 1], Bs[G - 1], ak, bk, splitk=1)
    assert rel_err(Cs[G - 1], one) < 2 * tol
    # bias + residual + saved pre-activation + activation, and the act' backward fusion
    bias = [torch.randn((N,), generator=g).cuda() for _ in range(G)]
    res = [torch.randn((M, N), generator=g).cuda() for _ in range(G)]
    Z = [torch.randn((M, N), generator=g).cuda() for _ in range(G)]
    for act_name, act in (("silu", 1), ("tanh", 2), (None, 0)):
        Y = torch.empty((G, M, N), device="cuda")
        P = torch.empty((G, M, N), device="cuda")
        ops.gemm_grouped(As, Bs, list(Y.unbind(0)), M, N, K, ak, bk, bias=bias, residual=res,
                         preact=list(P.unbind(0)), act=act, splitk=splitk)
        for i in range(G):
            t = refs[i] + bias[i].double().cpu() + res[i].double().cpu()
            assert rel_err(P[i], t) < tol
            assert rel_err(Y[i], rm.activation(t, act_name)) < (2e-5 if act_name == "tanh" else 5e-6)
    Gz = torch.empty((G, M, N), device="cuda")
    ops.gemm_grouped(As, Bs, list(Gz.unbind(0)), M, N, K, ak, bk, gradz=Z, gact=1, splitk=splitk)
    for i in range(G):
        Zd = Z[i].double().cpu().requires_grad_(True)
        rm.activation(Zd, "silu").sum().backward()
        assert rel_err(Gz[i], refs[i] * Zd.grad) < 5e-6
    # accumulate into existing outputs (weight gradients into .grad), with the A row sums riding
    out = torch.randn((G, M, N), generator=g).cuda()
    want = [out[i].double().cpu() + refs[i] for i in range(G)]
    rs = torch.randn((G, M), generator=g).cuda() if not ak else None
    rs_want = [rs[i].double().cpu() + As[i].double().cpu().sum(0) for i in range(G)] if rs is not None else None
    ops.gemm_grouped(As, Bs, list(out.unbind(0)), M, N, K, ak, bk, splitk=splitk, accumulate=True,
                     a_rowsum=list(rs.unbind(0)) if rs is not None else None)
    for i in range(G):
        assert rel_err(out[i], want[i]) < tol
        if rs is not None:
            assert rel_err(rs[i], rs_want[i]) < 2e-6 * max(1, K / 512) ** 0.5
    # sum over the members into one output (input gradient of MLPs that share their input)
    S = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_grouped(As, Bs, [S], M, N, K, ak, bk, splitk=splitk, sum_groups=True)
    assert rel_err(S, sum(refs)) < tol * G ** 0.5
    S0 = torch.randn((M, N), generator=g).cuda()
    want = S0.double().cpu() + sum(refs)
    ops.gemm_grouped(As, Bs, [S0], M, N, K, ak, bk, splitk=splitk, sum_groups=True, accumulate=True)
    assert rel_err(S0, want) < tol * G ** 0.5


def test_grouped_gemm_refuses_what_it_cannot_run():
    from qarig import ops
    A = torch.randn((128, 64)).cuda()
    C = torch.empty((128, 128)).cuda()
    assert not ops.gemm_grouped_supported(100, 128, 64)        # ragged rows
    assert not ops.gemm_grouped_supported(128, 128, 64, 8)     # half a k-tile per split
    with pytest.raises(RuntimeError, match="gemm_grouped"):
        ops.gemm_grouped([A[:100]], [A], [C[:100]], 100, 128, 64)
    with pytest.raises(RuntimeError, match="16-B aligned"):
        big = torch.randn((128, 68)).cuda()
        ops.gemm_grouped([big[:, 1:65]], [A], [C], 128, 128, 64)
    with pytest.raises(RuntimeError, match="plain epilogue"):
        ops.gemm_grouped([A], [A], [C], 128, 128, 64, bias=[C[0]], accumulate=True)


def _blocks(G, D, H, O, g):
    ps = []
    for _ in range(G):
        ps.append(tuple(torch.nn.Parameter(t.cuda()) for t in (
            torch.randn((H, D), generator=g) * 0.05, torch.randn((H,), generator=g) * 0.1,
            torch.randn((O, H), generator=g) * 0.05, torch.randn((O,), generator=g) * 0.1)))
    return ps


@pytest.mark.parametrize("G,M,D,H,act2,slots", [(3, 256, 128, 256, 0, True), (3, 256, 128, 256, 1, False),
                                                  (2, 384, 256, 512, 0, True), (14, 128, 128, 128, 0, True)])
def test_grouped_mlp_node_matches_separate_mlps_and_fp64(G, M, D, H, act2, slots):
    """_MLP2xG (every product one grouped launch) against G separate _MLP2 nodes and against an fp64
    evaluation: outputs, input gradient, every weight / bias gradient; with the gradients accumulated
    into pre-existing .grad buffers (the FlatAdam layout) and returned as tensors."""
    from qarig import functional as QF
    g = torch.Generator().manual_seed(G + M + D + H + act2)
    blocks = _blocks(G, D, H, D, g)
    x = torch.randn((2, M // 2, D), generator=g).cuda().requires_grad_(True)
    dys = [torch.randn((2, M // 2, D), generator=g).cuda() for _ in range(G)]
    if slots:
        for p in blocks:
            for t in p:
                t.grad = torch.zeros_like(t)
    old = QF.MLP_GROUPED
    try:
        QF.MLP_GROUPED = "1"
        assert QF._mlp_group_ok(x, blocks, M)
        ys = QF.mlp2xg(x, blocks, 1, act2)
        torch.autograd.backward(ys, dys)
        got = [[t.grad.clone() for t in p] for p in blocks]
        gx = x.grad.clone()
        x.grad = None
        for p in blocks:
            for t in p:
                t.grad = torch.zeros_like(t) if slots else None
        QF.MLP_GROUPED = "0"
        ys0 = QF.mlp2xg(x, blocks, 1, act2)
        torch.autograd.backward(ys0, dys)
    finally:
        QF.MLP_GROUPED = old
    # fp64
    xd = x.detach().double().cpu().requires_grad_(True)
    pd = [[t.detach().double().cpu().requires_grad_(True) for t in p] for p in blocks]
    silu = torch.nn.functional.silu
    yd = []
    for w1, b1, w2, b2 in pd:
        y = silu(xd @ w1.t() + b1) @ w2.t() + b2
        yd.append(silu(y) if act2 else y)
    torch.autograd.backward(yd, [d.double().cpu() for d in dys])
    for i in range(G):
        assert rel_err(ys[i], yd[i]) < 5e-6
        assert rel_err(ys[i], ys0[i]) < 5e-6
        for j in range(4):
            assert grad_err(got[i][j], pd[i][j].grad) < 1e-5, (i, j)
            assert grad_err(got[i][j], blocks[i][j].grad) < 1e-5, (i, j)
    assert grad_err(gx, xd.grad) < 1e-5
    assert grad_err(gx, x.grad) < 1e-5


def _encdec(seed=5, layers=3):
    from models.Transformer import Transformer
    torch.manual_seed(seed)
    m = Transformer(use_encoder=True, use_pos_cond=True, num_enc_layers=2, num_dec_layers=layers,
                    num_enc_embedding=40, num_dec_embedding=50, self_attn_heads=4, cross_attn_heads=4,
                    transformer_in_dim=128, transformer_out_dim=50, transformer_hidden_dim=256)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return m.cuda()


@pytest.mark.parametrize("mode,cond_form", [("all", "table"), ("layer", "table"), ("all", "tokens"), ("layer", "tokens")])
def test_transformer_step_grouped_vs_ungrouped(mode, cond_form):
    """One enc-dec training step with the grouped q/k/v + cross-attention k/v launches (both groupings of
    the cross-attention pairs) against the same step with grouping off: logits, loss, every gradient.
    cond_form "tokens": per-token conditioning (no position table) with its 9 projections per decoder
    layer evaluated as grouped launches (QF.CondTokens; one group for the decoder, or one per layer)
    against one Linear node per projection."""
    from qarig import functional as QF
    from qarig.optim import FlatAdam
    N, S, Se = 2, 128, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, 50, (N, S), generator=g).cuda()
    xe = torch.randint(0, 40, (N, Se), generator=g).cuda()
    tg = torch.randint(0, 50, (N, S), generator=g).cuda()
    pos = (torch.arange(S)[None] + torch.tensor([[0], [7]])).cuda()
    res = {}
    old = (QF.MLP_GROUPED, QF.CROSS_KV_GROUPING, QF.COND_TABLE_MIN_RATIO, QF.USE_COND_TABLE, QF.USE_COND_GROUPS,
           QF.COND_TABLE_GROUPING)
    try:
        QF.COND_TABLE_MIN_RATIO = 0
        QF.USE_COND_TABLE = cond_form == "table"
        QF.COND_TABLE_GROUPING = mode
        for tag, grouped in (("on", "1"), ("off", "0")):
            QF.MLP_GROUPED, QF.CROSS_KV_GROUPING = grouped, mode
            QF.USE_COND_GROUPS = grouped == "1"
            m = _encdec()
            opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
            opt.zero_grad()
            logits = m(x, xe, pos, pos_bound=S + 8)
            assert m._last_cond_form == ("table" if cond_form == "table" else "per_token")
            loss = QF.cross_entropy(logits.view(-1, 50), tg.flatten())
            loss.backward()
            res[tag] = (logits.detach().clone(), float(loss.detach()), opt.flat_grad.clone(),
                        {k: p.grad.clone() for k, p in m.named_parameters()})
    finally:
        (QF.MLP_GROUPED, QF.CROSS_KV_GROUPING, QF.COND_TABLE_MIN_RATIO, QF.USE_COND_TABLE, QF.USE_COND_GROUPS,
         QF.COND_TABLE_GROUPING) = old
    assert rel_err(res["on"][0], res["off"][0]) < 5e-6
    assert abs(res["on"][1] - res["off"][1]) < 1e-5
    for k, gref in res["off"][3].items():
        assert grad_err(res["on"][3][k], gref, floor=1e-5) < 5e-5, k
