"""Position-table conditioning (csrc/condtable.hip): the table path must reproduce the
reference's per-token evaluation -- same logits and loss (1e-5), same gradient for EVERY
parameter (5e-5 relative: summation order only) -- and its building blocks are checked on their own."""
import pytest
import torch

from conftest import grad_err, rel_err

pytestmark = pytest.mark.gpu


def test_rowmap_and_segment_sum():
    from qarig import ops
    g = torch.Generator().manual_seed(0)
    M, P, D = 5000, 37, 64
    idx = torch.randint(0, P - 3, (M,), generator=g).to(torch.int32)     # rows P-3.. stay empty
    off, rows = ops.rowmap_build(idx.cuda(), P)
    off, rows = off.cpu(), rows.cpu()
    assert off[0] == 0 and off[-1] == M
    for p in range(P):
        want = torch.nonzero(idx == p).flatten().to(torch.int32)
        assert torch.equal(rows[off[p]:off[p + 1]], want)                  # ascending token order
    src = torch.randn((M, D), generator=g)
    got = ops.segment_sum(src.cuda(), off.cuda(), rows.cuda())
    want = torch.zeros((P, D), dtype=torch.float64).index_add_(0, idx.long(), src.double())
    assert rel_err(got, want) < 1e-6
    assert not got[P - 3:].any()
    assert torch.equal(got, ops.segment_sum(src.cuda(), off.cuda(), rows.cuda()))   # deterministic
    ops.check_index_flag(torch.device("cuda"), "rowmap")
    bad = idx.clone()
    bad[17] = P
    ops.rowmap_build(bad.cuda(), P)
    with pytest.raises(IndexError):
        ops.check_index_flag(torch.device("cuda"), "rowmap")


def test_table_ops_match_per_token_ops():
    from qarig import functional as QF
    g = torch.Generator().manual_seed(1)
    N, S, D, P = 3, 40, 256, 50
    idx = (torch.randint(0, P - S + 1, (N, 1), generator=g) + torch.arange(S)[None]).reshape(-1)
    x = torch.randn((N, S, D), generator=g).cuda().requires_grad_(True)
    st = torch.randn((P, D), generator=g).cuda().requires_grad_(True)
    sh = torch.randn((P, D), generator=g).cuda().requires_grad_(True)
    gy = torch.randn((N, S, D), generator=g).cuda()
    cond = QF.CondTable(st, idx.to(torch.int32).cuda(), (N, S))
    for name in ("ln", "mul"):
        if name == "ln":
            y = QF.layernorm_mod_table(x, st, sh, cond)
            y_ref = QF.layernorm_mod(x, st[idx.cuda()].reshape(N, S, D), sh[idx.cuda()].reshape(N, S, D))
        else:
            y = QF.mul_table(x, st, cond)
            y_ref = QF.mul(x, st[idx.cuda()].reshape(N, S, D))
        assert torch.equal(y, y_ref)                 # forward: the same arithmetic per element
        got = torch.autograd.grad(y, (x, st, sh) if name == "ln" else (x, st), gy)
        want = torch.autograd.grad(y_ref, (x, st, sh) if name == "ln" else (x, st), gy)
        for a, b in zip(got, want):
            assert rel_err(a, b) < 2e-6


@pytest.mark.parametrize("use_encoder", [False, True])
def test_table_path_equals_per_token_path_for_every_parameter(use_encoder):
    from models.Transformer import Transformer
    from qarig import functional as QF
    torch.manual_seed(5)
    m = Transformer(use_encoder=use_encoder, use_pos_cond=True, num_enc_layers=1 if use_encoder else None,
                    num_dec_layers=2, num_enc_embedding=30 if use_encoder else None,
                    num_dec_embedding=50, self_attn_heads=8, cross_attn_heads=8 if use_encoder else None,
                    transformer_in_dim=64, transformer_out_dim=41, transformer_hidden_dim=128).cuda()
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    N, S, total = 6, 32, 40
    x = torch.randint(0, 50, (N, S), generator=g).cuda()
    e = torch.randint(0, 30, (N, 7), generator=g).cuda() if use_encoder else None
    t = torch.randint(0, 41, (N, S), generator=g).cuda()
    pos = (torch.randint(0, total - S + 1, (N, 1), generator=g) + torch.arange(S)[None]).cuda()
    res = {}
    ratio = QF.COND_TABLE_MIN_RATIO
    for table in (False, True):
        QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO = table, 0     # tiny shapes: force the table
        try:
            m.zero_grad()
            logits = m(x, e, pos, pos_bound=total if table else None)
            loss = QF.cross_entropy(logits.view(-1, 41), t.flatten())
            loss.backward()
            res[table] = (logits.detach().clone(), float(loss.detach()),
                          {n: p.grad.detach().clone() for n, p in m.named_parameters()})
        finally:
            QF.USE_COND_TABLE = True
    assert m._last_cond_form == "table"
    assert rel_err(res[True][0], res[False][0]) < 1e-5
    assert abs(res[True][1] - res[False][1]) < 1e-6 * abs(res[False][1])
    for n, gref in res[False][2].items():
        # the golden tests' gradient bar; the floor covers gradients that are zero in exact
        # arithmetic (key biases: softmax ignores a constant added to every key) and come out
        # as 1e-14-sized rounding noise in both forms
        assert grad_err(res[True][2][n], gref, floor=1e-7) < 5e-5, n
    # without the bound the table is sized from the data (one read-back), same result
    try:
        with torch.no_grad():
            assert rel_err(m(x, e, pos), m(x, e, pos, pos_bound=total)) < 1e-6
            assert m._last_cond_form == "table"
            # float positions (sampling) keep the per-token form
            assert rel_err(m(x, e, pos.float()), res[False][0]) < 1e-5
            assert m._last_cond_form == "per_token"
            # default heuristic: 192 tokens do not pay for a 128-row table
            QF.COND_TABLE_MIN_RATIO = ratio
            m(x, e, pos, pos_bound=total)
            assert m._last_cond_form == "per_token"
    finally:
        QF.COND_TABLE_MIN_RATIO = ratio


@pytest.mark.parametrize("grouping", ["all", "layer"])
def test_grouped_table_projections_match_per_token_path(grouping):
    """D = 256: the table projections run as grouped launches (one for the whole decoder, or one
    per layer as under data parallelism) with the two-GEMM backward; every parameter gradient
    must still match the per-token evaluation."""
    from models.Transformer import Transformer
    from qarig import functional as QF
    torch.manual_seed(7)
    m = Transformer(use_encoder=False, use_pos_cond=True, num_enc_layers=None, num_dec_layers=3,
                    num_enc_embedding=None, num_dec_embedding=60, self_attn_heads=32,
                    cross_attn_heads=None, transformer_in_dim=256, transformer_out_dim=41,
                    transformer_hidden_dim=512).cuda()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    N, S, total = 8, 64, 100
    x = torch.randint(0, 60, (N, S), generator=g).cuda()
    t = torch.randint(0, 41, (N, S), generator=g).cuda()
    pos = (torch.randint(0, total - S + 1, (N, 1), generator=g) + torch.arange(S)[None]).cuda()
    old = (QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO, QF.COND_TABLE_GROUPING)
    res = {}
    try:
        for table in (False, True):
            QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO, QF.COND_TABLE_GROUPING = table, 0, grouping
            m.zero_grad()
            calls = []
            orig = QF._TableProjections.apply
            if table:
                QF._TableProjections.apply = staticmethod(lambda *a: (calls.append(len(a)), orig(*a))[1])
            try:
                logits = m(x, None, pos, pos_bound=total)
                loss = QF.cross_entropy(logits.view(-1, 41), t.flatten())
                loss.backward()
            finally:
                QF._TableProjections.apply = orig
            if table:     # 6 projections (12 parameters + the table) per base decoder layer
                assert calls == ([1 + 12 * 3] if grouping == "all" else [13, 13, 13]), calls
            res[table] = (logits.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()})
    finally:
        QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO, QF.COND_TABLE_GROUPING = old
    assert rel_err(res[True][0], res[False][0]) < 1e-5
    for n, gref in res[False][1].items():
        assert grad_err(res[True][1][n], gref, floor=1e-7) < 5e-5, n


@pytest.mark.parametrize("slots", [True, False])
def test_long_position_tables_run_grouped_tile_launches(slots):
    """Sequences longer than 512 positions (BASELINE config 4: 1,025): the table projections go through the
    grouped 128 x 128-tile launches (18 members = two launches of 16 + 2) forward, for the summed table
    gradient and for the weight gradients + bias row sums -- into FlatAdam's .grad slots or as returned
    tensors; every parameter gradient against the per-token evaluation."""
    from models.Transformer import Transformer
    from qarig import functional as QF
    from qarig.optim import FlatAdam
    torch.manual_seed(8)
    m = Transformer(use_encoder=False, use_pos_cond=True, num_enc_layers=None, num_dec_layers=3,
                    num_enc_embedding=None, num_dec_embedding=60, self_attn_heads=16,
                    cross_attn_heads=None, transformer_in_dim=128, transformer_out_dim=41,
                    transformer_hidden_dim=256).cuda()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    if slots:
        FlatAdam(m.parameters(), lr=1e-3)          # points every .grad into its flat buffer
    N, S, total = 6, 128, 700                      # P = 768 rows
    x = torch.randint(0, 60, (N, S), generator=g).cuda()
    t = torch.randint(0, 41, (N, S), generator=g).cuda()
    pos = (torch.randint(0, total - S + 1, (N, 1), generator=g) + torch.arange(S)[None]).cuda()
    old = (QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO, QF.COND_TABLE_GROUPING)
    res = {}
    try:
        for table in (False, True):
            QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO, QF.COND_TABLE_GROUPING = table, 0, "all"
            for p in m.parameters():
                if slots:
                    p.grad.zero_()
                else:
                    p.grad = None
            calls = []
            orig = QF._TableProjections.apply
            if table:
                QF._TableProjections.apply = staticmethod(lambda *a: (calls.append(len(a)), orig(*a))[1])
            try:
                logits = m(x, None, pos, pos_bound=total)
                QF.cross_entropy(logits.view(-1, 41), t.flatten()).backward()
            finally:
                QF._TableProjections.apply = orig
            if table:
                assert calls == [1 + 12 * 3] and m._last_cond_form == "table", calls
            res[table] = (logits.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()})
    finally:
        QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO, QF.COND_TABLE_GROUPING = old
    assert rel_err(res[True][0], res[False][0]) < 1e-5
    for n, gref in res[False][1].items():
        assert grad_err(res[True][1][n], gref, floor=1e-7) < 5e-5, n


def test_out_of_range_positions_are_flagged_not_dereferenced():
    from qarig import functional as QF
    from qarig import ops
    tab = torch.randn((128, 256)).cuda()
    idx = torch.arange(200, dtype=torch.int32).cuda() - 20          # -20 .. 179, P = 128
    cond = QF.CondTable(tab, idx, (1, 200))
    assert int(cond.idx.min()) == 0 and int(cond.idx.max()) == 127
    y = QF.mul_table(torch.ones((1, 200, 256)).cuda(), tab, cond)
    assert torch.equal(y[0, 50], tab[30])
    with pytest.raises(IndexError):
        ops.check_index_flag(torch.device("cuda"), "positions")


def test_embedding_gradient_by_row_map_matches_index_add():
    from qarig import functional as QF
    g = torch.Generator().manual_seed(3)
    N, S, V, D = 32, 256, 100, 64                      # 8192 tokens: the row-map path
    ids = torch.randint(0, V, (N, S), generator=g).cuda()
    table = torch.randn((V, D), generator=g).cuda().requires_grad_(True)
    pe = torch.randn((S, D), generator=g).cuda()
    gy = torch.randn((N, S, D), generator=g).cuda()
    y = QF.embedding_pos(ids, table, pe)
    (got,) = torch.autograd.grad(y, table, gy)
    want = torch.zeros((V, D), dtype=torch.float64).index_add_(0, ids.cpu().reshape(-1),
                                                                 gy.double().cpu().reshape(-1, D))
    assert rel_err(got, want) < 2e-6


@pytest.mark.parametrize("grouping", ["all", "layer"])
def test_activation_checkpoint_with_grouped_table_matches_plain_run(grouping):
    """--use-activation-checkpoint with the position table in its grouped form (D % 256 == 0, the
    README shapes): the projections are cached on the CondTable, so they must be evaluated
    outside the checkpointed layer -- otherwise the recomputation saves fewer tensors than the
    original forward and torch raises CheckpointError.  Gradients must equal the plain run's."""
    from models.Transformer import Transformer
    from qarig import functional as QF
    g = torch.Generator().manual_seed(3)
    N, S, total = 8, 64, 100
    x = torch.randint(0, 60, (N, S), generator=g).cuda()
    t = torch.randint(0, 41, (N, S), generator=g).cuda()
    pos = (torch.randint(0, total - S + 1, (N, 1), generator=g) + torch.arange(S)[None]).cuda()
    old = (QF.COND_TABLE_MIN_RATIO, QF.COND_TABLE_GROUPING)
    res = {}
    try:
        QF.COND_TABLE_MIN_RATIO, QF.COND_TABLE_GROUPING = 0, grouping
        for ckpt in (False, True):
            torch.manual_seed(7)
            m = Transformer(use_encoder=False, use_pos_cond=True, num_enc_layers=None, num_dec_layers=3,
                            num_enc_embedding=None, num_dec_embedding=60, self_attn_heads=32,
                            cross_attn_heads=None, transformer_in_dim=256, transformer_out_dim=41,
                            transformer_hidden_dim=512, use_activation_checkpoint=ckpt).cuda()
            gw = torch.Generator().manual_seed(4)
            with torch.no_grad():
                for p in m.parameters():
                    if p.abs().max() == 0:
                        p.copy_(torch.randn(p.shape, generator=gw) * 0.05)
            logits = m(x, None, pos, pos_bound=total)
            assert m._last_cond_form == "table"
            loss = QF.cross_entropy(logits.view(-1, 41), t.flatten())
            loss.backward()
            res[ckpt] = (logits.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()})
    finally:
        QF.COND_TABLE_MIN_RATIO, QF.COND_TABLE_GROUPING = old
    assert torch.equal(res[True][0], res[False][0])
    for n, gref in res[False][1].items():
        assert torch.equal(res[True][1][n], gref), n      # same kernels, same order: bit-identical
