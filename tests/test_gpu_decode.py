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
    (4, 1, 8, 256), (13, 1, 1030, 1024), (12, 1, 600, 256), (6, 3, 2048, 512), (8, 1, 20, 1024), (16, 1, 40, 2048)])
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


@pytest.mark.parametrize("M,G,N,K", [(16, 3, 2048, 512), (16, 1, 512, 2048), (9, 1, 512, 512), (5, 1, 96, 1024)])
def test_row_split_and_all_rows_kernels_agree(M, G, N, K):
    """5 ... 16 rows: the kernel that splits the rows over the waves (decode_rows = 1, the default) against the
    one that holds all rows in every lane (decode_rows = 0) -- same results up to fp32 summation order, in the
    LayerNorm / residual / gate forms."""
    from qarig import ops, _lib
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn((M, K), generator=g) * 1.5 + 0.3).cuda()
    W = (torch.randn((G, N, K), generator=g) * 0.05).cuda()
    b = torch.randn((G, N), generator=g).cuda()
    sc, sh = torch.randn(K, generator=g).cuda(), torch.randn(K, generator=g).cuda()
    res, mul = torch.randn((M, N), generator=g).cuda(), torch.randn(N, generator=g).cuda()
    lib = _lib.load()
    outs = {}
    for v in (1, 0):
        old = lib.qarig_set_option(b"decode_rows", v)
        try:
            o = [ops.decode_linear(x, W, b, act=1)]
            if K <= 1024:
                o.append(ops.decode_linear(x, W, b, act=1, scale=sc, shift=sh))
            if G == 1:
                o.append(ops.decode_linear(x, W[0], b[0], act=1, residual=res, mul=mul))
            outs[v] = o
        finally:
            lib.qarig_set_option(b"decode_rows", old)
    for a_, c_ in zip(outs[1], outs[0]):
        assert rel_err(a_, c_) < 2e-6 and not torch.equal(a_, c_) or M * N < 64


def test_decode_embed_row_and_position_table():
    from qarig import ops
    g = torch.Generator().manual_seed(2)
    B, D, V, L, PD = 5, 64, 33, 12, 7 * 64
    table = torch.randn((V, D), generator=g).cuda()
    pe = torch.randn((L, D), generator=g).cuda()
    proj = torch.randn((L, PD), generator=g).cuda()
    ids = torch.randint(0, V, (B,), generator=g).cuda()
    row = torch.zeros(PD, device="cuda")
    ctl = torch.zeros(ops.DECODE_CTL_WORDS, dtype=torch.int32, device="cuda")
    for length in (0, 5, L - 1):
        x = ops.decode_embed(ids, table, pe, length=length, proj_table=proj, proj_row=row)
        assert torch.equal(x, table[ids] + pe[length]) and torch.equal(row, proj[length])
        ctl[0] = length
        steps = int(ctl[4])
        x2 = ops.decode_embed(ids, table, pe, ctl=ctl, length=99, proj_table=proj, proj_row=row)   # ctl wins
        assert torch.equal(x2, x) and int(ctl[4]) == steps + 1                                       # and counts the step
    assert torch.equal(ops.decode_embed(ids, table, None, length=0), table[ids])
    ops.check_index_flag(torch.device("cuda"), "valid ids")
    bad = ids.clone()
    bad[2] = V
    x = ops.decode_embed(bad, table, pe, length=1)
    assert not x[2].any() and torch.equal(x[0], table[ids[0]] + pe[1])
    with pytest.raises(IndexError):
        ops.check_index_flag(torch.device("cuda"), "decode_embed")


def _sample_once(logits, T, end, gen, shift, uniforms, ctl, slot, bw, forced=None, log=True, inc=False):
    from qarig import ops
    B, V = logits.shape
    ids = torch.zeros(B, dtype=torch.int64, device="cuda")
    chunk = torch.full((B, bw), -7, dtype=torch.int64, device="cuda")
    comb = torch.full((B,), 0.5, device="cuda")
    probs = torch.zeros((uniforms.shape[0], B, V), device="cuda") if log else None
    ops.decode_sample(logits, T, end, gen, shift, uniforms, ctl, slot, bw, ids, chunk, comb, forced=forced,
                      probs_log=probs, inc_len=inc)
    return ids, chunk, comb, probs


@pytest.mark.parametrize("V", [513, 8193, 41, 3])
def test_decode_sample_is_the_references_draw(V):
    """softmax(logits / T), <end> zeroed (generate mode), a token by inverse CDF of the row's uniform, the running
    product, the train-mode <end> -> 0 hack, forced tokens, the counters (generate_images.py:289-304)."""
    from qarig import ops
    g = torch.Generator().manual_seed(V)
    B, bw, draws, end, T = 6, 4, 5, V - 1, 0.7
    logits = (torch.randn((B, V), generator=g) * 3).cuda()
    uniforms = torch.rand((draws, B), generator=g).cuda()
    uniforms[3, 0], uniforms[3, 1] = 0.0, 0.99999994      # the ends of the interval
    ctl = torch.zeros(ops.DECODE_CTL_WORDS, dtype=torch.int32, device="cuda")
    for d, slot in ((0, 0), (1, 1), (3, 3)):
        ctl[2] = d - slot
        ctl[0] = 10
        ids, chunk, comb, probs = _sample_once(logits, T, end, True, 100, uniforms, ctl, slot, bw, inc=slot > 0)
        assert int(ctl[0]) == (11 if slot > 0 else 10) and int(ctl[2]) == d - slot
        want = torch.softmax(logits.double() / T, dim=1)
        want[:, end] = 0
        assert float((probs[d].double() - want).abs().max()) < 1e-6 and not probs[[i for i in range(draws) if i != d]].any()
        tok = ids - 100
        assert int(tok.min()) >= 0 and int(tok.max()) < V and not (tok == end).any()
        assert torch.equal(chunk[:, slot], ids) and (chunk[:, [s for s in range(bw) if s != slot]] == -7).all()
        cdf = torch.cumsum(want, dim=1)
        target = uniforms[d].double() * cdf[:, -1]
        rows = torch.arange(B, device="cuda")
        hi = cdf[rows, tok]
        lo = hi - want[rows, tok]
        assert ((lo - 1e-6 <= target) & (target < hi + 1e-6)).all(), "token outside its CDF interval"
        assert (want[rows, tok] > 0).all()
        assert float((comb.double() - 0.5 * want[rows, tok]).abs().max()) < 1e-6
    # slot -1: the slot is the device's step counter
    ctl[2], ctl[4] = 1, 2
    ids, chunk, comb, probs = _sample_once(logits, T, end, True, 0, uniforms, ctl, -1, bw)
    assert probs[3].any() and not probs[[0, 1, 2, 4]].any() and torch.equal(chunk[:, 2], ids) and (chunk[:, 3] == -7).all()
    ctl[4] = 0
    # forced tokens replace the draw (entries < 0 do not); train mode keeps <end> in play and maps it to 0
    forced = torch.full((draws, B), -1, dtype=torch.int64, device="cuda")
    forced[2] = torch.tensor([0, 1 % V, end, -1, 2 % V, end], device="cuda")
    ctl[2] = 2
    ids, chunk, comb, probs = _sample_once(logits, T, end, False, 0, uniforms, ctl, 0, bw, forced=forced)
    want = torch.softmax(logits.double() / T, dim=1)
    assert float((probs[2].double() - want).abs().max()) < 1e-6
    assert ids[0] == 0 and ids[1] == 1 % V and ids[2] == 0 and ids[5] == 0      # <end> -> 0
    assert ids[4] == (0 if 2 % V == end else 2 % V)
    assert float((comb[2].double() - 0.5 * want[2, end])).__abs__() < 1e-6           # the product saw p[<end>]


def test_decode_sample_candidate_major_draw_numbers():
    """beams > 0: row image * beams + c draws with the number the reference's candidate loop gives that draw --
    (candidate c) * beam_width + slot behind ctl[2], one column per image (generate_images.py:262-304) -- so the
    candidates run as rows of one batch and still consume the draws of the sequential order."""
    from qarig import ops
    g = torch.Generator().manual_seed(5)
    N, NB, bw, V, end, T = 3, 4, 4, 41, 40, 0.9
    B, draws = N * NB, 2 * NB * bw
    logits = (torch.randn((B, V), generator=g) * 2).cuda()
    uniforms = torch.rand((draws, N), generator=g).cuda()
    ctl = torch.zeros(ops.DECODE_CTL_WORDS, dtype=torch.int32, device="cuda")
    base, slot = NB * bw, 2
    ctl[2] = base
    ids = torch.zeros(B, dtype=torch.int64, device="cuda")
    chunk = torch.zeros((B, bw), dtype=torch.int64, device="cuda")
    comb = torch.ones(B, device="cuda")
    probs = torch.zeros((draws, N, V), device="cuda")
    ops.decode_sample(logits, T, end, True, 0, uniforms, ctl, slot, bw, ids, chunk, comb, probs_log=probs, beams=NB)
    want = torch.softmax(logits.double() / T, dim=1)
    want[:, end] = 0
    cdf = torch.cumsum(want, dim=1)
    used = torch.zeros(draws, dtype=torch.bool)
    for n in range(N):
        for c in range(NB):
            r, d = n * NB + c, base + c * bw + slot
            used[d] = True
            assert float((probs[d, n].double() - want[r]).abs().max()) < 1e-6
            target = float(uniforms[d, n].double() * cdf[r, -1])
            tok = int(ids[r])
            assert float(cdf[r, tok] - want[r, tok]) - 1e-6 <= target < float(cdf[r, tok]) + 1e-6
            assert abs(float(comb[r]) - float(want[r, tok])) < 1e-6
    assert not probs[~used].any()
    # forced entries follow the same numbering
    forced = torch.full((draws, N), -1, dtype=torch.int64, device="cuda")
    for n in range(N):
        for c in range(NB):
            forced[base + c * bw + slot, n] = (7 * n + c) % end
    ops.decode_sample(logits, T, end, True, 0, uniforms, ctl, slot, bw, ids, chunk, comb, forced=forced, beams=NB)
    assert ids.view(N, NB).tolist() == [[(7 * n + c) % end for c in range(NB)] for n in range(N)]
    with pytest.raises(Exception):
        ops.decode_sample(logits[:B - 1], T, end, True, 0, uniforms, ctl, slot, bw, ids[:B - 1], chunk[:B - 1],
                          comb[:B - 1], beams=NB)


def test_decode_sample_frequencies_follow_the_distribution():
    from qarig import ops
    g = torch.Generator().manual_seed(0)
    B, V, bw = 4096, 9, 1
    logits = torch.tensor([0.3, -1.0, 2.0, 0.0, 1.0, -3.0, 0.5, 0.1, 5.0]).repeat(B, 1).cuda()
    uniforms = torch.rand((1, B), generator=g).cuda()
    ctl = torch.zeros(ops.DECODE_CTL_WORDS, dtype=torch.int32, device="cuda")
    ids, _, _, probs = _sample_once(logits, 1.0, V - 1, True, 0, uniforms, ctl, 0, bw)
    p = probs[0, 0].double().cpu()
    p = p / p.sum()
    freq = torch.bincount(ids.cpu(), minlength=V).double() / B
    sigma = torch.sqrt(p * (1 - p) / B)
    assert freq[V - 1] == 0 and ((freq - p).abs() <= 4.5 * sigma + 1e-12).all(), (freq, p)


@pytest.mark.parametrize("N,NB,bw", [(4, 1, 4), (3, 4, 4), (5, 2, 1), (2, 3, 2)])
def test_chunk_bookkeeping_kernels_follow_the_reference_rule(N, NB, bw):
    """decide / rows / commit / advance against generate_images.py:325-345 restated in torch: per image keep the
    earlier candidate unless the new product is larger; the kept chunk's cache rows become every beam's rows;
    its tokens join the sequence; the counters move to the next chunk."""
    from qarig import ops
    g = torch.Generator().manual_seed(N * 10 + NB)
    B, layers, max_len, H, d, cur = N * NB, 3, 12, 2, 8, 5
    R = max(bw - 1, 1)
    kv = torch.randn((layers, 2, B, H, max_len, d), generator=g).cuda()
    staged = torch.zeros((layers, 2, N, H, R, d), device="cuda")
    tokens = torch.zeros((N, max_len + bw), dtype=torch.int64, device="cuda")
    ids = torch.zeros(B, dtype=torch.int64, device="cuda")
    best_p = torch.zeros(N, device="cuda")
    best_chunk = torch.zeros((N, bw), dtype=torch.int64, device="cuda")
    take = torch.zeros(N, dtype=torch.int32, device="cuda")
    ctl = torch.zeros(ops.DECODE_CTL_WORDS, dtype=torch.int32, device="cuda")
    ctl[0], ctl[1], ctl[2] = cur + bw - 1, cur, 40
    ref_p, ref_chunk, ref_rows = None, None, None
    cands = 3 if NB == 1 else 1
    for c in range(cands):
        comb = torch.rand(B, generator=g).cuda()
        if c == 1:
            comb[0] = ref_p[0]            # a tie keeps the earlier candidate
        chunk = torch.randint(0, 50, (B, bw), generator=g).cuda()
        kv[:, :, :, :, cur:cur + bw - 1] = torch.randn((layers, 2, B, H, bw - 1, d), generator=g).cuda()
        cv = comb.view(N, NB)
        pb = cv.argmax(dim=1)
        cbest = cv.max(dim=1).values
        pick = torch.arange(N, device="cuda") * NB + pb
        if ref_p is None:
            tk = torch.ones(N, dtype=torch.bool, device="cuda")
        else:
            tk = ~(ref_p >= cbest)
        new_rows = kv[:, :, pick, :, cur:cur + bw - 1].clone()
        ref_p = cbest.clone() if ref_p is None else torch.where(tk, cbest, ref_p)
        ref_chunk = chunk[pick].clone() if ref_chunk is None else torch.where(tk[:, None], chunk[pick], ref_chunk)
        ref_rows = new_rows if ref_rows is None else torch.where(tk[None, None, :, None, None, None], new_rows, ref_rows)
        per_set = NB * bw if NB == 2 else bw          # (one case counts the draws of a candidate-major set)
        ops.decode_decide(ctl, N, NB, bw, comb, chunk, best_p, best_chunk, take, draws=per_set)
        assert torch.equal(take.bool(), tk) and torch.equal(take[tk].long() - 1, pb[tk])
        assert (comb == 1).all() and int(ctl[3]) == c + 1 and int(ctl[2]) == 40 + (c + 1) * per_set and int(ctl[0]) == cur
        if bw > 1:
            ops.decode_rows(ctl, kv, staged, take, N, NB, restore=False)
        assert torch.equal(best_p, ref_p) and torch.equal(best_chunk, ref_chunk)
        if bw > 1:
            assert torch.equal(staged, ref_rows)
    before = kv.clone()
    if bw > 1:
        ops.decode_rows(ctl, kv, staged, take, N, NB, restore=True)
        want = before.clone()
        want[:, :, :, :, cur:cur + bw - 1] = ref_rows.repeat_interleave(NB, dim=2)
        assert torch.equal(kv, want)
    ops.decode_commit(ctl, N, NB, bw, best_chunk, tokens, ids)
    assert torch.equal(tokens[:, cur:cur + bw], ref_chunk) and not tokens[:, :cur].any() and not tokens[:, cur + bw:].any()
    assert torch.equal(ids, ref_chunk[:, -1].repeat_interleave(NB)) and int(ctl[0]) == cur + bw - 1
    ops.decode_advance(ctl, bw)
    assert ctl[:4].tolist() == [cur + bw, cur + bw, 40 + cands * per_set, 0]
