"""KV-cache decoding: the single-token step must reproduce the last row of the full-window
decoder (reference generate_images.py:283-290 evaluates the whole window per token), and
the cached generation loop must emit the tokens of the full-window loop."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _model(use_encoder, adaln0=False, heads=8, dim=64, hidden=128, layers=2, vocab=41, pos_cond=True):
    from models.Transformer import Transformer
    torch.manual_seed(3)
    kw = dict(use_encoder=use_encoder, use_pos_cond=pos_cond, num_enc_layers=2 if use_encoder else None,
              num_dec_layers=layers, num_enc_embedding=vocab if use_encoder else None,
              num_dec_embedding=vocab, self_attn_heads=heads,
              cross_attn_heads=heads if use_encoder else None, transformer_in_dim=dim,
              transformer_out_dim=vocab, transformer_hidden_dim=hidden)
    m = Transformer(**kw).cuda().eval()
    with torch.no_grad():       # AdaLN-zero style zero inits would hide the conditioning path
        for p in m.parameters():
            if p.abs().max() == 0:
                p.normal_(0, 0.05)
    return m


def test_attention_decode_matches_full_attention():
    from qarig import ops
    torch.manual_seed(0)
    for heads, d, L in ((8, 8, 150), (4, 16, 150), (2, 64, 150), (16, 4, 150), (3, 32, 150), (64, 8, 700), (6, 16, 300)):
        B, D, Lmax = 5, heads * d, L + 10
        q = torch.randn(B, 1, D, device="cuda")
        k = torch.randn(B, L + 1, D, device="cuda")
        v = torch.randn(B, L + 1, D, device="cuda")
        want = ops.attention_fwd(q, k, v, heads, False)[0][:, 0]
        kc = torch.zeros(B, Lmax, D, device="cuda")
        vc = torch.zeros(B, Lmax, D, device="cuda")
        kc[:, :L] = k[:, :L]
        vc[:, :L] = v[:, :L]
        got = ops.attention_decode(q[:, 0].contiguous(), k[:, L].contiguous(), v[:, L].contiguous(),
                                   kc, vc, L, heads)
        assert rel_err(got, want) < 1e-5
        assert torch.equal(kc[:, L], k[:, L]) and torch.equal(vc[:, L], v[:, L])   # appended
        assert not kc[:, L + 1:].any()
        # read-only form (cross-attention) on a strided batch view, and device-side length
        got2 = ops.attention_decode(q[:, 0].contiguous(), None, None, kc, vc, L + 1, heads)
        assert rel_err(got2, want) < 1e-5
        ln = torch.tensor([L + 1], dtype=torch.int32, device="cuda")
        got3 = ops.attention_decode(q[::2, 0].contiguous(), None, None, kc[::2], vc[::2], 0, heads,
                                    len_dev=ln)
        assert rel_err(got3, want[::2]) < 1e-5
        mul = torch.randn(B, D, device="cuda")
        got4 = ops.attention_decode(q[:, 0].contiguous(), None, None, kc, vc, L + 1, heads, o_mul=mul)
        assert rel_err(got4, want * mul) < 1e-5
        got5 = ops.attention_decode(q[:, 0].contiguous(), None, None, kc, vc, L + 1, heads, o_mul=mul[0].contiguous())
        assert rel_err(got5, want * mul[0]) < 1e-5          # one gate row for every sequence
        # head-major cache (B, H, max_len, d) -- DecodeCache's layout: append + attend, then read-only
        hm = lambda t: t.reshape(B, Lmax, heads, d).permute(0, 2, 1, 3).contiguous()
        kh, vh = hm(kc), hm(vc)
        kh[:, :, L], vh[:, :, L] = 0.0, 0.0
        got6 = ops.attention_decode(q[:, 0].contiguous(), k[:, L].contiguous(), v[:, L].contiguous(), kh, vh, L, heads,
                                    o_mul=mul)
        assert rel_err(got6, want * mul) < 1e-5
        assert torch.equal(kh, hm(kc)) and torch.equal(vh, hm(vc))      # the new row landed in every head's run
        got7 = ops.attention_decode(q[:, 0].contiguous(), None, None, kh, vh, 0, heads, len_dev=ln)
        assert rel_err(got7, want) < 1e-5


def test_grouped_skinny_gemm():
    from qarig import ops
    g = torch.Generator().manual_seed(0)
    G, M, N, K = 5, 70, 200, 512
    A = torch.randn((G, M, K), generator=g).cuda()
    W = (torch.randn((G, N, K), generator=g) * 0.1).cuda()
    b = torch.randn((G, N), generator=g).cuda()
    got = ops.gemm_grouped_skinny(A, W, b, act=1)
    want = torch.nn.functional.silu(torch.einsum("gmk,gnk->gmn", A.double().cpu(), W.double().cpu())
                                    + b.double().cpu()[:, None])
    assert got.shape == (G, M, N) and rel_err(got, want) < 5e-6
    for gi in range(G):      # same kernel, same order as the single-problem skinny path
        assert torch.equal(got[gi], ops.gemm(A[gi], W[gi], bias=b[gi], act=1))
    shared = ops.gemm_grouped_skinny(A[0], W, None, shared_a=True)
    assert rel_err(shared, torch.einsum("mk,gnk->gmn", A[0].double().cpu(), W.double().cpu())) < 5e-6
    with pytest.raises(RuntimeError):
        ops.gemm_grouped_skinny(torch.randn(2, 513, 256).cuda(), torch.randn(2, 16, 256).cuda())
    with pytest.raises(RuntimeError):
        ops.gemm_grouped_skinny(torch.randn(2, 8, 96).cuda(), torch.randn(2, 16, 96).cuda())


@pytest.mark.parametrize("M,G,N,K,form,gate", [
    (4, 1, 2048, 512, "adaln", False), (4, 3, 2048, 512, "adaln", False), (16, 1, 512, 2048, "none", True),
    (1, 1, 513, 256, "affine", True), (20, 3, 200, 512, "affine", False), (64, 1, 96, 1024, "adaln", True),
    (100, 2, 48, 256, "adaln", False), (33, 1, 2048, 512, "affine", False), (48, 1, 64, 512, "adaln", True)])
def test_skinny_gemm_with_layernorm_prologue_and_gate(M, G, N, K, form, gate):
    """qarig_gemm_skinny_ln_f32: act(LN(x) W_g^T + b_g) * gate in one launch against fp64 -- both
    LayerNorm forms of the decoder blocks (nn.LayerNorm affine; AdaLN scale(cond) * LN(x) + shift(cond),
    reference models/layers.py:130-153), every row-tile count of the kernel and the 64-row slabs."""
    from qarig import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    x = (torch.randn((M, K), generator=g) * 1.5 + 0.3).cuda()
    W = (torch.randn((G, N, K), generator=g) * 0.05).cuda()
    b = torch.randn((G, N), generator=g).cuda()
    gam, bet = torch.randn(K, generator=g).cuda(), torch.randn(K, generator=g).cuda()
    sc, sh = torch.randn((M, K), generator=g).cuda(), torch.randn((M, K), generator=g).cuda()
    mul = torch.randn((M, N), generator=g).cuda() if gate else None
    kw = {"affine": dict(gamma=gam, beta=bet), "adaln": dict(scale=sc, shift=sh), "none": {}}[form]
    if G == 1:
        got = ops.gemm_skinny_ln(x, W[0], b[0], act=1, mul=mul, **kw)[None]
    else:
        got = ops.gemm_skinny_ln(x, W, b, act=1, **kw)
    xd = x.double()
    h = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5)
    if form == "affine":
        h = h * gam.double() + bet.double()
    elif form == "adaln":
        h = sc.double() * h + sh.double()
    else:
        h = xd
    want = torch.nn.functional.silu(torch.einsum("mk,gnk->gmn", h, W.double()) + b.double()[:, None])
    if gate:
        want = want * mul.double()
    assert got.shape == (G, M, N) and rel_err(got, want) < 5e-6
    # the two-launch form it replaces (same kernels otherwise)
    from qarig import functional as QF
    with torch.no_grad():
        hn = QF.layernorm_affine(x, gam, bet) if form == "affine" else \
            (QF.layernorm_mod(x, sc, sh) if form == "adaln" else x)
        two = ops.gemm_grouped_skinny(hn, W, b, act=1, shared_a=True)
    if gate:
        two = two * mul
    assert rel_err(got, two) < 2e-6
    with pytest.raises(RuntimeError, match="pair"):
        ops.gemm_skinny_ln(x, W, b, gamma=gam)
    with pytest.raises(RuntimeError, match="exclusive"):
        ops.gemm_skinny_ln(x, W, b, gamma=gam, beta=bet, scale=sc, shift=sh)


@pytest.mark.parametrize("use_encoder,pos_cond", [(False, True), (True, True), (True, False)])
def test_decode_step_fused_norm_launches_equal_separate_launches(use_encoder, pos_cond, monkeypatch):
    """The step with LayerNorm / gate folded into the Linear launches (kvcache.FUSE_NORMS) against the
    step with the separate launches and against the full-window decoder; AdaLN blocks (pos_cond) and
    affine LayerNorm blocks."""
    from qarig import kvcache, _lib
    m = _model(use_encoder, heads=32, dim=256, hidden=512, pos_cond=pos_cond)
    B, S = 3, 9
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 41, (B, S), generator=g).cuda()
    pos = torch.rand(B, S, generator=g).cuda() * 20 if pos_cond else None
    outs, calls = {}, {}
    with torch.no_grad():
        enc = m.encode(torch.randint(0, 41, (B, 7), generator=g).cuda()) if use_encoder else None
        for fused in (True, False):
            monkeypatch.setattr(kvcache, "FUSE_NORMS", fused)
            cache = kvcache.DecodeCache(m, enc, B, S, graph=False)
            n0 = _lib.N_CALLS
            outs[fused] = [cache.step(ids[:, t], None if pos is None else pos[:, t], t) for t in range(S)]
            calls[fused] = _lib.N_CALLS - n0
        for t in range(S):
            want = m.decode(ids[:, :t + 1].contiguous(), enc,
                            None if pos is None else pos[:, :t + 1].contiguous())[:, -1]
            assert rel_err(outs[True][t], outs[False][t]) < 2e-6, t
            assert rel_err(outs[True][t], want) < 1e-5, t
    per_layer = (calls[False] - calls[True]) / (S * len(m.decoder_layers))
    assert per_layer >= (4 if use_encoder else 3) - (0 if pos_cond else 1)   # the launches that went away


def test_attention_decode_rejects_bad_length():
    from qarig import ops
    q = torch.randn(2, 32, device="cuda")
    kc = torch.zeros(2, 8, 32, device="cuda")
    with pytest.raises(RuntimeError):
        ops.attention_decode(q, q, q, kc, kc.clone(), 8, 4)      # append past the end
    with pytest.raises(RuntimeError):
        ops.attention_decode(q, None, None, kc, kc.clone(), 0, 4)   # nothing to attend


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("use_encoder", [False, True])
def test_decode_cache_step_matches_full_window(use_encoder, graph, wide):
    """wide: D = 256 / hidden = 512, the shapes on which the step stacks the cond projections
    and the q/k/v MLPs into grouped launches; narrow: the per-module launches."""
    from qarig.kvcache import DecodeCache
    m = _model(use_encoder, heads=32, dim=256, hidden=512) if wide else _model(use_encoder)
    B, S = 3, 12
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 41, (B, S), generator=g).cuda()
    pos = torch.rand(B, S, generator=g).cuda() * 20
    with torch.no_grad():
        enc = m.encode(torch.randint(0, 41, (B, 7), generator=g).cuda()) if use_encoder else None
        cache = DecodeCache(m, enc, B, S, graph=graph)
        assert cache._stacked == wide
        for t in range(S):
            got = cache.step(ids[:, t], pos[:, t], t)
            want = m.decode(ids[:, :t + 1].contiguous(), enc, pos[:, :t + 1].contiguous())[:, -1]
            assert rel_err(got, want) < 1e-5, t
        with pytest.raises(IndexError):
            cache.step(ids[:, 0], pos[:, 0], S)


@pytest.mark.parametrize("sampler", ["torch", "fused", "fused-one-by-one"])
@pytest.mark.parametrize("use_encoder,num_beam,bw,batch_beams,wide", [
    (False, 1, 1, False, False), (False, 3, 4, False, False), (True, 2, 4, False, False),
    (True, 3, 2, True, False), (False, 2, 4, True, False), (True, 2, 4, False, True),
    (True, 3, 2, True, True)])
def test_cached_generation_matches_full_window_loop(use_encoder, num_beam, bw, batch_beams, wide, sampler,
                                                    monkeypatch):
    """The cached loop (then its windowed continuation once the window slides) must emit the tokens of the
    reference-style full-window loop.  sampler "torch": same seed, same torch.multinomial call order;
    "fused": the full-window run's draws are recorded and forced into the in-graph sampler, whose probability
    row must equal the recorded one at every draw -- without batch_beams the candidates of a chunk run as rows of
    one batch under the reference's draw numbers; "fused-one-by-one": the candidates one after the other (what
    more than 16 rows fall back to)."""
    from conftest import DrawTape
    from qarig import sampling
    if sampler == "fused-one-by-one":
        if batch_beams or num_beam == 1:
            pytest.skip("same path as 'fused'")
        monkeypatch.setattr(sampling, "ORDERED_ROWS", 0)
        sampler = "fused"
    m = _model(use_encoder, heads=32, dim=256, hidden=512) if wide else _model(use_encoder)
    with torch.no_grad():
        m.classifier[1].linear_layer[0].bias[40] -= 20.0     # <end> out of the way
    N, total, sw = 3, 24, 16
    g = torch.Generator().manual_seed(4)
    lr_in = torch.randint(0, 40, (N, 6), generator=g).cuda() if use_encoder else None
    first = torch.randint(0, 40, (N, 1), generator=g).cuda()

    def run(cached):
        torch.manual_seed(11)
        return sampling.generate_tokens(m, first, lr_in, total, 0.05, True, sw, end_token=40,
                                        num_beam=num_beam, beam_width=bw, mode="generate",
                                        batch_beams=batch_beams, use_kv_cache=cached, sampler=sampler)
    if sampler == "torch":
        outs = [run(False), run(True)]
    else:
        tape = DrawTape(monkeypatch, tol=2e-5)
        outs = [tape.record(lambda: run(False)), tape.replay(0, lambda: run(True))]
        assert tape.fused_draws > 0
    assert outs[0].shape[1] >= total       # the loop overshoots to 1 + k*beam_width
    assert torch.equal(outs[0], outs[1])


def test_decode_caches_are_kept_per_model_rebound_to_new_inputs_and_follow_the_weights():
    """sampling keeps a model's decode cache (conditioning table, stacked weights, captured step graph, search
    buffers) between generations: a second generation re-binds it to its own encoder input -- same tokens as a
    freshly built cache gives -- and a change of the weights drops it."""
    from qarig import sampling
    m = _model(True)
    with torch.no_grad():
        m.classifier[1].linear_layer[0].bias[40] -= 20.0
    g = torch.Generator().manual_seed(21)
    N = 3
    lr_a, lr_b = (torch.randint(0, 40, (N, 6), generator=g).cuda() for _ in range(2))
    first = torch.randint(0, 40, (N, 1), generator=g).cuda()

    def run(lr):
        """(tokens, the probability row of every draw)"""
        torch.manual_seed(5)
        sampling.FUSED_DEBUG = {"log": True}
        try:
            toks = sampling.generate_tokens(m, first, lr, 24, 0.7, True, 16, end_token=40, num_beam=2, beam_width=4,
                                            mode="generate", use_kv_cache=True, sampler="fused")
            return torch.cat((toks.flatten().float(), sampling.FUSED_DEBUG["probs"].flatten()))
        finally:
            sampling.FUSED_DEBUG = None
    sampling.decode_cache_clear()
    a1 = run(lr_a)
    assert len(sampling._DECODE_CACHES) == 1
    (_, _, cache), = sampling._DECODE_CACHES.values()
    graph = cache._search.g_step
    b_kept = run(lr_b)                                  # re-bound: other encoder memory, same graph
    (_, _, c2), = sampling._DECODE_CACHES.values()
    assert c2 is cache and cache._search.g_step is graph
    assert torch.equal(run(lr_a), a1) and not torch.equal(a1, b_kept)
    sampling.decode_cache_clear()
    assert torch.equal(run(lr_b), b_kept)               # what a fresh cache gives
    (_, _, c3), = sampling._DECODE_CACHES.values()
    with torch.no_grad():
        m.dec_embedding.weight.mul_(1.5)                # the weights move: table, stacked copies and graph are stale
    moved = run(lr_b)
    (_, _, c4), = sampling._DECODE_CACHES.values()
    assert c4 is not c3
    sampling.decode_cache_clear()
    assert torch.equal(run(lr_b), moved)
    sampling.decode_cache_clear()


def test_batches_between_16_and_64_rows_are_generated_in_groups_of_16_rows():
    """5 images x 4 candidates = 20 rows: generated as 4 + 1 images through the model's kept cache -- the same tokens
    as generating the two groups by hand (same generator, same order); 1 x 4 and 16 x 4 rows are not split."""
    from qarig import sampling
    m = _model(True)
    with torch.no_grad():
        m.classifier[1].linear_layer[0].bias[40] -= 20.0
    g = torch.Generator().manual_seed(8)
    lr = torch.randint(0, 40, (16, 6), generator=g).cuda()
    first = torch.randint(0, 40, (16, 1), generator=g).cuda()
    calls = []
    real = sampling._generate_fused

    def spy(model, hr_input, *a, **k):
        calls.append(hr_input.shape[0])
        return real(model, hr_input, *a, **k)

    def run(sl):
        return sampling.generate_tokens(m, first[sl], lr[sl], 20, 0.7, True, 32, end_token=40, num_beam=4,
                                        beam_width=4, mode="generate", use_kv_cache=True, sampler="fused")
    sampling._generate_fused = spy
    try:
        torch.manual_seed(3)
        whole = run(slice(0, 5))
        assert calls == [4, 1]
        torch.manual_seed(3)
        parts = torch.cat((run(slice(0, 4)), run(slice(4, 5))), dim=0)
        assert torch.equal(whole, parts) and whole.shape == (5, 21)
        assert int(whole[:, 1:].min()) >= 0 and int(whole[:, 1:].max()) < 40
        del calls[:]
        run(slice(0, 1))
        run(slice(0, 16))
        assert calls == [1, 16]
    finally:
        sampling._generate_fused = real
        sampling.decode_cache_clear()


def test_generation_with_head_dim_without_a_cache_kernel():
    """heads=4 on a 48-wide model (head dim 12, served zero-padded by the window kernels): the
    generation loop keeps the reference's full-window evaluation instead of the cache (whose kernel takes
    the instantiated head dims) and emits the same tokens whether or not the cache was asked for."""
    from qarig import sampling
    m = _model(True, heads=4, dim=48, hidden=96)
    first = torch.randint(0, 40, (2, 1), generator=torch.Generator().manual_seed(9)).cuda()
    lr_in = torch.randint(0, 40, (2, 5), generator=torch.Generator().manual_seed(10)).cuda()
    assert not sampling._cacheable(m, first, True)
    assert sampling._cacheable(_model(True, heads=4, dim=64, hidden=96), first, True)
    outs = []
    for cached in (False, True):
        torch.manual_seed(5)
        outs.append(sampling.generate_tokens(m, first, lr_in, 12, 0.05, True, 16, end_token=40,
                                             mode="generate", use_kv_cache=cached))
    assert torch.equal(outs[0], outs[1]) and outs[0].shape[1] >= 12


@pytest.mark.parametrize("sampler", ["torch", "fused"])
def test_cached_train_mode_sampling_matches(sampler, monkeypatch):
    from conftest import DrawTape
    from qarig import sampling
    m = _model(False)
    N, total, sw = 2, 20, 32
    first = torch.randint(0, 40, (N, 1), generator=torch.Generator().manual_seed(9)).cuda()

    def run(cached):
        torch.manual_seed(5)
        return sampling.generate_tokens(m, first, None, total, 0.05, True, sw, end_token=40,
                                        mode="train", use_kv_cache=cached, sampler=sampler)
    if sampler == "torch":
        outs = [run(False), run(True)]
    else:
        tape = DrawTape(monkeypatch, tol=2e-5)
        outs = [tape.record(lambda: run(False)), tape.replay(0, lambda: run(True))]
        assert tape.fused_draws > 0
    assert outs[0].shape == (N, total + 1)
    assert torch.equal(outs[0], outs[1])


def test_config3_cascade_three_stages_readme_size_cached_equals_full_window(monkeypatch):
    """BASELINE config 3 as generate_images.py:101-366 runs it: base + 2 encoder-decoder stages at
    README sizes (512 / 2048 / 64 heads, 7 decoder + 5 encoder layers, K = 512), HR patch 8 -> 4 -> 2
    (16 / 64 / 256 tokens), 4 images, num_beam = beam_width = 4, window 256, T = 1.0, each stage
    conditioned on the previous stage's tokens.  The reference's algorithm (full window re-evaluated
    for every token, real torch.multinomial draws under a seed) is run first and every draw recorded;
    the KV-cache loop is then fed the same draws and must see the same probability row at EVERY draw
    (1e-5), keep the same chunks and end with the same tokens; <end> is never emitted."""
    from models.Transformer import Transformer
    from qarig import sampling
    K, N = 512, 4
    seqs = [16, 64, 256]

    def stage_model(s):
        base = s == 0
        torch.manual_seed(100 + s)
        m = Transformer(use_encoder=not base, use_pos_cond=True, num_enc_layers=None if base else 5,
                        num_dec_layers=7, num_enc_embedding=None if base else K,
                        num_dec_embedding=2 * K if base else K + 1, self_attn_heads=64,
                        cross_attn_heads=None if base else 64, transformer_in_dim=512,
                        transformer_out_dim=K + 1, transformer_hidden_dim=2048)
        g = torch.Generator().manual_seed(200 + s)
        with torch.no_grad():
            for p in m.parameters():
                if p.abs().max() == 0:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        return m.cuda().eval()

    from conftest import DrawTape
    tape = DrawTape(monkeypatch)

    def cascade(mode):
        """mode None: the reference's algorithm, every stage recorded as one tape segment; else the cached
        loop with that sampler, every stage replayed from its segment."""
        torch.manual_seed(69)
        prev = torch.randint(0, K, (N, 1), generator=torch.Generator().manual_seed(7)).cuda()
        per_stage, fused = [], 0
        for s in range(3):
            base = s == 0
            m = stage_model(s)
            first = prev if base else torch.full((N, 1), K, dtype=torch.int64, device="cuda")
            run = lambda: sampling.generate_tokens(m, first, None if base else prev, seqs[s], 1.0, True, 256,
                                                   end_token=K, shift=K if base else 0, num_beam=4, beam_width=4,
                                                   mode="generate", use_kv_cache=mode is not None, sampler=mode)
            toks = tape.record(run) if mode is None else tape.replay(s, run, mode)
            fused += tape.fused_draws if mode is not None else 0
            assert toks.shape[1] >= seqs[s]
            prev = (toks[:, 1:] - (K if base else 0))[:, :seqs[s]].contiguous()
            assert int(prev.min()) >= 0 and int(prev.max()) < K, "<end> or an out-of-vocabulary id was emitted"
            per_stage.append(prev.clone())
            del m
        return per_stage, fused

    want, _ = cascade(None)
    assert sum(len(seg) for seg in tape.segments) == sum(seqs) * 4          # num_beam draws per accepted token
    for mode in ("torch", "fused"):
        got, fused = cascade(mode)
        # the fused sampler makes every draw, those of the last chunk of the 256-token stage included, where the
        # window starts to slide (its 4 candidates x 4 tokens: cache steps + one full-window evaluation of 16 rows)
        assert fused == (sum(seqs) * 4 if mode == "fused" else 0)
        for s in range(3):
            assert torch.equal(want[s], got[s]), (mode, s)
