"""The hot loop's host-visible steps against reference-derived fixtures:
 - pipeline.tokenize / pipeline.slide / pipeline.train_step vs ONE full reference training step
   (BMU tokens, token assembly, <start>/<end>, vocabulary shift, window slicing at the recorded
   random offsets, window positions, loss, every gradient, post-Adam weights);
 - sampling.generate_tokens vs the reference's generation loop at legal temperatures, with the
   reference's recorded multinomial draws injected: the probabilities the product samples from
   must equal the reference's at every draw, and the chunks it keeps must be the reference's."""
import pytest
import torch

from conftest import grad_err, load_golden, rel_err

pytestmark = pytest.mark.gpu


def _tiny(use_enc, n_dec_emb, n_enc_emb, out_dim):
    from models.Transformer import Transformer
    return Transformer(use_encoder=use_enc, use_pos_cond=True, num_enc_layers=2 if use_enc else None,
                       num_dec_layers=2, num_enc_embedding=n_enc_emb if use_enc else None,
                       num_dec_embedding=n_dec_emb, self_attn_heads=4,
                       cross_attn_heads=2 if use_enc else None, transformer_in_dim=32,
                       transformer_out_dim=out_dim, transformer_hidden_dim=64, hidden_activation="silu")


def _codebook(w, p, image_dim=(8, 8)):
    from models.Codebook import Codebook
    cb = Codebook(patch_dim=(p, p), image_dim=image_dim, image_channel=4, num_embeddings=w.shape[0],
                  init_neighbour_range=4)
    with torch.no_grad():
        cb.codebook.weight.copy_(w)
    return cb.cuda()


@pytest.mark.parametrize("tag", ["base", "encdec"])
@pytest.mark.parametrize("table", [False, True])
def test_train_step_matches_reference_step(tag, table):
    from qarig import functional as QF
    from qarig import ops, pipeline
    from qarig.optim import FlatAdam
    g = load_golden("train_step_" + tag)
    base = tag == "base"
    K_lr, K_hr = g["lr_w"].shape[0], g["hr_w"].shape[0]
    lr_cb, hr_cb = _codebook(g["lr_w"], int(g["lr_patch"])), _codebook(g["hr_w"], int(g["hr_patch"]))
    m = _tiny(not base, K_lr + K_hr if base else K_hr + 1, K_lr, K_hr + 1)
    m.custom_load_state_dict(g["sd"])
    m = m.cuda()
    hr_in, lr_in, hr_tg = pipeline.tokenize(g["fmap"].cuda(), lr_cb, hr_cb, train_base_model=base)
    assert torch.equal(hr_in.cpu(), g["full_input"]) and torch.equal(hr_tg.cpu(), g["full_target"])
    if base:
        assert lr_in is None
    else:
        assert torch.equal(lr_in.cpu(), g["lr_input"])
    window = int(g["window"])
    assert pipeline.num_windows(hr_in.shape[1], window) == int(g["num_windows"])
    w_in, w_tg, pos = pipeline.slide(hr_in, hr_tg, window, g["rand_indices"])
    assert torch.equal(w_in.cpu(), g["hr_input"]) and torch.equal(w_tg.cpu(), g["hr_target"])
    assert torch.equal(pos.cpu(), g["pos"])
    # the fused form (one kernel from the BMU indices to the window) gives the same tensors
    f_in, f_lr, f_tg, f_pos = pipeline.tokenize_window(g["fmap"].cuda(), lr_cb, hr_cb, base, window,
                                                       g["rand_indices"])
    assert torch.equal(f_in, w_in) and torch.equal(f_tg, w_tg) and torch.equal(f_pos, pos)
    assert (f_lr is None) if base else torch.equal(f_lr, lr_in)
    n_in, _, n_tg, n_pos = pipeline.tokenize_window(g["fmap"].cuda(), lr_cb, hr_cb, base, None, None)
    assert torch.equal(n_in, hr_in) and torch.equal(n_tg, hr_tg) and n_pos is None
    opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
    old = (QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO)
    try:
        QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO = table, 0
        with torch.no_grad():
            logits = m(w_in, lr_in, pos, pos_bound=hr_in.shape[1])
        assert m._last_cond_form == ("table" if table else "per_token")
        assert rel_err(logits, g["logits"]) < 1e-5
        # gradients before the optimiser consumes them
        opt.zero_grad()
        QF.cross_entropy(m(w_in, lr_in, pos, pos_bound=hr_in.shape[1]).view(-1, K_hr + 1),
                         w_tg.flatten()).backward()
        for n, p in m.named_parameters():
            assert grad_err(p.grad, g["grads"][n]) < 5e-5, n
        loss = pipeline.train_step(m, opt, w_in, lr_in, w_tg, pos, pos_bound=hr_in.shape[1])
    finally:
        QF.USE_COND_TABLE, QF.COND_TABLE_MIN_RATIO = old
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for k, v in m.state_dict().items():
        assert rel_err(v, g["sd_after_adam"][k]) < 1e-5, k
    ops.check_index_flag(torch.device("cuda"), "train step golden")


@pytest.mark.parametrize("tag", ["base", "encdec"])
@pytest.mark.parametrize("loop", ["window", "cache-torch", "cache-fused"])
def test_generation_loop_matches_reference_draw_by_draw(tag, loop, monkeypatch):
    """The reference's sampling loop (generate_images.py:256-345) replayed draw by draw: full-window loop,
    KV-cache loop with one torch.multinomial call per token, KV-cache loop with the fused in-graph sampler
    (the reference's recorded tokens forced, its probability rows compared with the kernel's log)."""
    from conftest import DrawTape
    from qarig import sampling
    g = load_golden("generation_" + tag)
    base = tag == "base"
    K_lr, K_hr = int(g["K_lr"]), int(g["K_hr"])
    m = _tiny(not base, K_lr + K_hr if base else K_hr + 1, K_lr, K_hr + 1)
    m.custom_load_state_dict(g["sd"])
    m = m.cuda().eval()
    tape = DrawTape(monkeypatch)
    tape.add_segment(g["draw_probs"], g["draw_tokens"])
    sampler = "fused" if loop == "cache-fused" else "torch"
    got = tape.replay(0, lambda: sampling.generate_tokens(
        m, g["first_token"].cuda(), None if base else g["lr_input"].cuda(), int(g["total_seq"]),
        float(g["temperature"]), True, int(g["sliding_window"]), end_token=K_hr,
        shift=K_lr if base else 0, num_beam=int(g["num_beam"]), beam_width=int(g["beam_width"]),
        mode="generate", use_kv_cache=loop != "window", sampler=sampler), sampler)
    if loop == "cache-fused":
        assert tape.fused_draws > 0, "the fused sampler never ran"
    final = got[:, 1:].cpu() - (K_lr if base else 0)
    assert torch.equal(final, g["final_tokens"])
    bw = int(g["beam_width"])
    for c in range(g["kept_chunks"].shape[0]):
        assert torch.equal(got[:, 1 + c * bw:1 + (c + 1) * bw].cpu(), g["kept_chunks"][c])
