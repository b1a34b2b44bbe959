"""Pins the oracle (oracle/ref_models.py, oracle/bmu_oracle.c) to the reference:
every fixture in tests/golden/ was produced by the reference's own classes
(oracle/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import grad_err, load_golden, rel_err
from oracle import bmu as obmu
from oracle import ref_models as rm

TOL = 2e-6  # fp32 restatement vs fp32 reference, same torch build: summation-order noise only


def test_patchify_unpatchify_golden():
    g = load_golden("layers")
    for p in (1, 2, 4):
        pt = rm.patchify(g["x"], (p, p))
        assert torch.equal(pt, g[f"patch_p{p}"])
        assert torch.equal(rm.unpatchify(pt, (4, 4), (p, p)), g[f"unpatch_p{p}"])
        assert torch.equal(g[f"unpatch_p{p}"], g["x"])
        assert np.array_equal(obmu.patchify(g["x"].numpy(), (p, p)),
                              g[f"patch_p{p}"].reshape(-1, 2 * p * p).numpy())
    assert torch.equal(rm.patchify(g["xr"], (2, 3)), g["patch_rect"])
    # documented order: channel-major, then row, then column
    assert g["patch_p2"][0, 0].tolist() == [0, 1, 4, 5, 16, 17, 20, 21]


def test_positional_embeddings_golden():
    g = load_golden("layers")
    assert torch.equal(rm.positional_embeddings(32, torch.arange(1, 18)), g["pos_int"])
    assert torch.equal(rm.positional_embeddings(32, torch.arange(0, 300, 7, dtype=torch.float32)),
                       g["pos_float"])
    assert torch.equal(rm.positional_embeddings(512, torch.tensor([1, 2, 255, 256, 1023, 4096])),
                       g["pos_int_512"])


def test_autoencoder_golden():
    g = load_golden("autoencoder")
    sd = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    x = g["x"].clone().requires_grad_(True)
    z = rm.fc_encoder(sd, x, final_act="tanh", prefix="fc_encoder.fc_encoder_layer")
    y = rm.fc_decoder(sd, z, prefix="fc_decoder.fc_decoder_layer")
    assert rel_err(z, g["latent"]) < TOL
    assert rel_err(y, g["recon"]) < TOL
    loss = torch.nn.functional.mse_loss(y, x.detach())
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    assert rel_err(x.grad, g["x_grad"]) < 1e-5
    for k, v in g["grads"].items():
        assert rel_err(sd[k].grad, v) < 1e-5, k
    y2 = rm.autoencoder(g["sd"], g["x"], enc_act="tanh")
    assert rel_err(y2, g["recon"]) < TOL


def test_decoder_wide_golden():
    g = load_golden("decoder_wide")
    assert rel_err(rm.fc_decoder(g["sd"], g["z"]), g["recon"]) < TOL


BMU_CASES = ["trained_p1", "trained_p2", "trained_p4", "trained_p8", "trained_full", "ragged",
             "fresh_p4", "ties", "direct"]


@pytest.mark.parametrize("case", BMU_CASES)
def test_bmu_oracle_golden(case):
    """C oracle vs the reference's indices.  Bit-exact wherever the decision is not
    inside fp32 rounding noise; a differing row must be a reference-noise row: the two
    candidates' exact (double) distances agree to a few fp32 ulps (SURVEY 7 hard part 1)."""
    g = load_golden("bmu")[case]
    p = int(g["p"])
    x, w, ref = g["x"].numpy(), g["w"].numpy(), g["idx"].numpy().reshape(-1)
    got = obmu.bmu(x, w, (p, p))
    diff = np.nonzero(got != ref)[0]
    if case.startswith("trained") or case in ("ragged", "ties", "direct"):
        assert diff.size == 0, f"{diff.size} rows differ on well-separated data"
        return
    # degenerate (fresh-init) codebook: differences allowed only on noise rows
    xp = obmu.patchify(x, (p, p)).astype(np.float64)
    wd = w.astype(np.float64)
    for r in diff:
        da = np.sqrt(((xp[r] - wd[got[r]]) ** 2).sum())
        db = np.sqrt(((xp[r] - wd[ref[r]]) ** 2).sum())
        scale = np.sqrt((xp[r] ** 2).sum()) + 1e-30
        assert abs(da - db) <= 4 * np.finfo(np.float32).eps * scale, (r, da, db)
    assert diff.size <= 0.01 * ref.size


@pytest.mark.parametrize("case", BMU_CASES)
def test_bmu_torch_restatement_golden(case):
    g = load_golden("bmu")[case]
    p = int(g["p"])
    got = rm.codebook_bmu(g["w"], g["x"], (p, p), reshape=True)
    frac = (got != g["idx"]).double().mean()
    assert frac <= (0.01 if case == "fresh_p4" else 0.0)


def test_bmu_tie_lowest_index():
    g = load_golden("bmu")["ties"]
    assert int(g["idx"].reshape(-1)[0]) == 7  # rows 7/40/63 identical -> 7


def test_codebook_golden():
    g = load_golden("codebook")
    w = g["w"].clone().requires_grad_(True)
    x = g["x"]
    assert torch.equal(rm.codebook_bmu(w, x, (2, 2), reshape=True), g["bmu"])
    q = rm.codebook_forward(w, x, (8, 8), (2, 2), 4, use_gaussian=True)
    assert rel_err(q, g["fwd_gauss"]) < TOL
    loss = torch.nn.functional.mse_loss(q, x)
    loss.backward()
    assert rel_err(w.grad, g["w_grad_gauss"]) < 1e-5
    assert rel_err(rm.codebook_quantized_patches(w, x, (2, 2), 4, True), g["patches_gauss"]) < TOL
    w.grad = None
    q2 = rm.codebook_forward(w, x, (8, 8), (2, 2), 4, use_gaussian=False)
    assert torch.equal(q2, g["fwd_hard"])
    q2.square().mean().backward()
    assert rel_err(w.grad, g["w_grad_hard"]) < 1e-6
    assert torch.equal(rm.codebook_quantized_image(w, g["idx"], (8, 8), (2, 2)), g["quant_image"])
    assert torch.equal(rm.codebook_quantized_image(w, g["idx"], (8, 8), (2, 2), False),
                       g["quant_patches"])
    r, seq = 4, []
    for _ in range(6):
        r = rm.decrease_neighbourhood(r)
        seq.append(r)
    assert seq == g["neighbourhood_seq"].tolist() == [3, 2, 1, 1.0, 1.0, 1.0]
    assert rel_err(rm.codebook_forward(w, x, (8, 8), (2, 2), 2, True), g["fwd_gauss_r2"]) < TOL
    # sigma^2 for range 4
    assert abs(rm.neighbourhood_variance(4) - 0.8686) < 1e-4


TCFG = {
    "base": dict(use_encoder=False, use_pos_cond=False),
    "base_pos": dict(use_encoder=False, use_pos_cond=True),
    "encdec": dict(use_encoder=True, use_pos_cond=False),
    "encdec_pos": dict(use_encoder=True, use_pos_cond=True),
}


def tcfg(tag):
    c = dict(num_enc_layers=2, num_dec_layers=2, self_attn_heads=4, cross_attn_heads=2,
             hidden_activation="silu")
    c.update(TCFG[tag])
    return c


@pytest.mark.parametrize("tag", list(TCFG))
def test_transformer_golden(tag):
    g = load_golden("transformer_" + tag)
    cfg = tcfg(tag)
    sd = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    logits = rm.transformer_forward(sd, cfg, g["x_dec"], g.get("x_enc"), g.get("pos"))
    assert rel_err(logits, g["logits"]) < 1e-5
    loss = rm.cross_entropy(logits, g["target"])
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    loss.backward()
    for k, v in g["grads"].items():
        assert grad_err(sd[k].grad, v) < 2e-5, k
    if cfg["use_pos_cond"]:
        lf = rm.transformer_forward(g["sd"], cfg, g["x_dec"], g.get("x_enc"), g["pos"].float())
        assert rel_err(lf, g["logits_float_pos"]) < 1e-5
    # Adam(betas=(0.5,0.999)) step restated
    names = list(g["grads"].keys())
    params = [g["sd"][n].clone() for n in names]
    grads = [g["grads"][n] for n in names]  # isolate the optimiser from gradient noise
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    rm.adam_step(params, grads, m, v, step=1, lr=1e-3)
    for n, p in zip(names, params):
        assert rel_err(p, g["sd_after_adam"][n]) < 1e-5, n


def test_causal_invariance_property():
    g = load_golden("transformer_base")
    cfg = tcfg("base")
    x = g["x_dec"].clone()
    a = rm.transformer_forward(g["sd"], cfg, x)
    x[:, -1] = (x[:, -1] + 1) % 40
    b = rm.transformer_forward(g["sd"], cfg, x)
    assert torch.equal(a[:, :-1], b[:, :-1])
    assert not torch.equal(a[:, -1], b[:, -1])


# --- full train step and generation chunk fixtures (round 2) --------------
def _tiny_cfg(use_enc):
    return dict(use_encoder=use_enc, use_pos_cond=True, num_enc_layers=2 if use_enc else None,
                num_dec_layers=2, self_attn_heads=4, cross_attn_heads=2 if use_enc else None,
                hidden_activation="silu")


@pytest.mark.parametrize("tag", ["base", "encdec"])
def test_train_step_golden_pins_oracle(tag):
    """The oracle's BMU, forward, CE, gradients and Adam against one full reference training step
    (train_quantized_transformer.py:404-508) incl. the window it was taken on."""
    g = load_golden("train_step_" + tag)
    lp, hp = int(g["lr_patch"]), int(g["hr_patch"])
    x = g["fmap"].numpy()
    assert np.array_equal(obmu.bmu(x, g["lr_w"].numpy(), (lp, lp)).reshape(3, -1), g["lr_indices"].numpy())
    assert np.array_equal(obmu.bmu(x, g["hr_w"].numpy(), (hp, hp)).reshape(3, -1), g["hr_indices"].numpy())
    sd = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    logits = rm.transformer_forward(sd, _tiny_cfg(tag == "encdec"), g["hr_input"],
                                    g.get("lr_input"), g["pos"])
    assert rel_err(logits, g["logits"]) < 1e-5
    loss = rm.cross_entropy(logits, g["hr_target"])
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    loss.backward()
    names = list(g["grads"])
    for k in names:
        assert grad_err(sd[k].grad, g["grads"][k]) < 5e-5, k
    with torch.no_grad():
        ps = [sd[k] for k in names]
        rm.adam_step(ps, [p.grad for p in ps], [torch.zeros_like(p) for p in ps],
                     [torch.zeros_like(p) for p in ps], 1, 1e-3)
    for k in names:
        assert rel_err(sd[k], g["sd_after_adam"][k]) < 1e-5, k


@pytest.mark.parametrize("tag", ["base", "encdec"])
def test_generation_golden_is_self_consistent_and_pins_oracle_probabilities(tag):
    """Replays the recorded draws on the oracle model: every probability row the reference
    sampled from must come out of the oracle's logits (softmax(logits / T), <end> zeroed) at the
    recorded window / positions, and the kept chunks must follow the product-of-probabilities
    rule with >= ties."""
    g = load_golden("generation_" + tag)
    base = tag == "base"
    K_lr, K_hr = int(g["K_lr"]), int(g["K_hr"])
    total, nb, bw, sw = int(g["total_seq"]), int(g["num_beam"]), int(g["beam_width"]), int(g["sliding_window"])
    T = float(g["temperature"])
    cfg = _tiny_cfg(not base)
    N = g["first_token"].shape[0]
    hr, pos, start, d = g["first_token"], torch.zeros((N, 1)), 0, 0
    chunk = 0
    with torch.no_grad():
        while hr.shape[1] < total:
            cur = hr.shape[1]
            best_in = best_p = None
            for _ in range(nb):
                comb, t_idx, t_in, t_pos = 1.0, start, hr, pos
                for tok in range(bw):
                    if t_in.shape[1] >= sw:
                        t_idx += 1
                        t_pos = t_pos[:, 1:]
                    logits = rm.transformer_forward(g["sd"], cfg, t_in[:, t_idx:], g.get("lr_input"),
                                                    t_pos)[:, -1, :]
                    probs = torch.softmax(logits / T, dim=1)
                    probs[:, K_hr] = 0.0
                    assert float((probs - g["draw_probs"][d]).abs().max()) < 2e-6, (chunk, tok)
                    nxt = g["draw_tokens"][d][:, None]
                    assert bool((g["draw_probs"][d][torch.arange(N), nxt.squeeze(1)] > 0).all())
                    comb = comb * g["draw_probs"][d][torch.arange(N), nxt.squeeze(1)]
                    d += 1
                    t_in = torch.cat((t_in, nxt + (K_lr if base else 0)), dim=1)
                    t_pos = torch.cat((t_pos, torch.full((N, 1), float(cur + tok + 1))), dim=1)
                if best_p is None:
                    best_in, best_p = t_in, comb
                else:
                    keep = best_p >= comb
                    best_p = torch.where(keep, best_p, comb)
                    best_in = torch.where(keep[:, None], best_in, t_in)
            start, hr, pos = t_idx, best_in, t_pos
            assert torch.equal(hr[:, -bw:], g["kept_chunks"][chunk])
            chunk += 1
    assert d == g["draw_tokens"].shape[0]
    final = hr[:, 1:] - (K_lr if base else 0)
    assert torch.equal(final, g["final_tokens"])
    assert torch.equal(pos.long(), g["final_positions"])
    assert g["final_positions"][0, :3].tolist() != [0, 1, 2]        # the 0, 2, 3, ... numbering quirk
