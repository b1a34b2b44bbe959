"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz.

Runs ONLY in the build container: it imports the reference's own `models` package
from /root/reference (pure torch) and records inputs, weights and
outputs of tiny configurations as data fixtures.  The reference never travels to the
GPU box; the fixtures (data only) do.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from models.layers import patchify, unpatchify, get_positional_embeddings  # noqa: E402
from models.Autoencoder import Autoencoder  # noqa: E402
from models.FC_Encoder import FC_Encoder  # noqa: E402
from models.FC_Decoder import FC_Decoder  # noqa: E402
from models.Codebook import Codebook  # noqa: E402
from models.Transformer import Transformer  # noqa: E402


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = vv.detach().cpu().numpy() if torch.is_tensor(vv) else np.asarray(vv)
        else:
            flat[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def liven(model, std=0.05, seed=11):
    """The AdaLN-Zero scale/shift/gate weights start at exactly 0, which would make
    the conditioning path invisible to a parity check: give them small values."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * std)


def gen_layers():
    x = torch.arange(2 * 2 * 4 * 4, dtype=torch.float32).reshape(2, 2, 4, 4)
    out = {}
    for p in (1, 2, 4):
        pt = patchify(x, (p, p))
        out[f"patch_p{p}"] = pt
        out[f"unpatch_p{p}"] = unpatchify(pt, (4, 4), (p, p))
    xr = torch.arange(1 * 3 * 4 * 6, dtype=torch.float32).reshape(1, 3, 4, 6)
    out["patch_rect"] = patchify(xr, (2, 3))
    out["pos_int"] = get_positional_embeddings(32, torch.arange(1, 18))
    out["pos_float"] = get_positional_embeddings(32, torch.arange(0, 300, 7, dtype=torch.float32))
    out["pos_int_512"] = get_positional_embeddings(512, torch.tensor([1, 2, 255, 256, 1023, 4096]))
    save("layers", x=x, xr=xr, **out)


def gen_autoencoder():
    torch.manual_seed(3)
    cfg = dict(num_layers=2, image_channel=3, min_channel=8, max_channel=16, latent_channel=4,
               hidden_activation_type="silu", use_final_enc_activation=True,
               encoder_activation_type="tanh", use_final_dec_activation=True,
               decoder_activation_type="tanh")
    m = Autoencoder(**cfg)
    g = torch.Generator().manual_seed(0)
    x = (2 * torch.rand((2, 3, 16, 16), generator=g) - 1).requires_grad_(True)
    z = m.get_latent(x)
    y = m.recon_image(z)
    loss = torch.nn.functional.mse_loss(y, x.detach())
    loss.backward()
    grads = {n: p.grad for n, p in m.named_parameters()}
    save("autoencoder", sd=m.state_dict(), x=x, latent=z, recon=y, loss=loss, x_grad=x.grad,
         grads=grads)
    # bare halves built from the same weights through the loader hacks
    enc = FC_Encoder(num_layers=2, image_channel=3, min_channel=8, max_channel=16, latent_channel=4,
                     use_final_activation=True, final_activation_type="tanh")
    enc.custom_load_state_dict(m.state_dict(), ignore_msgs=True)
    dec = FC_Decoder(num_layers=2, image_channel=3, min_channel=8, max_channel=16, latent_channel=4)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        dec.custom_load_state_dict(m.state_dict())
    with torch.no_grad():
        z2 = enc(x)
        y2 = dec(z2)
    assert torch.equal(z2, z) and torch.equal(y2, y)
    # wider decoder, one README-like layer shape mix (channels 32/64) on a 8x8 latent
    torch.manual_seed(4)
    dec2 = FC_Decoder(num_layers=2, image_channel=3, min_channel=32, max_channel=64, latent_channel=4)
    zz = torch.tanh(torch.randn((2, 4, 8, 8), generator=g))
    with torch.no_grad():
        yy = dec2(zz)
    save("decoder_wide", sd=dec2.state_dict(), z=zz, recon=yy)


def gen_bmu():
    g = torch.Generator().manual_seed(1)
    cases = {}

    def run(tag, x, w, p, image_dim):
        cb = Codebook(patch_dim=(p, p), image_dim=image_dim, image_channel=x.shape[1],
                      num_embeddings=w.shape[0], init_neighbour_range=4)
        with torch.no_grad():
            cb.codebook.weight.copy_(w)
            idx = cb.get_patches_bmu(x, reshape=True)
        cases[f"{tag}/x"] = x
        cases[f"{tag}/w"] = w
        cases[f"{tag}/p"] = np.int64(p)
        cases[f"{tag}/idx"] = idx

    lat = torch.tanh(torch.randn((4, 4, 32, 32), generator=g))
    for p, K in ((1, 1024), (2, 512), (4, 512), (8, 128)):
        w = torch.tanh(torch.randn((K, 4 * p * p), generator=g))
        run(f"trained_p{p}", lat, w, p, (32, 32))
    lat16 = torch.tanh(torch.randn((40, 4, 16, 16), generator=g))
    run("trained_full", lat16, torch.tanh(torch.randn((64, 1024), generator=g)), 16, (16, 16))
    # ragged extents: K not a multiple of the tile, odd row count
    lat_odd = torch.tanh(torch.randn((3, 4, 12, 20), generator=g))
    run("ragged", lat_odd, torch.tanh(torch.randn((77, 16), generator=g)), 2, (12, 20))
    # fresh-init codebook: uniform(-1/K, 1/K)  (Codebook.py:44-46) -- degenerate
    K = 512
    run("fresh_p4", lat, (torch.rand((K, 64), generator=g) * 2 - 1) / K, 4, (32, 32))
    # exact ties: duplicated codebook rows -> lowest index must win
    w = torch.tanh(torch.randn((64, 16), generator=g))
    w[40] = w[7]
    w[63] = w[7]
    xt = lat[:1].clone()
    xt[0, :, 0:2, 0:2] = w[7].reshape(4, 2, 2)
    run("ties", xt, w, 2, (32, 32))
    # small-input branch of cdist (<= 25 rows on both sides): direct formula
    run("direct", torch.tanh(torch.randn((1, 4, 4, 4), generator=g)),
        torch.tanh(torch.randn((16, 16), generator=g)), 2, (4, 4))
    save("bmu", **cases)


def gen_codebook():
    g = torch.Generator().manual_seed(2)
    K, p = 48, 2
    cb = Codebook(patch_dim=(p, p), image_dim=(8, 8), image_channel=4, num_embeddings=K,
                  init_neighbour_range=4)
    with torch.no_grad():
        cb.codebook.weight.copy_(torch.tanh(torch.randn((K, 16), generator=g)))
    x = torch.tanh(torch.randn((3, 4, 8, 8), generator=g))
    out = {"x": x, "w": cb.codebook.weight.detach().clone()}
    out["bmu"] = cb.get_patches_bmu(x, reshape=True)
    q = cb(x, use_gaussian=True)
    loss = torch.nn.functional.mse_loss(q, x)
    loss.backward()
    out["fwd_gauss"] = q
    out["loss_gauss"] = loss
    out["w_grad_gauss"] = cb.codebook.weight.grad.clone()
    cb.codebook.weight.grad = None
    out["patches_gauss"] = cb.get_quantized_patches(x, use_gaussian=True)
    q2 = cb(x, use_gaussian=False)
    out["fwd_hard"] = q2
    q2.square().mean().backward()
    out["w_grad_hard"] = cb.codebook.weight.grad.clone()
    idx = torch.randint(0, K, (3, 16), generator=g)
    out["idx"] = idx
    out["quant_image"] = cb.get_quantized_image(idx)
    out["quant_patches"] = cb.get_quantized_image(idx, unpatchify_input=False)
    seq = []
    for _ in range(6):
        cb.decrease_neighbourhood()
        seq.append(cb.neighbourhood_range)
    out["neighbourhood_seq"] = np.asarray(seq, dtype=np.float64)
    cb.neighbourhood_range = 2
    out["fwd_gauss_r2"] = cb(x, use_gaussian=True)
    save("codebook", **out)


def gen_transformer():
    g = torch.Generator().manual_seed(5)
    for tag, use_enc, use_pos in (("base", False, False), ("base_pos", False, True),
                                  ("encdec", True, False), ("encdec_pos", True, True)):
        torch.manual_seed(7)
        cfg = dict(use_encoder=use_enc, use_pos_cond=use_pos, num_enc_layers=2 if use_enc else None,
                   num_dec_layers=2, num_enc_embedding=24 if use_enc else None,
                   num_dec_embedding=40, self_attn_heads=4,
                   cross_attn_heads=2 if use_enc else None, transformer_in_dim=32,
                   transformer_out_dim=33, transformer_hidden_dim=64, hidden_activation="silu")
        m = Transformer(**cfg)
        liven(m)
        N, S, Se = 3, 12, 5
        x_dec = torch.randint(0, 40, (N, S), generator=g)
        x_enc = torch.randint(0, 24, (N, Se), generator=g) if use_enc else None
        pos = None
        if use_pos:
            start = torch.randint(0, 50, (N,), generator=g)
            pos = start[:, None] + torch.arange(S)[None, :]
        target = torch.randint(0, 33, (N, S), generator=g)
        logits = m(x_dec, x_enc, pos)
        loss = torch.nn.functional.cross_entropy(logits.view(N * S, -1), target.flatten())
        loss.backward()
        grads = {n: p.grad.clone() for n, p in m.named_parameters()}
        out = dict(sd={k: v.clone() for k, v in m.state_dict().items()}, x_dec=x_dec,
                   target=target, logits=logits, loss=loss, grads=grads)
        if use_enc:
            out["x_enc"] = x_enc
        if use_pos:
            out["pos"] = pos
            with torch.no_grad():
                out["logits_float_pos"] = m(x_dec, x_enc, pos.float())
        # one Adam step exactly as the reference trains (betas 0.5/0.999)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
        opt.step()
        out["sd_after_adam"] = {k: v.clone() for k, v in m.state_dict().items()}
        save("transformer_" + tag, **out)


def _tiny_transformer(use_enc, n_dec_emb, n_enc_emb, out_dim, seed):
    torch.manual_seed(seed)
    cfg = dict(use_encoder=use_enc, use_pos_cond=True, num_enc_layers=2 if use_enc else None,
               num_dec_layers=2, num_enc_embedding=n_enc_emb if use_enc else None,
               num_dec_embedding=n_dec_emb, self_attn_heads=4,
               cross_attn_heads=2 if use_enc else None, transformer_in_dim=32,
               transformer_out_dim=out_dim, transformer_hidden_dim=64, hidden_activation="silu")
    m = Transformer(**cfg)
    liven(m)
    return m


def _codebook(p, K, g, image_dim=(8, 8)):
    cb = Codebook(patch_dim=(p, p), image_dim=image_dim, image_channel=4, num_embeddings=K,
                  init_neighbour_range=4)
    with torch.no_grad():
        cb.codebook.weight.copy_(torch.tanh(torch.randn((K, 4 * p * p), generator=g)))
    return cb


def gen_train_step():
    """ONE full training step as the reference's loop performs it
    (train_quantized_transformer.py:404-508): BMU tokenisation with both codebooks, token
    assembly (base: LR token + shifted HR ids; enc-dec: <start> + HR ids, LR ids to the
    encoder), <end>-terminated target, unfold + per-sample random window + absolute window
    positions, forward, CE, backward, Adam(0.5, 0.999).  The loop body below follows those
    lines statement by statement on the reference's own Codebook / Transformer classes."""
    g = torch.Generator().manual_seed(21)
    N, window = 3, 8
    fmap = torch.tanh(torch.randn((N, 4, 8, 8), generator=g))
    for tag, base in (("base", True), ("encdec", False)):
        K_lr, K_hr = 16, 32
        lr_cb = _codebook(8 if base else 4, K_lr, g)     # base: one LR token per latent
        hr_cb = _codebook(2, K_hr, g)
        m = _tiny_transformer(not base, K_lr + K_hr if base else K_hr + 1, K_lr, K_hr + 1, seed=31)
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
        ce = torch.nn.CrossEntropyLoss()
        with torch.no_grad():
            lr_indices = lr_cb.get_patches_bmu(fmap, reshape=True)
            hr_indices = hr_cb.get_patches_bmu(fmap, reshape=True)
        if base:
            hr_input = torch.cat((lr_indices, hr_indices + K_lr), dim=1)
            lr_input = None
        else:
            start_tensor = torch.tensor([[K_hr]]).repeat(N, 1)
            hr_input = torch.cat((start_tensor, hr_indices), dim=1)
            lr_input = lr_indices
        end_tensor = torch.tensor([[K_hr]]).repeat(N, 1)
        hr_target = torch.cat((hr_indices, end_tensor), dim=1)
        full_in, full_tg = hr_input.clone(), hr_target.clone()
        hr_input_unfold = hr_input.unfold(dimension=1, size=window, step=1)
        hr_target_unfold = hr_target.unfold(dimension=1, size=window, step=1)
        _, num_sliding_windows, _ = hr_input_unfold.shape
        torch.manual_seed(123)
        rand_indices = torch.randint(low=0, high=num_sliding_windows, size=(N,))
        hr_input = hr_input_unfold[torch.arange(N), rand_indices, :]
        hr_target = hr_target_unfold[torch.arange(N), rand_indices, :]
        sliding_window_indices = rand_indices.unsqueeze(dim=1) + torch.arange(window).unsqueeze(dim=0)
        m.train()
        opt.zero_grad()
        out = m(x_dec=hr_input, x_enc=lr_input, pos_cond=sliding_window_indices)
        _, Seq, C = out.shape
        loss = ce(out.view(N * Seq, C), hr_target.flatten())
        loss.backward()
        grads = {n: p.grad.clone() for n, p in m.named_parameters()}
        opt.step()
        arrays = dict(sd=sd0, fmap=fmap, lr_w=lr_cb.codebook.weight.detach().clone(),
                      hr_w=hr_cb.codebook.weight.detach().clone(), lr_patch=np.int64(8 if base else 4),
                      hr_patch=np.int64(2), window=np.int64(window), lr_indices=lr_indices,
                      hr_indices=hr_indices, full_input=full_in, full_target=full_tg,
                      num_windows=np.int64(num_sliding_windows), rand_indices=rand_indices,
                      hr_input=hr_input, hr_target=hr_target, pos=sliding_window_indices,
                      logits=out, loss=loss, grads=grads,
                      sd_after_adam={k: v.clone() for k, v in m.state_dict().items()})
        if not base:
            arrays["lr_input"] = lr_input
        save("train_step_" + tag, **arrays)


def gen_generation():
    """the sampling loop of generate_images.py:256-345 driven on the reference's
    Transformer at LEGAL temperatures (0.5 and 1.0), num_beam 2 x beam_width 4, sliding window
    8 on 16-token sequences (so the window slides for half of the run), torch.manual_seed(69).
    Every draw is recorded: the probability row the reference sampled from (after the <end>
    zeroing), the token torch.multinomial returned, and the sequences kept after each chunk.
    The loop below follows the reference's statements; the test replays the product's loop with
    the recorded draws injected, which pins softmax(logits / T), the <end> masking, the draw
    order over beams, the vocabulary shift, the 0,2,3,... position numbering, the
    probability-product comparison and the >= tie rule."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(41)
    N, K_lr, K_hr = 3, 16, 32
    total_Seq, num_beam, beam_width, sliding_window = 16, 2, 4, 8
    prev_tokens = torch.randint(0, K_lr, (N, 4), generator=g)          # stage k-1's output
    for tag, index, temperature in (("base", "0", 1.0), ("encdec", "1", 0.5)):
        base = index == "0"
        m = _tiny_transformer(not base, K_lr + K_hr if base else K_hr + 1, K_lr, K_hr + 1,
                              seed=51 if base else 52)
        m.eval()
        torch.manual_seed(69)
        draws_p, draws_t, kept = [], [], []
        with torch.no_grad():
            if base:
                lr_input = None
                hr_input = torch.randint(low=0, high=K_lr, size=(N, 1))
            else:
                lr_input = prev_tokens
                hr_input = torch.tensor([[K_hr]]).repeat(N, 1)
            first = hr_input.clone()
            pos_indices = torch.zeros((N, 1))
            start_index = 0
            _, curr_num_seq = hr_input.shape
            while curr_num_seq < total_Seq:
                best_hr_input = None
                best_combined_prob = None
                for _ in range(num_beam):
                    total_combined_prob = 1.0
                    temp_index = start_index
                    temp_hr_input = hr_input
                    temp_pos_indices = pos_indices
                    for token_count in range(beam_width):
                        _, temp_Seq = temp_hr_input.shape
                        if temp_Seq >= sliding_window:
                            temp_index = temp_index + 1
                            temp_pos_indices = temp_pos_indices[:, 1:]
                        temp_hr_window = temp_hr_input[:, temp_index:]
                        out_seq = m(x_dec=temp_hr_window, x_enc=lr_input, pos_cond=temp_pos_indices)
                        out_seq = out_seq[:, -1, :]
                        probs = F.softmax(out_seq / temperature, dim=1)
                        probs[:, K_hr] = 0.0
                        next_token = torch.multinomial(probs, 1)
                        draws_p.append(probs.clone())
                        draws_t.append(next_token.squeeze(1).clone())
                        next_token_probs = probs[torch.arange(N), next_token.squeeze(dim=1)]
                        total_combined_prob = total_combined_prob * next_token_probs
                        if base:
                            next_token = next_token + K_lr
                        temp_hr_input = torch.cat((temp_hr_input, next_token), dim=1)
                        temp_indices = torch.tensor([[curr_num_seq + token_count + 1]]).repeat(N, 1)
                        temp_pos_indices = torch.cat((temp_pos_indices, temp_indices), dim=1)
                    if best_combined_prob is None:
                        best_hr_input = temp_hr_input
                        best_combined_prob = total_combined_prob
                    else:
                        mask_prob = (best_combined_prob >= total_combined_prob).float()
                        best_combined_prob = (mask_prob * best_combined_prob) + ((1 - mask_prob) * total_combined_prob)
                        mask_seq = mask_prob[:, None]
                        best_hr_input = (mask_seq * best_hr_input) + ((1 - mask_seq) * temp_hr_input)
                start_index = temp_index
                hr_input = best_hr_input.long()
                pos_indices = temp_pos_indices.long()
                _, curr_num_seq = hr_input.shape
                kept.append(hr_input[:, -beam_width:].clone())
            final = hr_input[:, 1:]
            if base:
                final = final - K_lr
        arrays = dict(sd={k: v.clone() for k, v in m.state_dict().items()}, first_token=first,
                      K_lr=np.int64(K_lr), K_hr=np.int64(K_hr), total_seq=np.int64(total_Seq),
                      num_beam=np.int64(num_beam), beam_width=np.int64(beam_width),
                      sliding_window=np.int64(sliding_window), temperature=np.float64(temperature),
                      draw_probs=torch.stack(draws_p), draw_tokens=torch.stack(draws_t),
                      kept_chunks=torch.stack(kept), final_tokens=final,
                      final_positions=pos_indices)
        if not base:
            arrays["lr_input"] = lr_input
        save("generation_" + tag, **arrays)


if __name__ == "__main__":
    torch.set_num_threads(4)
    if "--only-new" in sys.argv:       # round 2 additions (the round-1 fixtures stay byte-identical)
        gen_train_step()
        gen_generation()
        sys.exit(0)
    gen_layers()
    gen_autoencoder()
    gen_bmu()
    gen_codebook()
    gen_transformer()
    gen_train_step()
    gen_generation()
