"""End-to-end runs of the re-hosted command lines on the GPU (tiny shapes): autoencoder
training -> latents -> two codebooks -> base + encoder-decoder Transformer training (with
the periodic checkpoint + autoregressive sample) -> cascade generation.  Checks the
reference's output tree and checkpoint dict schemas."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG

pytestmark = pytest.mark.gpu


def run(script, *args, cwd):
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(PKG, script), *map(str, args)], cwd=cwd, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{script} failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r.stdout + r.stderr


def run2(script, *args, cwd, port=29633):
    """The script under torchrun with two ranks (gloo: both on the test box's one GPU; production: one GPU per
    rank over RCCL)."""
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""), QARIG_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(PKG, script),
                        *map(str, args)], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"2-rank {script} failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r.stdout + r.stderr


def test_cli_pipeline(tmp_path):
    from PIL import Image
    sys.path.insert(0, PKG)
    from dataset_loader._tinydb_json import write_all
    from utils.model_utils import load_model
    from models.FC_Encoder import FC_Encoder
    t = str(tmp_path)
    rng = np.random.default_rng(0)
    recs = []
    for i in range(12):
        p = os.path.join(t, f"img{i}.png")
        Image.fromarray(rng.integers(0, 255, (32, 32, 3), dtype=np.uint8)).save(p)
        recs.append({"image_fpath": p})
    write_all(os.path.join(t, "dataset.json"), recs)
    ae_cfg = dict(model_lr=1e-3, num_layers=2, image_channel=3, min_channel=8, max_channel=16,
                  latent_channel=4, hidden_activation_type="silu", use_final_enc_activation=True,
                  encoder_activation_type="tanh", use_final_dec_activation=True,
                  decoder_activation_type="tanh")
    json.dump(ae_cfg, open(os.path.join(t, "ae.json"), "w"))
    out = run("train_autoencoder.py", "--device", "cuda", "--dataset-path", f"{t}/dataset.json",
              "--batch-size", 4, "--checkpoint-step", 2, "--max-epoch", 1, "--config-path",
              f"{t}/ae.json", "--out-dir", f"{t}/ae", cwd=t)
    assert "Cum. Steps: 3 | Steps: 3 / 3" in out
    ok, ae = load_model(f"{t}/ae/models_checkpoint/model_2.pt")
    assert ok and set(ae) == set(ae_cfg) - {"model_lr"} | {"model", "model_optimizer"}
    assert os.path.exists(f"{t}/ae/images/recon_2.jpg") and os.path.exists(f"{t}/ae/Autoencoder.log")

    # latents with the trained encoder, through the re-hosted generate_fmap_dataset.py
    out = run("generate_fmap_dataset.py", "--device", "cuda", "--batch-size", 5, "--num-files-folder",
              8, "--dataset-path", f"{t}/dataset.json", "--model-path",
              f"{t}/ae/models_checkpoint/model_2.pt", "--out-dir", f"{t}/fmaps", cwd=t)
    assert "3 / 3" in out and os.path.exists(f"{t}/fmaps/1/8")     # folder roll-over at 8 files
    from dataset_loader._tinydb_json import read_all
    frecs = read_all(f"{t}/fmaps/all_dataset.json")
    assert len(frecs) == 12 and frecs[0]["image_path"] == recs[0]["image_fpath"]
    z0 = np.load(frecs[0]["fmap_path"])
    assert z0.shape == (4, 8, 8) and z0.dtype == np.float32
    enc = FC_Encoder(num_layers=2, image_channel=3, min_channel=8, max_channel=16, latent_channel=4,
                     use_final_activation=True, final_activation_type="tanh")
    enc.custom_load_state_dict(ae["model"], ignore_msgs=True)
    from dataset_loader.image_dataset import ImageDataset
    with torch.no_grad():
        zz = enc.cuda()(ImageDataset(f"{t}/dataset.json")[0][None].cuda())[0].cpu().numpy()
    assert np.array_equal(zz, z0)
    # ... and against the ORACLE encoder (fp64) on the same decoded image: the dump is the
    # reference's encoder output, not merely self-consistent
    from oracle import ref_models as rm
    x0 = ImageDataset(f"{t}/dataset.json")[0][None]
    sd64 = {k: v.double() for k, v in ae["model"].items()}
    zref = rm.fc_encoder(sd64, x0.double(), final_act="tanh", prefix="fc_encoder.fc_encoder_layer")[0]
    assert float((torch.from_numpy(z0).double() - zref).abs().max() / zref.abs().max()) < 1e-5
    # two ranks, a contiguous share of the files each: the same files under the same numbers, one index
    run2("generate_fmap_dataset.py", "--device", "cuda", "--batch-size", 5, "--num-files-folder", 8, "--dataset-path",
         f"{t}/dataset.json", "--model-path", f"{t}/ae/models_checkpoint/model_2.pt", "--out-dir", f"{t}/fmaps_w2", cwd=t)
    frecs2 = read_all(f"{t}/fmaps_w2/all_dataset.json")
    assert [r["image_path"] for r in frecs2] == [r["image_path"] for r in frecs]
    for a_, b_ in zip(frecs, frecs2):
        assert os.path.relpath(a_["fmap_path"], f"{t}/fmaps") == os.path.relpath(b_["fmap_path"], f"{t}/fmaps_w2")
        assert open(a_["fmap_path"], "rb").read() == open(b_["fmap_path"], "rb").read()
    os.replace(f"{t}/fmaps/all_dataset.json", f"{t}/fmaps.json")

    for name, p, k in (("lr", 8, 8), ("mid", 2, 16), ("hr", 1, 16)):
        json.dump(dict(model_lr=1e-2, neighbourhood_step=2, image_H=8, image_W=8, image_C=4, patch_H=p,
                       patch_W=p, num_embeddings=k), open(f"{t}/cb_{name}.json", "w"))
        out = run("train_codebook.py", "--device", "cuda", "--dataset-path", f"{t}/fmaps.json",
                  "--decoder-path", f"{t}/ae/models_checkpoint/model_2.pt", "--batch-size", 4,
                  "--checkpoint-step", 2, "--max-epoch", 1, "--config-path", f"{t}/cb_{name}.json",
                  "--out-dir", f"{t}/cb_{name}", cwd=t)
        assert "Neighbourhood Range" in out
        ok, cb = load_model(f"{t}/cb_{name}/models_checkpoint/codebook_2.pt")
        assert ok and set(cb) == {"patch_dim", "image_dim", "image_C", "num_embeddings",
                                  "neighbourhood_range", "global_steps", "checkpoint"}
        assert cb["global_steps"] == 2 and list(cb["checkpoint"]) == ["codebook.weight"]

    # prune: a device histogram of BMU indices; units below the threshold are dropped
    out = run("prune_codebook.py", "--device", "cuda", "--dataset-path", f"{t}/fmaps.json",
              "--codebook-path", f"{t}/cb_mid/models_checkpoint/codebook_2.pt", "--batch-size", 4,
              "--prune-threshold", 3, "--out-dir", f"{t}/prune", cwd=t)
    ok, pr = load_model(f"{t}/prune/models_checkpoint/pruned_codebook.pt")
    counts = [int(l.split(": ")[1].replace(",", "")) for l in out.splitlines()
              if ": " in l and l.split(": ")[0].isdigit()]
    assert ok and len(counts) == 16 and sum(counts) == 12 * 16
    assert pr["num_embeddings"] == sum(c >= 3 for c in counts) \
        == pr["checkpoint"]["codebook.weight"].shape[0]
    # the histogram against the oracle: np.bincount of the C oracle's BMU indices over the dump
    from oracle import bmu as obmu
    ok, cbm = load_model(f"{t}/cb_mid/models_checkpoint/codebook_2.pt")
    wmid = cbm["checkpoint"]["codebook.weight"].numpy()
    fm_all = np.stack([np.load(r["fmap_path"]) for r in read_all(f"{t}/fmaps.json")])
    want = np.bincount(obmu.bmu(fm_all, wmid, (2, 2)), minlength=16)
    assert counts == want.tolist()
    assert np.array_equal(pr["checkpoint"]["codebook.weight"].numpy(), wmid[want >= 3])
    # two ranks: a share of the files each, one all-reduce of the histogram; rank 0 prints and writes the same
    out2 = run2("prune_codebook.py", "--device", "cuda", "--dataset-path", f"{t}/fmaps.json", "--codebook-path",
                f"{t}/cb_mid/models_checkpoint/codebook_2.pt", "--batch-size", 4, "--prune-threshold", 3, "--out-dir",
                f"{t}/prune_w2", cwd=t, port=29634)
    counts2 = [int(l.split(": ")[1].replace(",", "")) for l in out2.splitlines()
               if ": " in l and l.split(": ")[0].isdigit()]
    ok, pr2 = load_model(f"{t}/prune_w2/models_checkpoint/pruned_codebook.pt")
    assert ok and counts2 == counts
    assert np.array_equal(pr2["checkpoint"]["codebook.weight"].numpy(), pr["checkpoint"]["codebook.weight"].numpy())

    tcfg = dict(model_lr=1e-3, num_enc_layers=1, num_dec_layers=2, cross_attn_heads=2,
                self_attn_heads=4, in_dim=32, hidden_dim=64, hidden_activation="silu",
                use_sliding_window=False, sliding_window=None)
    json.dump(tcfg, open(f"{t}/t_base.json", "w"))
    common = ["--device", "cuda", "--dataset-path", f"{t}/fmaps.json", "--decoder-path",
              f"{t}/ae/models_checkpoint/model_2.pt", "--batch-size", 4, "--checkpoint-step", 2,
              "--max-epoch", 1, "--test-num-sample", 3]
    out = run("train_quantized_transformer.py", *common, "--train-base-model", "--lr-codebook-path",
              f"{t}/cb_lr/models_checkpoint/codebook_2.pt", "--hr-codebook-path",
              f"{t}/cb_mid/models_checkpoint/codebook_2.pt", "--config-path", f"{t}/t_base.json",
              "--out-dir", f"{t}/t_base", cwd=t)
    assert "Cum. Steps: 3 | Steps: 3 / 3" in out and "16 / 16" in out
    ok, md = load_model(f"{t}/t_base/models_checkpoint/model_2.pt")
    assert ok and md["train_base_model"] is True and md["num_dec_embedding"] == 8 + 16
    assert md["transformer_out_dim"] == 17 and "model_optimizer" in md
    for f in ("ground_truth_2", "low_res_cond_2", "high_res_example_2", "high_res_recon_2"):
        assert os.path.exists(f"{t}/t_base/images/{f}.jpg")
    # resume with the saved optimiser state
    run("train_quantized_transformer.py", *common, "--train-base-model", "--lr-codebook-path",
        f"{t}/cb_lr/models_checkpoint/codebook_2.pt", "--hr-codebook-path",
        f"{t}/cb_mid/models_checkpoint/codebook_2.pt", "--config-path", f"{t}/t_base.json",
        "--out-dir", f"{t}/t_base2", "--model-path", f"{t}/t_base/models_checkpoint/model_2.pt",
        "--load-optim", "--max-steps", 1, cwd=t)

    # the same run replayed from a captured HIP graph (3 steps: 2 eager warm-ups + 1 replay)
    out_g = run("train_quantized_transformer.py", *common, "--train-base-model", "--lr-codebook-path",
                f"{t}/cb_lr/models_checkpoint/codebook_2.pt", "--hr-codebook-path",
                f"{t}/cb_mid/models_checkpoint/codebook_2.pt", "--config-path", f"{t}/t_base.json",
                "--out-dir", f"{t}/t_base_graph", "--graph-step", cwd=t)
    assert "Cum. Steps: 3 | Steps: 3 / 3" in out_g
    final = float([ln for ln in out_g.splitlines() if "Cum. Steps: 3 |" in ln][-1].split("Recon Loss:")[1])
    assert 0.0 < final < 10.0          # (weights are drawn unseeded: only sanity here; bit-equality
    #                                     with the eager step is tests/test_gpu_dp.py's graphed test)

    # --gemm-x3 (additive): the flag reaches the library option (the tiny products of this run stay on the fp32
    # kernels; the kernel's parity is tests/test_gpu_switches.py)
    out_x = run("train_quantized_transformer.py", *common, "--train-base-model", "--lr-codebook-path",
                f"{t}/cb_lr/models_checkpoint/codebook_2.pt", "--hr-codebook-path",
                f"{t}/cb_mid/models_checkpoint/codebook_2.pt", "--config-path", f"{t}/t_base.json",
                "--out-dir", f"{t}/t_base_x3", "--gemm-x3", "--max-steps", 1, cwd=t)
    assert "Cum. Steps: 1" in out_x

    tcfg2 = dict(tcfg, use_sliding_window=True, sliding_window=32)
    json.dump(tcfg2, open(f"{t}/t_s1.json", "w"))
    out = run("train_quantized_transformer.py", *common, "--lr-codebook-path",
              f"{t}/cb_mid/models_checkpoint/codebook_2.pt", "--hr-codebook-path",
              f"{t}/cb_hr/models_checkpoint/codebook_2.pt", "--config-path", f"{t}/t_s1.json",
              "--out-dir", f"{t}/t_s1", "--use-activation-checkpoint", cwd=t)
    assert "64 / 64" in out
    ok, md1 = load_model(f"{t}/t_s1/models_checkpoint/model_2.pt")
    assert ok and md1["train_base_model"] is False and md1["num_enc_embedding"] == 16
    assert md1["num_dec_embedding"] == 17 and md1["sliding_window"] == 32

    gen = {"0": dict(model_path=f"{t}/t_base/models_checkpoint/model_2.pt", temperature=1.0,
                     lr_codebook_path=f"{t}/cb_lr/models_checkpoint/codebook_2.pt",
                     hr_codebook_path=f"{t}/cb_mid/models_checkpoint/codebook_2.pt", num_beam=2,
                     beam_width=4),
           "1": dict(model_path=f"{t}/t_s1/models_checkpoint/model_2.pt", temperature=0.9,
                     lr_codebook_path=f"{t}/cb_mid/models_checkpoint/codebook_2.pt",
                     hr_codebook_path=f"{t}/cb_hr/models_checkpoint/codebook_2.pt", num_beam=2,
                     beam_width=8)}
    json.dump(gen, open(f"{t}/gen.json", "w"))
    out = run("generate_images.py", "--device", "cuda", "--decoder-path",
              f"{t}/ae/models_checkpoint/model_2.pt", "--num-images", 3, "--seed", 69,
              "--config-path", f"{t}/gen.json", "--out-dir", f"{t}/gen", cwd=t)
    assert "Model: 0" in out and "Model: 1" in out and "64 / 64" in out
    for f in ("recon_model_Cond", "recon_model_0", "recon_model_1"):
        assert os.path.exists(f"{t}/gen/images/{f}.jpg")
    # same seed -> same images (deterministic kernels + seeded sampler)
    run("generate_images.py", "--device", "cuda", "--decoder-path",
        f"{t}/ae/models_checkpoint/model_2.pt", "--num-images", 3, "--seed", 69, "--config-path",
        f"{t}/gen.json", "--out-dir", f"{t}/gen2", cwd=t)
    a = open(f"{t}/gen/images/recon_model_1.jpg", "rb").read()
    b = open(f"{t}/gen2/images/recon_model_1.jpg", "rb").read()
    assert a == b
    # two ranks (gloo: both on the test box's one GPU; production: one GPU each over RCCL): the images are sharded
    # 2 + 1, rank 0 writes the same three grids of all three images; deterministic for a seed
    grids = []
    for tag in ("gen_w2a", "gen_w2b"):
        run2("generate_images.py", "--device", "cuda", "--decoder-path", f"{t}/ae/models_checkpoint/model_2.pt",
             "--num-images", "3", "--seed", "69", "--config-path", f"{t}/gen.json", "--out-dir", f"{t}/{tag}", cwd=t,
             port=29631)
        for f in ("recon_model_Cond", "recon_model_0", "recon_model_1"):
            assert os.path.exists(f"{t}/{tag}/images/{f}.jpg")
        grids.append(open(f"{t}/{tag}/images/recon_model_1.jpg", "rb").read())
    assert grids[0] == grids[1]
    assert Image.open(f"{t}/gen_w2a/images/recon_model_1.jpg").size == Image.open(f"{t}/gen/images/recon_model_1.jpg").size
    # --device cpu is refused loudly
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, "generate_images.py"), "--device", "cpu",
                        "--decoder-path", "x", "--config-path", f"{t}/gen.json", "--out-dir", t],
                       capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "MI355X only" in (r.stdout + r.stderr)


def _oracle_generate(fwd, hr_input, lr_input, total_seq, temperature, use_sw, sw, k_hr, shift,
                     num_beam, beam_width):
    """generate_images.py:256-345 restated on the CPU oracle model (test infrastructure)."""
    N = hr_input.shape[0]
    pos = torch.zeros((N, 1)) if use_sw else None
    start = 0
    cur = hr_input.shape[1]
    while cur < total_seq:
        best_in = best_p = None
        for _ in range(num_beam):
            comb = 1.0
            t_idx, t_in, t_pos = start, hr_input, pos
            for tok in range(beam_width):
                if use_sw and t_in.shape[1] >= sw:
                    t_idx += 1
                    t_pos = t_pos[:, 1:]
                logits = fwd(t_in[:, t_idx:], lr_input, t_pos)[:, -1, :]
                probs = torch.softmax(logits / temperature, dim=1)
                probs[:, k_hr] = 0.0
                nxt = torch.multinomial(probs, 1)
                comb = comb * probs[torch.arange(N), nxt.squeeze(1)]
                t_in = torch.cat((t_in, nxt + shift), dim=1)
                if use_sw:
                    t_pos = torch.cat((t_pos, torch.tensor([[cur + tok + 1]]).repeat(N, 1)), dim=1)
            if best_p is None:
                best_in, best_p = t_in, comb
            else:
                m = (best_p >= comb).float()
                best_p = m * best_p + (1 - m) * comb
                best_in = m[:, None] * best_in + (1 - m[:, None]) * t_in
        start = t_idx
        hr_input = best_in.long()
        if use_sw:
            pos = t_pos.long()
        cur = hr_input.shape[1]
    return hr_input


@pytest.mark.parametrize("use_kv_cache", [False, True])
def test_generation_loop_real_model_matches_oracle_restatement(use_kv_cache):
    """At a near-zero temperature the sampler is an argmax, so the GPU generation loop
    (device sampler, real HIP model) must emit exactly the tokens of the reference's loop
    restated on the CPU oracle model."""
    from conftest import load_golden
    from models.Transformer import Transformer
    from oracle import ref_models as rm
    from qarig import sampling
    g = load_golden("transformer_base_pos")
    g["sd"]["classifier.1.linear_layer.0.bias"][32] -= 10.0   # <end> never the argmax
    m = Transformer(use_encoder=False, use_pos_cond=True, num_enc_layers=None, num_dec_layers=2,
                    num_enc_embedding=None, num_dec_embedding=40, self_attn_heads=4,
                    cross_attn_heads=None, transformer_in_dim=32, transformer_out_dim=33,
                    transformer_hidden_dim=64)
    m.custom_load_state_dict(g["sd"])
    m = m.cuda().eval()
    cfg = dict(use_encoder=False, use_pos_cond=True, num_dec_layers=2, self_attn_heads=4,
               hidden_activation="silu")
    N, total, sw, T = 3, 24, 10, 1e-4
    first = torch.randint(0, 7, (N, 1), generator=torch.Generator().manual_seed(5))
    fwd = lambda x, lr, pos: rm.transformer_forward(g["sd"], cfg, x, lr, pos)
    torch.manual_seed(0)
    want = _oracle_generate(fwd, first, None, total, T, True, sw, 32, 7, 2, 4)
    torch.manual_seed(0)
    got = sampling.generate_tokens(m, first.cuda(), None, total, T, True, sw, end_token=32, shift=7,
                                   num_beam=2, beam_width=4, mode="generate",
                                   use_kv_cache=use_kv_cache)
    assert got.shape == want.shape == (N, 25)
    assert torch.equal(got.cpu(), want)


class _FakeModel:
    """Stand-in with integer, well-separated logits that depend on everything the loop is
    responsible for (window content and length, last position index, encoder input): pins
    the loop logic without being hostage to near-ties of an untrained network."""

    def __init__(self, use_encoder):
        self.use_encoder = use_encoder

    def encode(self, x):
        return x

    def decode(self, x_dec, enc, pos):
        n, s = x_dec.shape
        key = x_dec.sum(1) * 3 + s
        if pos is not None:
            assert pos.shape == (n, s)
            key = key + pos[:, -1].long() * 5 + pos[:, 0].long()
        if enc is not None:
            key = key + enc.sum(1)
        v = torch.arange(33, device=x_dec.device)
        logits = -(((v[None, :] * 7 + key[:, None]) % 33).float())
        logits[:, 32] = -1000.0    # <end> never the argmax (its probability gets zeroed)
        return logits[:, None, :].expand(n, s, 33)


@pytest.mark.parametrize("use_enc,use_sw,num_beam,bw", [(False, True, 2, 4), (True, True, 3, 8),
                                                        (True, False, 1, 4), (False, False, 2, 2)])
def test_generation_loop_logic_matches_reference_restatement(use_enc, use_sw, num_beam, bw):
    from qarig import sampling
    fake = _FakeModel(use_enc)
    N, total, sw, T = 4, 24, 10, 1e-4
    gen = torch.Generator().manual_seed(1)
    lr_in = torch.randint(0, 24, (N, 5), generator=gen) if use_enc else None
    shift = 0 if use_enc else 7
    first = torch.full((N, 1), 32) if use_enc else torch.randint(0, 7, (N, 1), generator=gen)
    fwd = lambda x, lr, pos: fake.decode(x, lr, pos)
    torch.manual_seed(0)
    want = _oracle_generate(fwd, first, lr_in, total, T, use_sw, sw, 32, shift, num_beam, bw)
    torch.manual_seed(0)
    got = sampling.generate_tokens(fake, first.cuda(), lr_in.cuda() if use_enc else None, total, T,
                                   use_sw, sw, end_token=32, shift=shift, num_beam=num_beam,
                                   beam_width=bw, mode="generate")
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("use_enc,use_sw", [(False, True), (True, True), (True, False)])
def test_batched_beams_match_sequential_search_when_deterministic(use_enc, use_sw):
    """With argmax sampling every candidate chunk is identical, so the batched-beam
    variant must reproduce the sequential loop's tokens exactly."""
    from qarig import sampling
    fake = _FakeModel(use_enc)
    N, total, sw, T = 3, 24, 10, 1e-4
    gen = torch.Generator().manual_seed(2)
    lr_in = torch.randint(0, 24, (N, 5), generator=gen).cuda() if use_enc else None
    shift = 0 if use_enc else 7
    first = (torch.full((N, 1), 32) if use_enc else torch.randint(0, 7, (N, 1), generator=gen)).cuda()
    a = sampling.generate_tokens(fake, first, lr_in, total, T, use_sw, sw, end_token=32, shift=shift,
                                 num_beam=3, beam_width=4, mode="generate")
    b = sampling.generate_tokens(fake, first, lr_in, total, T, use_sw, sw, end_token=32, shift=shift,
                                 num_beam=3, beam_width=4, mode="generate", batch_beams=True)
    assert torch.equal(a, b)
