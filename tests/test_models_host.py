"""Host-side logic of the drop-in classes that needs no GPU: state-dict key parity with
the reference, identical default initialisation under a seed, loader semantics."""
import contextlib
import io

import torch

from conftest import load_golden


def test_transformer_keys_and_init_match_reference():
    from models.Transformer import Transformer
    g = load_golden("transformer_encdec_pos")
    torch.manual_seed(7)  # the seed oracle/make_goldens.py used before constructing the reference
    m = Transformer(use_encoder=True, use_pos_cond=True, num_enc_layers=2, num_dec_layers=2,
                    num_enc_embedding=24, num_dec_embedding=40, self_attn_heads=4,
                    cross_attn_heads=2, transformer_in_dim=32, transformer_out_dim=33,
                    transformer_hidden_dim=64, hidden_activation="silu")
    sd = m.state_dict()
    assert set(sd) == set(g["sd"])
    same = 0
    for k, v in sd.items():
        assert v.shape == g["sd"][k].shape, k
        if v.abs().max() > 0:           # zero-initialised tensors were perturbed in the golden
            assert torch.equal(v, g["sd"][k]), k
            same += 1
    assert same > 100  # same construction order => same RNG stream => bit-identical init


def test_autoencoder_keys_and_loader_hacks():
    from models.Autoencoder import Autoencoder
    from models.FC_Decoder import FC_Decoder
    from models.FC_Encoder import FC_Encoder
    g = load_golden("autoencoder")
    torch.manual_seed(3)
    ae = Autoencoder(num_layers=2, image_channel=3, min_channel=8, max_channel=16, latent_channel=4,
                     encoder_activation_type="tanh")
    assert set(ae.state_dict()) == set(g["sd"])
    for k, v in ae.state_dict().items():
        assert torch.equal(v, g["sd"][k]), k
    enc = FC_Encoder(num_layers=2, image_channel=3, min_channel=8, max_channel=16, latent_channel=4)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        enc.custom_load_state_dict(g["sd"])
    assert "No Layer found: fc_decoder" in out.getvalue()      # decoder keys reported, not fatal
    assert torch.equal(enc.state_dict()["fc_encoder_layer.0.conv_layer.0.weight"],
                       g["sd"]["fc_encoder.fc_encoder_layer.0.conv_layer.0.weight"])
    dec = FC_Decoder(num_layers=2, image_channel=3, min_channel=8, max_channel=16, latent_channel=4)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        dec.custom_load_state_dict(g["sd"])
    assert "Skipping: fc_encoder" in out.getvalue()            # "decoder" filter
    assert torch.equal(dec.state_dict()["fc_decoder_layer.0.0.conv_layer.0.weight"],
                       g["sd"]["fc_decoder.fc_decoder_layer.0.0.conv_layer.0.weight"])
    # shape mismatch is skipped, never raises
    bad = {"fc_decoder_layer.5.conv_layer.0.bias": torch.zeros(7)}
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        dec.custom_load_state_dict(bad)
    assert "Skipped: fc_decoder_layer.5.conv_layer.0.bias" in out.getvalue()


def test_codebook_init_and_channel_rules():
    from models.Codebook import Codebook
    from models.FC_Decoder import FC_Decoder
    from models.FC_Encoder import FC_Encoder
    cb = Codebook(patch_dim=(4, 4), image_dim=(32, 32), image_channel=4, num_embeddings=512)
    w = cb.codebook.weight
    assert w.shape == (512, 64) and float(w.abs().max()) <= 1 / 512
    assert list(cb.state_dict()) == ["codebook.weight"]
    enc = FC_Encoder(num_layers=2, image_channel=3, min_channel=256, max_channel=512, latent_channel=4)
    shapes = [tuple(m.conv_layer[0].weight.shape[:2]) for m in enc.fc_encoder_layer]
    assert shapes == [(256, 3), (256, 256), (512, 256), (512, 512), (512, 512), (4, 512)]
    dec = FC_Decoder(num_layers=2, image_channel=3, min_channel=256, max_channel=512, latent_channel=4)
    assert tuple(dec.fc_decoder_layer[2].conv_layer[0].weight.shape) == (512, 256, 4, 4)
    assert tuple(dec.fc_decoder_layer[5].conv_layer[0].weight.shape) == (3, 256, 3, 3)
    assert sum(p.numel() for p in enc.parameters()) + sum(p.numel() for p in dec.parameters()) \
        == 14_997_255


def test_readme_transformer_param_counts():
    from models.Transformer import Transformer
    base = Transformer(use_encoder=False, use_pos_cond=True, num_enc_layers=None, num_dec_layers=7,
                       num_enc_embedding=None, num_dec_embedding=1024, self_attn_heads=64,
                       cross_attn_heads=None, transformer_in_dim=512, transformer_out_dim=513,
                       transformer_hidden_dim=2048)
    assert sum(p.numel() for p in base.parameters()) == 78_226_433


def test_unsupported_head_dim_fails_at_construction():
    """The reference accepts any `heads` dividing in_dim; the HIP attention kernels serve every head
    dim up to 128 (4/8/16/32/64/128 directly, the others zero-padded to the next), so a wider one must fail
    when the model is built, not at the first forward."""
    import pytest
    from models.layers import AttentionLayer
    AttentionLayer(heads=64, in_dim=512, hidden_dim=64)
    assert AttentionLayer(heads=4, in_dim=48, hidden_dim=64).head_dim == 12     # runs padded to 16
    assert AttentionLayer(heads=4, in_dim=512, hidden_dim=64).head_dim == 128    # the wide-head kernels
    assert AttentionLayer(heads=5, in_dim=480, hidden_dim=64).head_dim == 96     # padded to 128
    with pytest.raises(ValueError, match="head dim"):
        AttentionLayer(heads=2, in_dim=512, hidden_dim=64)      # d = 256
    with pytest.raises(ValueError, match="head dim"):
        AttentionLayer(heads=7, in_dim=512, hidden_dim=64)      # not a divisor


def test_checkpoint_dicts_load_with_the_weights_only_unpickler(tmp_path):
    """utils.model_utils.load_model defaults to weights_only=True: the checkpoint dict schemas of
    the four CLIs (reference train_quantized_transformer.py:519-534, train_codebook.py:271-278,
    train_autoencoder.py:235-247) must round-trip through it."""
    import torch
    from utils.model_utils import load_model, save_model
    sd = {"w": torch.randn(3, 4), "b": torch.zeros(4)}
    opt = {"state": {0: {"step": torch.tensor(3.0), "exp_avg": torch.zeros(3, 4),
                         "exp_avg_sq": torch.zeros(3, 4)}},
           "param_groups": [{"lr": 1e-4, "betas": (0.5, 0.999), "eps": 1e-8, "weight_decay": 0,
                             "amsgrad": False, "params": [0]}]}
    dicts = {
        "model_1.pt": {"train_base_model": True, "use_sliding_window": True, "sliding_window": 256,
                       "num_enc_embedding": None, "num_dec_embedding": 1024, "num_enc_layers": None,
                       "num_dec_layers": 7, "self_attn_heads": 64, "cross_attn_heads": None,
                       "transformer_in_dim": 512, "transformer_out_dim": 513,
                       "transformer_hidden_dim": 2048, "hidden_activation": "silu", "model": sd,
                       "model_optimizer": opt},
        "codebook_1.pt": {"patch_dim": (4, 4), "image_dim": (32, 32), "image_C": 4, "num_embeddings": 512,
                          "neighbourhood_range": 1.0, "global_steps": 10, "checkpoint": sd},
        "ae_1.pt": {"image_channel": 3, "min_channel": 256, "max_channel": 512, "latent_channel": 4,
                    "num_layers": 2, "hidden_activation_type": "silu", "use_final_enc_activation": True,
                    "encoder_final_activation": "tanh", "use_final_dec_activation": True,
                    "decoder_final_activation": "tanh", "checkpoint": sd, "optimizer": opt},
    }
    for name, d in dicts.items():
        assert save_model(str(tmp_path), name, d)
        ok, got = load_model(str(tmp_path / "models_checkpoint" / name))
        assert ok and set(got) == set(d)
        assert got.get("patch_dim", (4, 4)) == (4, 4)
        inner = got.get("model", got.get("checkpoint"))
        assert torch.equal(inner["w"], sd["w"])
