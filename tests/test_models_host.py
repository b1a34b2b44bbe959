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
