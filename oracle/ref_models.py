"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Functional, CPU, fp32 (or fp64 when fed fp64 tensors) restatement of the reference's
hot path, written from the behaviour documented in SURVEY.md 8a.  It takes weights as
a flat {reference state-dict key: tensor} mapping so that a reference checkpoint, a
golden fixture or the product modules' state_dict() can all drive it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.

Pinned to the reference by tests/golden/*.npz, produced by oracle/make_goldens.py
from the reference's own classes (see tests/test_oracle_golden.py).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------- layers.py:8-96
def patchify(image, patch_dim):
    """(N,C,H,W) -> (N,Seq,C*pH*pW); patch grid row-major, inside a patch
    channel-major then row then column.  reference models/layers.py:8-34"""
    pH, pW = patch_dim
    N, C, H, W = image.shape
    gh, gw = H // pH, W // pW
    x = image[:, :, :gh * pH, :gw * pW] if (H % pH or W % pW) else image
    x = x.reshape(N, C, gh, pH, gw, pW).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(N, gh * gw, C * pH * pW)


def unpatchify(patches, image_dim, patch_dim):
    """Inverse of patchify.  reference models/layers.py:37-71"""
    H, W = image_dim
    pH, pW = patch_dim
    N, _, D = patches.shape
    gh, gw = H // pH, W // pW
    C = D // (pH * pW)
    x = patches.reshape(N, gh, gw, C, pH, pW).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(N, C, gh * pH, gw * pW)


def activation(x, kind):
    """reference models/layers.py:74-80"""
    if kind is None:
        return x
    if kind == "silu":
        return x * torch.sigmoid(x)
    if kind == "tanh":
        return torch.tanh(x)
    if kind == "sigmoid":
        return torch.sigmoid(x)
    raise KeyError(kind)


def positional_frequencies(emb_dim, device=None):
    """f_i = exp(-i * ln(10000)/(half-1)), fp32.  reference models/layers.py:84-91"""
    half = emb_dim // 2
    c = math.log(10_000) / (half - 1)
    return torch.exp(torch.arange(half, dtype=torch.float32, device=device) * -c)


def positional_embeddings(emb_dim, pos_index):
    """cat(sin(pos*f), cos(pos*f)) -> (len(pos), emb_dim).  reference layers.py:83-96"""
    ang = pos_index[:, None] * positional_frequencies(emb_dim, pos_index.device)[None, :]
    return torch.cat((ang.sin(), ang.cos()), dim=1)


# --------------------------------------------------------------- small param ops
def _lin(sd, prefix, x, act=None):
    """LinearLayer: nn.Linear then optional activation.  reference layers.py:234-254"""
    y = F.linear(x, sd[prefix + ".linear_layer.0.weight"], sd[prefix + ".linear_layer.0.bias"])
    return activation(y, act)


def _mlp2(sd, prefix, x, hidden_act, out_act=None):
    return _lin(sd, prefix + ".1", _lin(sd, prefix + ".0", x, hidden_act), out_act)


def _norm(sd, prefix, x, cond, use_adaln0):
    """AdaLNZero (layers.py:130-153) or affine LayerNorm."""
    d = x.shape[-1]
    if use_adaln0:
        xn = F.layer_norm(x, (d,))
        scale = F.linear(cond, sd[prefix + ".scale_layer.scale.weight"],
                         sd[prefix + ".scale_layer.scale.bias"])
        shift = F.linear(cond, sd[prefix + ".shift_layer.shift.weight"],
                         sd[prefix + ".shift_layer.shift.bias"])
        return scale * xn + shift
    return F.layer_norm(x, (d,), sd[prefix + ".weight"], sd[prefix + ".bias"])


def _residual(sd, prefix, x, x_skip, cond, use_scale, act):
    """ResidualLinearLayer: act(Linear(x * scale(cond)) + skip).  layers.py:258-304"""
    if use_scale:
        x = x * F.linear(cond, sd[prefix + ".scale_layer.scale.weight"],
                         sd[prefix + ".scale_layer.scale.bias"])
    y = _lin(sd, prefix + ".linear", x)
    if (prefix + ".skip_linear.linear_layer.0.weight") in sd:
        x_skip = _lin(sd, prefix + ".skip_linear", x_skip)
    return activation(y + x_skip, act)


def attention(sd, prefix, x, heads, act, cross_cond=None, masked=False):
    """AttentionLayer.  reference models/layers.py:370-474"""
    q = _mlp2(sd, prefix + ".q_block", x, act)
    src = x if cross_cond is None else cross_cond
    k = _mlp2(sd, prefix + ".k_block", src, act)
    v = _mlp2(sd, prefix + ".v_block", src, act)
    N, Sq, D = q.shape
    Sk = k.shape[1]
    d = D // heads
    qh = q.reshape(N, Sq, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(N, Sk, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(N, Sk, heads, d).permute(0, 2, 1, 3)
    s = torch.matmul(qh, kh.transpose(-1, -2)) / (d ** 0.5)
    if masked:
        S = x.shape[1]
        m = torch.triu(torch.ones((1, 1, S, S), device=q.device, dtype=s.dtype), diagonal=1)
        s = s * (1 - m) + 2e9 * m
        s = s.masked_fill(s >= 2e9, float("-inf"))
    p = torch.softmax(s, dim=3)
    o = torch.matmul(p, vh)
    return o.permute(0, 2, 1, 3).reshape(N, Sq, D)


def transformer_block(sd, prefix, x, heads_self, heads_cross, act, cross_cond, pos_cond,
                      use_cross, masked, adaln0, use_scale):
    """TransformerBlock = self-attn block [-> cross-attn block] -> FFN block.
    reference models/layers.py:478-667"""
    p = prefix + ".self_attn_block"
    h = _norm(sd, p + ".self_attn_norm", x, pos_cond, adaln0)
    h = attention(sd, p + ".self_attn", h, heads_self, act, None, masked)
    x = _residual(sd, p + ".self_attn_res", h, x, pos_cond, use_scale, act)
    if use_cross:
        p = prefix + ".cross_attn_block"
        h = _norm(sd, p + ".cross_attn_norm", x, pos_cond, adaln0)
        h = attention(sd, p + ".cross_attn", h, heads_cross, act, cross_cond, False)
        x = _residual(sd, p + ".cross_attn_res", h, x, pos_cond, use_scale, act)
    p = prefix + ".feedforward_block"
    h = _norm(sd, p + ".feedforward_norm", x, pos_cond, adaln0)
    h = _mlp2(sd, p + ".feedforward", h, act, act)
    return _residual(sd, p + ".feedforward_res", h, x, pos_cond, use_scale, act)


def transformer_forward(sd, cfg, x_dec, x_enc=None, pos_cond=None):
    """Transformer.forward.  reference models/Transformer.py:122-202
    cfg keys: use_encoder, use_pos_cond, num_enc_layers, num_dec_layers,
    self_attn_heads, cross_attn_heads, hidden_activation."""
    act = cfg.get("hidden_activation", "silu")
    dt = sd["dec_embedding.weight"].dtype
    enc = None
    if cfg["use_encoder"]:
        enc = F.embedding(x_enc, sd["enc_embedding.weight"])
        S, D = enc.shape[1], enc.shape[2]
        pos = torch.arange(1, S + 1, device=enc.device)
        enc = enc + positional_embeddings(D, pos).to(dt).unsqueeze(0)
        for i in range(cfg["num_enc_layers"]):
            enc = transformer_block(sd, f"encoder_layers.{i}", enc, cfg["self_attn_heads"], None,
                                    act, None, None, False, False, False, False)
    x = F.embedding(x_dec, sd["dec_embedding.weight"])
    N, S, D = x.shape
    pos = torch.arange(1, S + 1, device=x.device)
    x = x + positional_embeddings(D, pos).to(dt).unsqueeze(0)
    cond = None
    if cfg["use_pos_cond"]:
        cond = positional_embeddings(D, pos_cond.flatten()).to(dt).reshape(N, S, D)
        cond = _mlp2(sd, "pos_cond_layer", cond, act)
    for i in range(cfg["num_dec_layers"]):
        x = transformer_block(sd, f"decoder_layers.{i}", x, cfg["self_attn_heads"],
                              cfg.get("cross_attn_heads"), act, enc, cond, cfg["use_encoder"],
                              True, cfg["use_pos_cond"], cfg["use_pos_cond"])
    # classifier: LinearLayer(use_activation=True) default activation is silu
    return _mlp2(sd, "classifier", x, "silu")


# -------------------------------------------------------------- conv autoencoder
def _conv(sd, prefix, x, stride, act):
    """ConvLayer / DownsampleConvLayer: Conv2d 3x3 pad 1.  layers.py:157-184,211-230"""
    y = F.conv2d(x, sd[prefix + ".conv_layer.0.weight"], sd[prefix + ".conv_layer.0.bias"],
                 stride=stride, padding=1)
    return activation(y, act)


def _upconv(sd, prefix, x, act):
    """UpsampleConvLayer: ConvTranspose2d 4x4 s2 p1.  layers.py:188-207"""
    y = F.conv_transpose2d(x, sd[prefix + ".conv_layer.0.weight"],
                           sd[prefix + ".conv_layer.0.bias"], stride=2, padding=1)
    return activation(y, act)


def fc_encoder(sd, x, num_layers=2, hidden_act="silu", use_final_activation=True,
               final_act="tanh", prefix="fc_encoder_layer"):
    """FC_Encoder.forward.  reference models/FC_Encoder.py:12-89"""
    i = 0
    x = _conv(sd, f"{prefix}.{i}", x, 1, hidden_act); i += 1
    for _ in range(num_layers):
        x = _conv(sd, f"{prefix}.{i}", x, 1, hidden_act); i += 1
        x = _conv(sd, f"{prefix}.{i}", x, 2, hidden_act); i += 1
    return _conv(sd, f"{prefix}.{i}", x, 1, final_act if use_final_activation else None)


def fc_decoder(sd, x, num_layers=2, hidden_act="silu", use_final_activation=True,
               final_act="tanh", prefix="fc_decoder_layer"):
    """FC_Decoder.forward.  reference models/FC_Decoder.py:12-96"""
    x = _conv(sd, f"{prefix}.0.0", x, 1, hidden_act)
    x = _conv(sd, f"{prefix}.0.1", x, 1, hidden_act)
    i = 1
    for _ in range(num_layers):
        x = _conv(sd, f"{prefix}.{i}", x, 1, hidden_act); i += 1
        x = _upconv(sd, f"{prefix}.{i}", x, hidden_act); i += 1
    return _conv(sd, f"{prefix}.{i}", x, 1, final_act if use_final_activation else None)


def autoencoder(sd, x, num_layers=2, hidden_act="silu", use_final_enc_activation=True,
                enc_act="silu", use_final_dec_activation=True, dec_act="tanh"):
    """Autoencoder.forward.  reference models/Autoencoder.py:63-74"""
    z = fc_encoder(sd, x, num_layers, hidden_act, use_final_enc_activation, enc_act,
                   "fc_encoder.fc_encoder_layer")
    return fc_decoder(sd, z, num_layers, hidden_act, use_final_dec_activation, dec_act,
                      "fc_decoder.fc_decoder_layer")


# ---------------------------------------------------------------------- codebook
def codebook_distances(x_flat, weight):
    """torch.cdist default compute mode restated: matmul form when either side has
    more than 25 rows, direct form otherwise (SURVEY.md 7 hard part 1)."""
    if x_flat.shape[0] > 25 or weight.shape[0] > 25:
        x2 = (x_flat * x_flat).sum(1, keepdim=True)
        w2 = (weight * weight).sum(1)[None, :]
        d2 = x2 + w2 - 2.0 * (x_flat @ weight.t())
        return d2.clamp_min(0).sqrt()
    diff = x_flat[:, None, :] - weight[None, :, :]
    return (diff * diff).sum(-1).sqrt()


def codebook_bmu(weight, x, patch_dim, reshape=False):
    """Codebook.get_patches_bmu.  reference models/Codebook.py:77-99"""
    xp = patchify(x, patch_dim)
    N, Seq, D = xp.shape
    idx = torch.argmin(codebook_distances(xp.reshape(N * Seq, D), weight), dim=-1)
    return idx.reshape(N, Seq) if reshape else idx


def neighbourhood_variance(neighbourhood_range):
    """sigma^2 so that the Gaussian is 0.1 at the range.  Codebook.py:118"""
    return -(neighbourhood_range / (2 * math.log(0.1)))


def codebook_quantized_patches(weight, x, patch_dim, neighbourhood_range, use_gaussian=True,
                               bmu=None):
    """Codebook.get_quantized_patches.  reference models/Codebook.py:102-135"""
    if bmu is None:
        bmu = codebook_bmu(weight, x, patch_dim)
    N = x.shape[0]
    K, D = weight.shape
    if use_gaussian:
        j = torch.arange(K, device=x.device)[None, :]
        g = torch.exp(-(((j - bmu[:, None]) ** 2) / (2 * neighbourhood_variance(neighbourhood_range))))
        q = g.to(weight.dtype) @ weight
    else:
        q = weight[bmu]
    return q.view(N, -1, D)


def codebook_quantized_image(weight, indices, image_dim, patch_dim, unpatchify_input=True):
    """Codebook.get_quantized_image.  reference models/Codebook.py:138-154"""
    N, Seq = indices.shape
    q = weight[indices.flatten()].view(N, Seq, weight.shape[1])
    return unpatchify(q, image_dim, patch_dim) if unpatchify_input else q


def codebook_forward(weight, x, image_dim, patch_dim, neighbourhood_range, use_gaussian=True,
                     bmu=None):
    """Codebook.forward.  reference models/Codebook.py:156-164"""
    q = codebook_quantized_patches(weight, x, patch_dim, neighbourhood_range, use_gaussian, bmu)
    return unpatchify(q, image_dim, patch_dim)


def decrease_neighbourhood(neighbourhood_range):
    """Codebook.decrease_neighbourhood (ignores `steps`).  Codebook.py:68-74"""
    return 1.0 if neighbourhood_range <= 1 else neighbourhood_range - 1


# ------------------------------------------------------------ loss / optimiser
def cross_entropy(logits, target):
    """nn.CrossEntropyLoss() mean over N*S.  train_quantized_transformer.py:337,496-502"""
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), target.flatten())


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr, beta1=0.5, beta2=0.999, eps=1e-8):
    """torch.optim.Adam(betas=(0.5,0.999)) single step, no weight decay, no amsgrad
    (train_quantized_transformer.py:317-320).  In-place on the given tensors."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))
