"""Drop-in `models.layers` for MI355X: same free functions, class names, constructor
arguments, sub-module names (hence state-dict keys) and default initialisation order as
the reference's models/layers.py, but every forward runs hand-written HIP kernels
through the C ABI (qarig.functional / qarig.ops).  Parameters stay ordinary
nn.Parameters owned by torch; the nn.Linear / nn.Conv2d children exist to hold them
(and to give identical init and key names) -- their own forward() is never called.

CUDA(HIP) tensors only: a CPU tensor raises (no fallback; the CPU restatement is the
test oracle under oracle/).
"""
import torch.nn as nn

from qarig import functional as QF
from qarig import ops
from qarig.ops import act_id


# reference models/layers.py:8-34
def patchify(image, patch_dim=(4, 4)):
    return ops.patchify(image, patch_dim)


# reference models/layers.py:37-71
def unpatchify(patches, image_dim=(32, 32), patch_dim=(4, 4)):
    return ops.unpatchify(patches, image_dim, patch_dim)


_ACTIVATIONS = {"silu": nn.SiLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


# reference models/layers.py:74-80 (KeyError on an unknown name, as the ModuleDict lookup)
def get_activation(activation_type):
    return _ACTIVATIONS[activation_type]()


# reference models/layers.py:83-96
def get_positional_embeddings(emb_dim, pos_index):
    return ops.posemb(pos_index, emb_dim)


def _lin_params(layer):
    """(weight, bias) of the nn.Linear inside a LinearLayer."""
    lin = layer.linear_layer[0]
    return lin.weight, lin.bias


class ScaleLayer(nn.Module):
    """reference models/layers.py:100-111 (weight zero-initialised, bias default)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.scale = nn.Linear(in_dim, out_dim)
        nn.init.zeros_(self.scale.weight)

    def forward(self, x):
        return QF.linear_act(x, self.scale.weight, self.scale.bias)


class ShiftLayer(nn.Module):
    """reference models/layers.py:115-126."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.shift = nn.Linear(in_dim, out_dim)
        nn.init.zeros_(self.shift.weight)

    def forward(self, x):
        return QF.linear_act(x, self.shift.weight, self.shift.bias)


class AdaLNZero(nn.Module):
    """scale(cond) * LayerNorm(x) + shift(cond); reference models/layers.py:130-153.
    The norm, the modulation multiply and the add are one kernel."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.norm = nn.LayerNorm(out_dim, elementwise_affine=False)
        self.scale_layer = ScaleLayer(in_dim=in_dim, out_dim=out_dim)
        self.shift_layer = ShiftLayer(in_dim=in_dim, out_dim=out_dim)

    def forward(self, x, cond, with_skip=False):
        if isinstance(cond, QF.CondTable):     # projections once per position, indexed per token
            return QF.layernorm_mod_table(x, cond.projection(self.scale_layer.scale),
                                          cond.projection(self.shift_layer.shift), cond, self.norm.eps,
                                          with_skip=with_skip)
        if isinstance(cond, QF.CondTokens):    # per-token cond, projections evaluated in groups
            return QF.layernorm_mod(x, cond.projection(self.scale_layer.scale),
                                    cond.projection(self.shift_layer.shift), self.norm.eps, with_skip=with_skip)
        return QF.layernorm_mod(x, self.scale_layer(cond), self.shift_layer(cond), self.norm.eps,
                                with_skip=with_skip)


def _norm_forward(norm, x, cond, use_adaln0):
    """(normalised x, alias of x for the block's skip connection): both gradients of x then meet in
    the normalisation node's backward, whose kernel adds them (no separate accumulation launch)."""
    if use_adaln0:
        return norm(x, cond=cond, with_skip=True)
    return QF.layernorm_affine(x, norm.weight, norm.bias, norm.eps, with_skip=True)


class ConvLayer(nn.Module):
    """Conv2d(k, stride, padding) [+ activation]; reference models/layers.py:157-184."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1,
                 use_activation=True, activation_type="silu"):
        super().__init__()
        mods = [nn.Conv2d(in_channels=in_channels, out_channels=out_channels,
                          kernel_size=kernel_size, stride=stride, padding=padding)]
        self._act = act_id(activation_type) if use_activation else 0
        if use_activation:
            mods.append(get_activation(activation_type))
        self.conv_layer = nn.Sequential(*mods)

    def forward(self, x):
        conv = self.conv_layer[0]
        return QF.conv2d_act(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], self._act)


class UpsampleConvLayer(nn.Module):
    """ConvTranspose2d(4, stride 2, padding 1) + activation; reference layers.py:188-207."""

    def __init__(self, in_channels, out_channels, activation_type="silu"):
        super().__init__()
        self._act = act_id(activation_type)
        self.conv_layer = nn.Sequential(
            nn.ConvTranspose2d(in_channels=in_channels, out_channels=out_channels, kernel_size=4,
                               stride=2, padding=1),
            get_activation(activation_type))

    def forward(self, x):
        conv = self.conv_layer[0]
        return QF.conv_transpose2d_act(x, conv.weight, conv.bias, self._act)


class DownsampleConvLayer(nn.Module):
    """Conv2d(3, stride 2, padding 1) + activation; reference layers.py:211-230."""

    def __init__(self, in_channels, out_channels, activation_type="silu"):
        super().__init__()
        self._act = act_id(activation_type)
        self.conv_layer = nn.Sequential(
            nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=2,
                      padding=1),
            get_activation(activation_type))

    def forward(self, x):
        conv = self.conv_layer[0]
        return QF.conv2d_act(x, conv.weight, conv.bias, 2, 1, self._act)


class LinearLayer(nn.Module):
    """nn.Linear [+ activation]; reference models/layers.py:234-254.  Bias and
    activation are the GEMM's epilogue."""

    def __init__(self, in_dim, out_dim, use_activation=True, activation_type="silu"):
        super().__init__()
        mods = [nn.Linear(in_dim, out_dim)]
        self._act = act_id(activation_type) if use_activation else 0
        if use_activation:
            mods.append(get_activation(activation_type=activation_type))
        self.linear_layer = nn.Sequential(*mods)

    def forward(self, x):
        w, b = _lin_params(self)
        return QF.linear_act(x, w, b, act=self._act)


def _mlp2_forward(seq, x):
    """Two stacked LinearLayers (q/k/v blocks, FFN, pos-cond MLP, classifier) as one
    fused autograd node."""
    w1, b1 = _lin_params(seq[0])
    w2, b2 = _lin_params(seq[1])
    return QF.mlp2(x, w1, b1, w2, b2, seq[0]._act, seq[1]._act)


class ResidualLinearLayer(nn.Module):
    """act(Linear(x * scale(cond)) + skip(x_skip)); reference models/layers.py:258-304.
    Skip-add and activation ride in the GEMM epilogue."""

    def __init__(self, in_dim=512, out_dim=512, skip_dim=512, cond_dim=512, use_scale_layer=False,
                 activation_type="silu"):
        super().__init__()
        self.use_scale_layer = use_scale_layer
        if self.use_scale_layer:
            self.scale_layer = ScaleLayer(in_dim=cond_dim, out_dim=in_dim)
        self.linear = LinearLayer(in_dim=in_dim, out_dim=out_dim, use_activation=False)
        if skip_dim != out_dim:
            self.skip_linear = LinearLayer(in_dim=skip_dim, out_dim=out_dim, use_activation=False)
        else:
            self.skip_linear = nn.Identity()
        self.activation = get_activation(activation_type=activation_type)
        self._act = act_id(activation_type)

    def forward(self, x, x_skip, cond=None):
        if self.use_scale_layer:
            if isinstance(cond, QF.CondTable):
                x = QF.mul_table(x, cond.projection(self.scale_layer.scale), cond)
            elif isinstance(cond, QF.CondTokens):
                x = QF.mul(x, cond.projection(self.scale_layer.scale))
            else:
                x = QF.mul(x, self.scale_layer(cond))
        x_skip = self.skip_linear(x_skip)
        w, b = _lin_params(self.linear)
        return QF.linear_act(x, w, b, residual=x_skip, act=self._act)


class FeedforwardBlock(nn.Module):
    """norm -> Linear+act -> Linear+act -> residual; reference layers.py:308-366."""

    def __init__(self, in_dim=512, hidden_dim=512, cond_dim=512, use_adaln0=False,
                 use_scale_layer=False, activation_type="silu"):
        super().__init__()
        self.use_adaln0 = use_adaln0
        if self.use_adaln0:
            self.feedforward_norm = AdaLNZero(in_dim=cond_dim, out_dim=in_dim)
        else:
            self.feedforward_norm = nn.LayerNorm(in_dim)
        self.feedforward = nn.Sequential(
            LinearLayer(in_dim=in_dim, out_dim=hidden_dim, use_activation=True,
                        activation_type=activation_type),
            LinearLayer(in_dim=hidden_dim, out_dim=in_dim, use_activation=True,
                        activation_type=activation_type))
        self.feedforward_res = ResidualLinearLayer(
            in_dim=in_dim, out_dim=in_dim, skip_dim=in_dim, cond_dim=cond_dim,
            use_scale_layer=use_scale_layer, activation_type=activation_type)

    def forward(self, x, cond=None):
        h, xs = _norm_forward(self.feedforward_norm, x, cond, self.use_adaln0)
        h = _mlp2_forward(self.feedforward, h)
        return self.feedforward_res(x=h, x_skip=xs, cond=cond)


class AttentionLayer(nn.Module):
    """q/k/v two-layer MLPs + multi-head attention; reference layers.py:370-474.
    Heads are addressed in place in the (N,S,H*d) layout; scores never reach HBM."""

    def __init__(self, heads=8, in_dim=512, cross_cond_dim=512, hidden_dim=2_048,
                 use_cross_attn=True, use_masked_attn=True, activation_type="silu"):
        super().__init__()
        if in_dim % heads or ops.attention_head_dim(in_dim // heads) is None:
            # the reference accepts any divisor of in_dim; head dims that are not instantiated run
            # zero-padded on the next one (QF.attention), only heads wider than 128 have no kernel
            raise ValueError(f"AttentionLayer: in_dim {in_dim} / heads {heads} gives head dim "
                             f"{in_dim / heads:g}; the MI355X kernels serve head dims up to "
                             f"{ops.ATTENTION_HEAD_DIMS[-1]} that divide in_dim")
        self.heads = heads
        self.head_dim = in_dim // heads
        self.use_cross_attn = use_cross_attn
        self.use_masked_attn = use_masked_attn
        if not self.use_cross_attn:
            cross_cond_dim = in_dim

        def block(src_dim):
            return nn.Sequential(
                LinearLayer(in_dim=src_dim, out_dim=hidden_dim, use_activation=True,
                            activation_type=activation_type),
                LinearLayer(in_dim=hidden_dim, out_dim=in_dim, use_activation=False))

        self.q_block = block(in_dim)
        self.k_block = block(cross_cond_dim)
        self.v_block = block(cross_cond_dim)

    def kv_params(self):
        """((w1, b1, w2, b2) of k_block, the same of v_block, act1, act2) when the two blocks can
        run as grouped launches on their common input (same activations), else None."""
        kb, vb = self.k_block, self.v_block
        if (kb[0]._act, kb[1]._act) != (vb[0]._act, vb[1]._act):
            return None
        return ((*_lin_params(kb[0]), *_lin_params(kb[1])), (*_lin_params(vb[0]), *_lin_params(vb[1])),
                kb[0]._act, kb[1]._act)

    def forward(self, x, cross_cond=None, kv=None):
        """kv (additive): (k, v) already evaluated by the caller -- the cross-attention k / v MLPs of
        every decoder layer read the same encoder output, so models.Transformer evaluates them for
        all layers at once."""
        if not self.use_cross_attn:      # q, k, v from the same rows: one autograd node
            blocks = (self.q_block, self.k_block, self.v_block)
            if len({(b[0]._act, b[1]._act) for b in blocks}) == 1:
                params = [(*_lin_params(b[0]), *_lin_params(b[1])) for b in blocks]
                q, k, v = QF.mlp2x3(x, params, blocks[0][0]._act, blocks[0][1]._act)
                return QF.attention(q, k, v, self.heads, self.use_masked_attn)
        src = cross_cond if self.use_cross_attn else x
        q = _mlp2_forward(self.q_block, x)
        if kv is not None:
            k, v = kv
        else:
            kvp = self.kv_params()
            if kvp is not None:          # k and v from the same rows: grouped launches
                k, v = QF.mlp2xg(src, kvp[:2], kvp[2], kvp[3])
            else:
                k = _mlp2_forward(self.k_block, src)
                v = _mlp2_forward(self.v_block, src)
        return QF.attention(q, k, v, self.heads, self.use_masked_attn)


class SelfAttentionBlock(nn.Module):
    """norm -> self-attention -> residual; reference models/layers.py:478-534."""

    def __init__(self, heads=8, in_dim=512, cond_dim=512, hidden_dim=512, use_adaln0=False,
                 use_scale_layer=False, use_masked_attn=True, activation_type="silu"):
        super().__init__()
        self.use_adaln0 = use_adaln0
        if self.use_adaln0:
            self.self_attn_norm = AdaLNZero(in_dim=cond_dim, out_dim=in_dim)
        else:
            self.self_attn_norm = nn.LayerNorm(in_dim)
        self.self_attn = AttentionLayer(
            heads=heads, in_dim=in_dim, hidden_dim=hidden_dim, use_cross_attn=False,
            use_masked_attn=use_masked_attn, activation_type=activation_type)
        self.self_attn_res = ResidualLinearLayer(
            in_dim=in_dim, out_dim=in_dim, skip_dim=in_dim, cond_dim=cond_dim,
            use_scale_layer=use_scale_layer, activation_type=activation_type)

    def forward(self, x, cond=None):
        h, xs = _norm_forward(self.self_attn_norm, x, cond, self.use_adaln0)
        h = self.self_attn(h)
        return self.self_attn_res(x=h, x_skip=xs, cond=cond)


class CrossAttentionBlock(nn.Module):
    """norm -> cross-attention -> residual; reference models/layers.py:538-599."""

    def __init__(self, heads=8, in_dim=512, cond_dim=512, cross_cond_dim=512, hidden_dim=512,
                 use_adaln0=False, use_scale_layer=False, activation_type="silu"):
        super().__init__()
        self.use_adaln0 = use_adaln0
        if self.use_adaln0:
            self.cross_attn_norm = AdaLNZero(in_dim=cond_dim, out_dim=in_dim)
        else:
            self.cross_attn_norm = nn.LayerNorm(in_dim)
        self.cross_attn = AttentionLayer(
            heads=heads, in_dim=in_dim, cross_cond_dim=cross_cond_dim, hidden_dim=hidden_dim,
            use_cross_attn=True, use_masked_attn=False, activation_type=activation_type)
        self.cross_attn_res = ResidualLinearLayer(
            in_dim=in_dim, out_dim=in_dim, skip_dim=in_dim, cond_dim=cond_dim,
            use_scale_layer=use_scale_layer, activation_type=activation_type)

    def forward(self, x, cross_cond, cond=None, kv=None):
        h, xs = _norm_forward(self.cross_attn_norm, x, cond, self.use_adaln0)
        h = self.cross_attn(x=h, cross_cond=cross_cond, kv=kv)
        return self.cross_attn_res(x=h, cond=cond, x_skip=xs)


class TransformerBlock(nn.Module):
    """self-attn block [-> cross-attn block] -> FFN block; reference layers.py:603-667."""

    def __init__(self, in_dim=512, cond_dim=512, cross_cond_dim=512, hidden_dim=512,
                 self_attn_heads=8, cross_attn_heads=8, use_cross_attn=True, use_masked_attn=True,
                 use_adaln0=False, use_scale_layer=False, activation_type="silu"):
        super().__init__()
        self.use_cross_attn = use_cross_attn
        self.self_attn_block = SelfAttentionBlock(
            heads=self_attn_heads, in_dim=in_dim, cond_dim=cond_dim, hidden_dim=hidden_dim,
            use_adaln0=use_adaln0, use_scale_layer=use_scale_layer,
            use_masked_attn=use_masked_attn, activation_type=activation_type)
        if self.use_cross_attn:
            self.cross_attn_block = CrossAttentionBlock(
                heads=cross_attn_heads, in_dim=in_dim, cond_dim=cond_dim,
                cross_cond_dim=cross_cond_dim, hidden_dim=hidden_dim, use_adaln0=use_adaln0,
                use_scale_layer=use_scale_layer, activation_type=activation_type)
        self.feedforward_block = FeedforwardBlock(
            in_dim=in_dim, hidden_dim=hidden_dim, use_adaln0=use_adaln0, cond_dim=cond_dim,
            use_scale_layer=use_scale_layer, activation_type=activation_type)

    def forward(self, x, cross_cond=None, pos_cond=None, cross_kv=None):
        x = self.self_attn_block(x, cond=pos_cond)
        if self.use_cross_attn:
            x = self.cross_attn_block(x, cond=pos_cond, cross_cond=cross_cond, kv=cross_kv)
        return self.feedforward_block(x, cond=pos_cond)
