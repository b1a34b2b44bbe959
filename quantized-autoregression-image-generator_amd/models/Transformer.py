"""Drop-in `models.Transformer.Transformer` (reference models/Transformer.py:16-202):
optional encoder stack + DiT-style decoder stack + position-conditioning MLP +
classifier, same constructor, state-dict keys and forward signature; compute on HIP."""
import torch
import torch.nn as nn
from torch.utils import checkpoint

from qarig import functional as QF
from qarig import ops

from ._loading import load_matching
from .layers import LinearLayer, TransformerBlock, _mlp2_forward


class Transformer(nn.Module):
    def __init__(self, use_encoder=True, use_pos_cond=True, num_enc_layers=5, num_dec_layers=10,
                 num_enc_embedding=512, num_dec_embedding=512, self_attn_heads=8,
                 cross_attn_heads=8, transformer_in_dim=512, transformer_out_dim=512,
                 transformer_hidden_dim=4096, hidden_activation="silu",
                 use_activation_checkpoint=False):
        super().__init__()
        self.use_encoder = use_encoder
        self.use_pos_cond = use_pos_cond
        self.use_activation_checkpoint = use_activation_checkpoint

        if self.use_encoder:
            self.enc_embedding = nn.Embedding(num_embeddings=num_enc_embedding,
                                              embedding_dim=transformer_in_dim)
            self.encoder_layers = nn.ModuleList(
                TransformerBlock(in_dim=transformer_in_dim, hidden_dim=transformer_hidden_dim,
                                 self_attn_heads=self_attn_heads, use_cross_attn=False,
                                 use_masked_attn=False, use_adaln0=False, use_scale_layer=False,
                                 activation_type=hidden_activation)
                for _ in range(num_enc_layers))

        self.dec_embedding = nn.Embedding(num_embeddings=num_dec_embedding,
                                          embedding_dim=transformer_in_dim)
        self.decoder_layers = nn.ModuleList(
            TransformerBlock(in_dim=transformer_in_dim, cond_dim=transformer_in_dim,
                             cross_cond_dim=transformer_in_dim, hidden_dim=transformer_hidden_dim,
                             self_attn_heads=self_attn_heads, cross_attn_heads=cross_attn_heads,
                             use_cross_attn=self.use_encoder, use_masked_attn=True,
                             use_adaln0=self.use_pos_cond, use_scale_layer=self.use_pos_cond,
                             activation_type=hidden_activation)
            for _ in range(num_dec_layers))

        if self.use_pos_cond:
            self.pos_cond_layer = nn.Sequential(
                LinearLayer(in_dim=transformer_in_dim, out_dim=transformer_hidden_dim,
                            use_activation=True, activation_type=hidden_activation),
                LinearLayer(in_dim=transformer_hidden_dim, out_dim=transformer_in_dim,
                            use_activation=False))

        self.classifier = nn.Sequential(
            LinearLayer(in_dim=transformer_in_dim, out_dim=transformer_hidden_dim,
                        use_activation=True),
            LinearLayer(in_dim=transformer_hidden_dim, out_dim=transformer_out_dim,
                        use_activation=False))
        self._pe_cache = {}

    def custom_load_state_dict(self, state_dict):
        load_matching(self, state_dict)

    def _sequence_pe(self, seq, dim, device):
        """sinusoid of positions 1..seq (reference Transformer.py:130-139,159-167);
        constant per (seq, dim), so built once and kept on the device."""
        key = (seq, dim, str(device))
        pe = self._pe_cache.get(key)
        if pe is None:
            pe = ops.posemb(torch.arange(1, seq + 1, device=device), dim)
            self._pe_cache[key] = pe
        return pe

    def encode(self, x_enc):
        """Encoder half alone (constant across the decode steps of one stage, so
        generation can call it once)."""
        table = self.enc_embedding.weight
        enc = QF.embedding_pos(x_enc, table, self._sequence_pe(x_enc.shape[1], table.shape[1],
                                                               table.device))
        for layer in self.encoder_layers:
            if self.use_activation_checkpoint and torch.is_grad_enabled():
                enc = checkpoint.checkpoint(layer, enc, use_reentrant=False)
            else:
                enc = layer(enc)
        return enc

    def _cond(self, pos_cond, N, S, D, pos_bound):
        """Conditioning of the decoder blocks.  Integer positions (training): a position table
        -- the sinusoid, pos_cond_layer and (inside the blocks) every scale/shift projection
        evaluated once per position 0..P-1 instead of once per token -- whenever that is at
        least QF.COND_TABLE_MIN_RATIO (4) times fewer rows; P = pos_bound when the caller knows it (no host sync), else
        max(pos)+1.  Float positions (sampling) keep the per-token form."""
        if QF.USE_COND_TABLE and not pos_cond.dtype.is_floating_point:
            P = int(pos_bound) if pos_bound is not None else int(pos_cond.max().item()) + 1
            # whole 128-row tiles: every GEMM on the table (forward, d-input, d-weight with
            # K = P) then takes the interior kernels; rows past the bound are never indexed
            P = (P + 127) // 128 * 128
            # (training pays the table's backward too: worth it from a 4-fold row reduction; inference from 2-fold)
            ratio = QF.COND_TABLE_MIN_RATIO if torch.is_grad_enabled() else min(2, QF.COND_TABLE_MIN_RATIO)
            if P > 0 and ratio * P <= N * S:
                dev = pos_cond.device
                tab = ops.posemb(torch.arange(P, device=dev), D)
                tab = _mlp2_forward(self.pos_cond_layer, tab)
                self._last_cond_form = "table"
                return QF.CondTable(tab, pos_cond.reshape(-1).to(torch.int32).contiguous(), (N, S),
                                    groups=self._cond_linear_groups())
        self._last_cond_form = "per_token"
        cond = ops.posemb(pos_cond.flatten(), D).reshape(N, S, D)
        cond = _mlp2_forward(self.pos_cond_layer, cond)
        if torch.is_grad_enabled() and QF.USE_COND_GROUPS and ops.gemm_grouped_supported(N * S, D, D):
            # training rows without a (worthwhile) position table: the projections of `cond` in groups
            return QF.CondTokens(cond, groups=self._cond_linear_groups())
        return cond

    def _cond_linears_per_layer(self):
        """Per decoder layer, the nn.Linear modules that project `cond` inside its blocks."""
        per_layer = []
        for layer in self.decoder_layers:
            out = []
            blocks = [(layer.self_attn_block, "self_attn_norm", "self_attn_res")]
            if layer.use_cross_attn:
                blocks.append((layer.cross_attn_block, "cross_attn_norm", "cross_attn_res"))
            blocks.append((layer.feedforward_block, "feedforward_norm", "feedforward_res"))
            for blk, norm_attr, res_attr in blocks:
                norm, res = getattr(blk, norm_attr), getattr(blk, res_attr)
                if blk.use_adaln0:
                    out += [norm.scale_layer.scale, norm.shift_layer.shift]
                if res.use_scale_layer:
                    out.append(res.scale_layer.scale)
            per_layer.append(out)
        return per_layer

    def _cond_linear_groups(self):
        """The projections of `cond` grouped for the grouped launches of the position table: ONE
        group (all layers) on a single GPU; one group per decoder layer under data parallelism,
        so that each layer's projection gradients are final when that layer's backward is and
        its all-reduce bucket can leave early."""
        per_layer = self._cond_linears_per_layer()
        mode = QF.COND_TABLE_GROUPING
        if mode is None:
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            mode = "layer" if multi else "all"
        if mode == "layer":
            return per_layer
        return [[l for group in per_layer for l in group]]

    def _cross_kv_all_layers(self, enc):
        """(k, v) of the cross-attention of EVERY decoder layer, evaluated up front as grouped
        launches: the 2 x layers k / v MLPs (reference models/layers.py:389-418, 581-599) all read
        the encoder output.  Single process only (None otherwise): under data parallelism each
        layer groups its own k / v pair, so that its weight gradients are final when the layer's
        backward is and its all-reduce bucket can leave early.  QARIG_CROSS_KV_GROUPING=all|layer
        overrides."""
        mode = QF.CROSS_KV_GROUPING
        if mode is None:
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            mode = "layer" if multi else "all"
        if mode != "all":
            return None
        kvp = [layer.cross_attn_block.cross_attn.kv_params() for layer in self.decoder_layers]
        if any(p is None for p in kvp) or len({p[2:] for p in kvp}) != 1:
            return None
        out = []
        per = ops.GEMM_MAX_GROUPS // 2                 # layers per grouped launch
        for i in range(0, len(kvp), per):
            chunk = kvp[i:i + per]
            res = QF.mlp2xg(enc, [blk for p in chunk for blk in p[:2]], chunk[0][2], chunk[0][3])
            out += [(res[2 * j], res[2 * j + 1]) for j in range(len(chunk))]
        return out

    def decode(self, x_dec, enc=None, pos_cond=None, pos_bound=None):
        table = self.dec_embedding.weight
        N, S = x_dec.shape
        D = table.shape[1]
        x = QF.embedding_pos(x_dec, table, self._sequence_pe(S, D, table.device))
        cond = None
        if self.use_pos_cond:
            cond = self._cond(pos_cond, N, S, D, pos_bound)
        ckpt = self.use_activation_checkpoint and torch.is_grad_enabled()
        kvs = None
        if enc is not None and self.use_encoder and not ckpt and torch.is_grad_enabled():
            kvs = self._cross_kv_all_layers(enc)
        per_layer = self._cond_linears_per_layer() if ckpt and isinstance(cond, (QF.CondTable, QF.CondTokens)) \
            else None
        for li, layer in enumerate(self.decoder_layers):
            if ckpt:
                if per_layer is not None:
                    # grouped table projections are cached on `cond`: evaluate this layer's group
                    # OUTSIDE the checkpointed region, so that the original forward and the
                    # recomputation both find them cached and save the same tensors
                    cond.ensure(per_layer[li])
                x = checkpoint.checkpoint(layer, x, cross_cond=enc, pos_cond=cond,
                                          use_reentrant=False)
            else:
                x = layer(x=x, cross_cond=enc, pos_cond=cond,
                          cross_kv=kvs[li] if kvs is not None else None)
        return _mlp2_forward(self.classifier, x)

    def forward(self, x_dec, x_enc=None, pos_cond=None, pos_bound=None):
        """pos_bound (optional, additive): exclusive upper bound of the integer positions in
        pos_cond, so that the position table can be sized without reading pos_cond back."""
        enc = self.encode(x_enc) if self.use_encoder else None
        return self.decode(x_dec, enc, pos_cond, pos_bound)
