"""Key/value cache for autoregressive decoding.

The reference samples every token by re-running the decoder over the whole window
(generate_images.py:283-307; train_quantized_transformer.py:600-640).  While the window has
not started to slide, the keys and values of tokens already in it never change: the
self-attention is causal, the sequence position of a token is its index in the window and
its `pos_cond` value is fixed when it is appended.  `DecodeCache.step` therefore evaluates
ONLY the new token: per decoder layer it runs the q/k/v MLPs on one row per sequence,
appends k/v to the cache and attends with `qarig_attention_decode`; cross-attention keys and
values of the (constant) encoder output are computed once.  The logits equal the last row
of `Transformer.decode` on the full window up to fp32 summation order.

The step issues 100-200 small launches (skinny GEMMs, norms, one attention wave per head), so
it is launch-bound from Python; with `graph=True` the step is captured once into a HIP graph
(torch.cuda.CUDAGraph over the library's stream launches) and replayed per token.  What
changes between tokens - ids, pos_cond value, the window-index sinusoid row and the cache
length - lives in device buffers that are refreshed before each replay; the attention
kernel reads the length from device memory (`len_dev`).

The cache stops being valid when the window slides (every token's window index shifts);
`sampling.generate_tokens` falls back to the full-window evaluation from there on.
"""
import os

import torch

from models.layers import _lin_params, _mlp2_forward
from . import functional as QF
from . import ops

# LayerNorm (affine or AdaLN form) inside the launch of the Linear that reads it, the feed-forward
# gate multiply inside the launch of the Linear that produces its operand: 15 -> 11 dependent launches
# per encoder-decoder layer and token.  False: the separate launches (the tests compare both).
FUSE_NORMS = True


class DecodeCache:
    def __init__(self, model, enc, batch, max_len, graph=None):
        if not all(layer.self_attn_block.self_attn.use_masked_attn for layer in model.decoder_layers):
            raise ValueError("a KV cache needs causal decoder self-attention")
        self.model = model
        self.batch = batch
        self.max_len = max_len
        table = model.dec_embedding.weight
        self.dim = table.shape[1]
        dev = table.device
        n_layers = len(model.decoder_layers)
        # (layer, k|v, sequence, row, channel): one tensor so that beam bookkeeping can save
        # or restore a chunk of rows for every layer with one copy.
        self.kv = torch.zeros((n_layers, 2, batch, max_len, self.dim), dtype=torch.float32,
                              device=dev)
        self.pe = model._sequence_pe(max_len, self.dim, dev)
        self.cross = []
        with torch.no_grad():
            for layer in model.decoder_layers:
                if layer.use_cross_attn:
                    at = layer.cross_attn_block.cross_attn
                    self.cross.append((_mlp2_forward(at.k_block, enc).contiguous(),
                                       _mlp2_forward(at.v_block, enc).contiguous()))
                else:
                    self.cross.append(None)
        self._stack_weights()
        self._graph = None
        if graph is None:
            graph = os.environ.get("QARIG_DECODE_GRAPH", "1") != "0"
        if graph:
            self._capture()

    def _capture(self):
        """Static input buffers, one eager warm-up (sizes the GEMM workspaces outside the
        capture; it only touches cache row 0, which the first real step rewrites), capture."""
        dev = self.kv.device
        self._ids = torch.zeros(self.batch, dtype=torch.int64, device=dev)
        self._pos = torch.zeros(self.batch, dtype=torch.float32, device=dev)
        self._pe_row = torch.zeros((1, self.dim), dtype=torch.float32, device=dev)
        self._len = torch.zeros(1, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            self._forward(self._ids, self._pos, self._pe_row, 0, self._len)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            self._out = self._forward(self._ids, self._pos, self._pe_row, 0, self._len)
        self._graph = graph

    @torch.no_grad()
    def step(self, ids, pos, length):
        """ids (B,) int64: the token at window index `length`; pos (B,) fp32 `pos_cond` of
        that token or None.  Appends the token's keys/values and returns logits (B, V)."""
        if not 0 <= length < self.max_len:
            raise IndexError(f"cache position {length} outside [0, {self.max_len})")
        if self._graph is None:
            return self._forward(ids, pos, self.pe[length:length + 1], length, None)
        self._ids.copy_(ids.reshape(self.batch))
        if pos is not None:
            self._pos.copy_(pos.reshape(self.batch))
        self._pe_row.copy_(self.pe[length:length + 1])
        self._len.fill_(length)
        self._graph.replay()
        return self._out.clone()

    # -- launch-count reduction ----------------------------------------------------------------
    # A dependent kernel costs ~5 us in graph replay however small it is, so the step time
    # is the NUMBER of launches on the chain.  Two groups of small GEMMs are therefore
    # issued as one grouped launch each (qarig_gemm_grouped_skinny_f32) from weights stacked
    # once at construction: every projection of `cond` (AdaLN scale/shift and residual scale
    # layers of ALL decoder layers: 9 per enc-dec layer) and the q/k/v MLPs of a
    # self-attention layer (2 launches instead of 6); the residual layer's x * scale(cond)
    # rides in the attention kernel's output.  30 -> 15 launches per enc-dec layer.
    # (Issuing the independent work on side streams as parallel graph branches was measured
    # slower than the serial chain: cross-branch dependencies cost more than they hide.)
    def _stack_weights(self):
        model, D = self.model, self.dim
        ws, bs, self._proj_idx = [], [], []

        def add(lin):
            ws.append(lin.weight)
            bs.append(lin.bias)
            return len(ws) - 1

        ok = True
        for layer in model.decoder_layers:
            idx = {}
            blocks = [("self", layer.self_attn_block, "self_attn_norm", "self_attn_res")]
            if layer.use_cross_attn:
                blocks.append(("cross", layer.cross_attn_block, "cross_attn_norm", "cross_attn_res"))
            blocks.append(("ffn", layer.feedforward_block, "feedforward_norm", "feedforward_res"))
            for name, blk, norm_attr, res_attr in blocks:
                norm, res = getattr(blk, norm_attr), getattr(blk, res_attr)
                if blk.use_adaln0:
                    idx[name + "_norm"] = (add(norm.scale_layer.scale), add(norm.shift_layer.shift))
                if res.use_scale_layer:
                    idx[name + "_scale"] = add(res.scale_layer.scale)
            self._proj_idx.append(idx)
        ok = ok and all(w.shape == (D, D) for w in ws) and D % 256 == 0 and self.batch <= 512
        self._qkv = []
        for layer in model.decoder_layers:
            at = layer.self_attn_block.self_attn
            blocks = (at.q_block, at.k_block, at.v_block)
            w1 = [_lin_params(b[0])[0] for b in blocks]
            ok = ok and all(w.shape == w1[0].shape for w in w1) and w1[0].shape[0] % 256 == 0 \
                and len({b[0]._act for b in blocks}) == 1 and len({b[1]._act for b in blocks}) == 1
            self._qkv.append(blocks)
        self._stacked = bool(ok)
        if not self._stacked:
            return
        with torch.no_grad():
            if ws:
                self._proj_w = torch.stack([w.detach() for w in ws]).contiguous()
                self._proj_b = torch.stack([b.detach() for b in bs]).contiguous()
            packed = []
            for blocks in self._qkv:
                packed.append((
                    torch.stack([_lin_params(b[0])[0].detach() for b in blocks]).contiguous(),
                    torch.stack([_lin_params(b[0])[1].detach() for b in blocks]).contiguous(),
                    torch.stack([_lin_params(b[1])[0].detach() for b in blocks]).contiguous(),
                    torch.stack([_lin_params(b[1])[1].detach() for b in blocks]).contiguous(),
                    blocks[0][0]._act, blocks[0][1]._act))
            self._qkv = packed

    def _cond_projections(self, cond):
        """Per decoder layer: dict key -> tensor(s) (B,1,D) that depend on `cond` alone."""
        layers = self.model.decoder_layers
        if cond is None:
            return [{} for _ in layers]
        B, D = self.batch, self.dim
        if self._stacked and self._proj_idx and any(self._proj_idx):
            allp = ops.gemm_grouped_skinny(cond.reshape(B, D), self._proj_w, self._proj_b,
                                           shared_a=True)                       # (P, B, D)
            pick = lambda v: (allp[v[0]], allp[v[1]]) if isinstance(v, tuple) else allp[v]
            return [{k: pick(v) for k, v in idx.items()} for idx in self._proj_idx]
        out = []
        for layer in layers:
            proj = {}
            blocks = [("self", layer.self_attn_block, "self_attn_norm", "self_attn_res")]
            if layer.use_cross_attn:
                blocks.append(("cross", layer.cross_attn_block, "cross_attn_norm", "cross_attn_res"))
            blocks.append(("ffn", layer.feedforward_block, "feedforward_norm", "feedforward_res"))
            for name, blk, norm_attr, res_attr in blocks:
                norm, res = getattr(blk, norm_attr), getattr(blk, res_attr)
                if blk.use_adaln0:
                    proj[name + "_norm"] = (norm.scale_layer(cond), norm.shift_layer(cond))
                if res.use_scale_layer:
                    proj[name + "_scale"] = res.scale_layer(cond)
            out.append(proj)
        return out

    def _norm(self, norm, x, proj, key, use_adaln0):
        if use_adaln0:
            scale, shift = proj[key]
            return QF.layernorm_mod(x, scale.reshape(x.shape), shift.reshape(x.shape), norm.norm.eps)
        return QF.layernorm_affine(x, norm.weight, norm.bias, norm.eps)

    def _ln_mlp(self, norm, x, proj, key, use_adaln0, seq, mul=None, stacked=None):
        """The block's two-layer MLP on LayerNorm(x): the norm rides in the first Linear's launch
        (qarig_gemm_skinny_ln_f32) and `mul` -- the residual layer's scale(cond) -- in the second's.
        stacked: (w1, b1, w2, b2, act1, act2) with a leading group dimension (the q/k/v MLPs).
        Returns (B, D), or (G, B, D) for a stacked MLP; None when the shapes do not fit the fused launch."""
        B, D = self.batch, self.dim
        if stacked is not None:
            w1, b1, w2, b2, act1, act2 = stacked
        else:
            (w1, b1), (w2, b2) = _lin_params(seq[0]), _lin_params(seq[1])
            act1, act2 = seq[0]._act, seq[1]._act
        if D % 256 or B > 512 or w1.shape[-1] != D or w2.shape[-1] % 256 or b1 is None or b2 is None:
            return None
        x2 = x.reshape(B, D)
        if use_adaln0:
            scale, shift = proj[key]
            hid = ops.gemm_skinny_ln(x2, w1, b1, act1, scale=scale.reshape(B, D), shift=shift.reshape(B, D),
                                     eps=norm.norm.eps)
        else:
            hid = ops.gemm_skinny_ln(x2, w1, b1, act1, gamma=norm.weight, beta=norm.bias, eps=norm.eps)
        if stacked is not None:
            return ops.gemm_grouped_skinny(hid, w2, b2, act=act2)
        return ops.gemm_skinny_ln(hid, w2, b2, act2, mul=None if mul is None else mul.reshape(B, D))

    def _residual(self, res, x, x_skip, proj, key, scaled=False):
        """ResidualLinearLayer.forward with the scale projection supplied (`scaled`: the
        producer of x already applied it)."""
        if res.use_scale_layer and not scaled:
            x = QF.mul(x, proj[key].reshape(x.shape))
        w, b = _lin_params(res.linear)
        return QF.linear_act(x, w, b, residual=res.skip_linear(x_skip), act=res._act)

    def _forward(self, ids, pos, pe_row, length, len_dev):
        model, B, D = self.model, self.batch, self.dim
        x = QF.embedding_pos(ids.reshape(B, 1), model.dec_embedding.weight, pe_row)
        cond = None
        if model.use_pos_cond:
            cond = ops.posemb(pos.reshape(B), D).reshape(B, 1, D)
            cond = _mlp2_forward(model.pos_cond_layer, cond)
        projections = self._cond_projections(cond)
        for li, layer in enumerate(model.decoder_layers):
            proj = projections[li]
            sab = layer.self_attn_block
            at = sab.self_attn
            qkv = self._ln_mlp(sab.self_attn_norm, x, proj, "self_norm", sab.use_adaln0, None,
                               stacked=self._qkv[li]) if self._stacked and FUSE_NORMS else None
            if qkv is not None:
                q, k, v = qkv
            else:
                h = self._norm(sab.self_attn_norm, x, proj, "self_norm", sab.use_adaln0)
                if self._stacked:
                    w1, b1, w2, b2, act1, act2 = self._qkv[li]
                    hid = ops.gemm_grouped_skinny(h.reshape(B, D), w1, b1, act=act1, shared_a=True)
                    q, k, v = ops.gemm_grouped_skinny(hid, w2, b2, act=act2)
                else:
                    q = _mlp2_forward(at.q_block, h).reshape(B, D)
                    k = _mlp2_forward(at.k_block, h).reshape(B, D)
                    v = _mlp2_forward(at.v_block, h).reshape(B, D)
            o_mul = proj["self_scale"].reshape(B, D) if "self_scale" in proj else None
            o = ops.attention_decode(q, k, v, self.kv[li, 0], self.kv[li, 1], length, at.heads,
                                     len_dev=len_dev, o_mul=o_mul)
            x = self._residual(sab.self_attn_res, o.reshape(B, 1, D), x, proj, "self_scale", scaled=True)
            if layer.use_cross_attn:
                cab = layer.cross_attn_block
                at = cab.cross_attn
                ck, cv = self.cross[li]
                q = self._ln_mlp(cab.cross_attn_norm, x, proj, "cross_norm", cab.use_adaln0, at.q_block) \
                    if FUSE_NORMS else None
                if q is None:
                    h = self._norm(cab.cross_attn_norm, x, proj, "cross_norm", cab.use_adaln0)
                    q = _mlp2_forward(at.q_block, h).reshape(B, D)
                o_mul = proj["cross_scale"].reshape(B, D) if "cross_scale" in proj else None
                o = ops.attention_decode(q, None, None, ck, cv, ck.shape[1], at.heads, o_mul=o_mul)
                x = self._residual(cab.cross_attn_res, o.reshape(B, 1, D), x, proj, "cross_scale",
                                   scaled=True)
            fb = layer.feedforward_block
            gate = proj["ffn_scale"] if fb.feedforward_res.use_scale_layer else None
            h = self._ln_mlp(fb.feedforward_norm, x, proj, "ffn_norm", fb.use_adaln0, fb.feedforward, mul=gate) \
                if FUSE_NORMS else None
            if h is not None:
                x = self._residual(fb.feedforward_res, h.reshape(B, 1, D), x, proj, "ffn_scale", scaled=True)
            else:
                h = self._norm(fb.feedforward_norm, x, proj, "ffn_norm", fb.use_adaln0)
                h = _mlp2_forward(fb.feedforward, h)
                x = self._residual(fb.feedforward_res, h, x, proj, "ffn_scale")
        return _mlp2_forward(model.classifier, x).reshape(B, -1)

    def rows(self, lo, hi):
        """View of cache rows [lo, hi) of every layer: (layer, 2, B, hi-lo, D)."""
        return self.kv[:, :, :, lo:hi]
