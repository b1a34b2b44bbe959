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

The step issues ~200 small launches (skinny GEMMs, norms, one attention wave per head), so
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

from models.layers import _mlp2_forward, _norm_forward
from . import functional as QF
from . import ops


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

    def _forward(self, ids, pos, pe_row, length, len_dev):
        model, B, D = self.model, self.batch, self.dim
        x = QF.embedding_pos(ids.reshape(B, 1), model.dec_embedding.weight, pe_row)
        cond = None
        if model.use_pos_cond:
            cond = ops.posemb(pos.reshape(B), D).reshape(B, 1, D)
            cond = _mlp2_forward(model.pos_cond_layer, cond)
        for li, layer in enumerate(model.decoder_layers):
            sab = layer.self_attn_block
            at = sab.self_attn
            h = _norm_forward(sab.self_attn_norm, x, cond, sab.use_adaln0)
            q = _mlp2_forward(at.q_block, h).reshape(B, D)
            k = _mlp2_forward(at.k_block, h).reshape(B, D)
            v = _mlp2_forward(at.v_block, h).reshape(B, D)
            o = ops.attention_decode(q, k, v, self.kv[li, 0], self.kv[li, 1], length, at.heads,
                                     len_dev=len_dev)
            x = sab.self_attn_res(x=o.reshape(B, 1, D), x_skip=x, cond=cond)
            if layer.use_cross_attn:
                cab = layer.cross_attn_block
                at = cab.cross_attn
                ck, cv = self.cross[li]
                h = _norm_forward(cab.cross_attn_norm, x, cond, cab.use_adaln0)
                q = _mlp2_forward(at.q_block, h).reshape(B, D)
                o = ops.attention_decode(q, None, None, ck, cv, ck.shape[1], at.heads)
                x = cab.cross_attn_res(x=o.reshape(B, 1, D), cond=cond, x_skip=x)
            x = layer.feedforward_block(x, cond=cond)
        return _mlp2_forward(model.classifier, x).reshape(B, -1)

    def rows(self, lo, hi):
        """View of cache rows [lo, hi) of every layer: (layer, 2, B, hi-lo, D)."""
        return self.kv[:, :, :, lo:hi]
