"""Key/value cache for autoregressive decoding, and the device-resident sampling loop on top of it.

The reference samples every token by re-running the decoder over the whole window
(generate_images.py:283-307; train_quantized_transformer.py:600-640).  While the window has
not started to slide, the keys and values of tokens already in it never change: the
self-attention is causal, the sequence position of a token is its index in the window and
its `pos_cond` value is fixed when it is appended.  `DecodeCache.step` therefore evaluates
ONLY the new token: per decoder layer it runs the q/k/v MLPs on one row per sequence,
appends k/v to the cache and attends with `qarig_decode_attention`; cross-attention keys and
values of the (constant) encoder output are computed once.  The logits equal the last row
of `Transformer.decode` on the full window up to fp32 summation order.

The step is ~80 dependent launches whose time is memory LATENCY, not bytes (csrc/decode.hip), so:
  * every Linear runs on `decode_linear_kernel` (all loads issued before the first wait) with the
    LayerNorm in front of it, the residual layer's skip-add / activation and gate inside the launch;
  * `cond` depends on a token's position alone and the positions of a stage are known before its loop:
    with `positions` given the cache evaluates `pos_cond_layer` and EVERY ScaleLayer / ShiftLayer projection
    (models/layers.py:100-153, 258-304: 63 Linear layers, 66 MB of weights at README size) once per window
    position at construction; a step's first launch copies the current position's row;
  * the step is captured into a HIP graph; what changes between tokens (ids, cache length) lives in device
    buffers -- `ctl` (int32 control words, csrc/decode.hip DecCtl) -- that the kernels read.

`DecodeCache.begin_search / run_chunk / finish_search` keep the whole best-of-`num_beam` chunk search of
generate_images.py:256-345 on the device: sampling (`qarig_decode_sample`) and candidate bookkeeping
(`qarig_decode_decide` / `_rows` / `_commit` / `_advance`) are single launches between replays of the step's
graph, all reading and writing device state; the host enqueues them without reading anything back until the
stage is finished.

The cache stops being valid when the window slides (every token's window index shifts);
`sampling.generate_tokens` falls back to the full-window evaluation from there on.
"""
import os
from types import SimpleNamespace

import torch

from models.layers import _lin_params, _mlp2_forward
from . import functional as QF
from . import ops

# LayerNorm (affine or AdaLN form) inside the launch of the Linear that reads it, the feed-forward
# gate multiply inside the launch of the Linear that produces its operand: 15 -> 11 dependent launches
# per encoder-decoder layer and token.  False: the separate launches (the tests compare both).
FUSE_NORMS = True

_WARM_SHAPES = set()        # step shapes that have run eagerly once in this process (begin_search)


def _rows(t, B):
    """A (D,) row of the per-position table as the (B, D) operand the general kernels take."""
    return t if t.dim() == 2 else t.expand(B, t.shape[0]).contiguous()


class DecodeCache:
    def __init__(self, model, enc, batch, max_len, graph=None, positions=None):
        """positions: optional sequence of max_len floats, the `pos_cond` value of the token at each
        window index (known before the loop: generate_images.py:306-322 numbers them cur + tok + 1);
        given, the conditioning path is evaluated once per position here and `step` ignores `pos`."""
        if not all(layer.self_attn_block.self_attn.use_masked_attn for layer in model.decoder_layers):
            raise ValueError("a KV cache needs causal decoder self-attention")
        self.model = model
        self.batch = batch
        self.max_len = max_len
        table = model.dec_embedding.weight
        self.dim = table.shape[1]
        dev = table.device
        n_layers = len(model.decoder_layers)
        heads = {layer.self_attn_block.self_attn.heads for layer in model.decoder_layers}
        if len(heads) != 1:
            raise ValueError("a KV cache needs one self-attention head count for all decoder layers")
        self.heads = heads.pop()
        # (layer, k|v, sequence, head, row, head channel): one tensor so that beam bookkeeping can save or
        # restore a chunk of rows for every layer with one copy; head-major, so that the keys of a head are
        # contiguous (the attention kernel's lane-per-key loads then read 2-KB runs instead of 32 B per 2 KB).
        self.kv = torch.zeros((n_layers, 2, batch, self.heads, max_len, self.dim // self.heads),
                              dtype=torch.float32, device=dev)
        self.pe = model._sequence_pe(max_len, self.dim, dev)
        self.ctl = torch.zeros(ops.DECODE_CTL_WORDS, dtype=torch.int32, device=dev)
        self.cross = self._cross_kv(enc)
        self._enc_shape = None if enc is None else tuple(enc.shape)
        self._stack_weights()
        self._table = None
        if positions is not None and model.use_pos_cond and self._proj_lin and self.dim % 4 == 0:
            self._build_table(positions)
        self._graph = None
        self._search = None
        if graph is None:
            graph = os.environ.get("QARIG_DECODE_GRAPH", "1") != "0"
        if graph:
            self._capture()

    @torch.no_grad()
    def _cross_kv(self, enc):
        """Per decoder layer: the encoder memory's (k, v) of its cross-attention, head-major, or None."""
        out = []
        for layer in self.model.decoder_layers:
            if layer.use_cross_attn:
                at = layer.cross_attn_block.cross_attn
                hm = lambda t: t.reshape(t.shape[0], t.shape[1], at.heads, -1).permute(0, 2, 1, 3).contiguous()
                out.append((hm(_mlp2_forward(at.k_block, enc)), hm(_mlp2_forward(at.v_block, enc))))
            else:
                out.append(None)
        return out

    @torch.no_grad()
    def rebind(self, enc):
        """The cache for another generation with the SAME model weights, batch and window: the encoder memory's
        keys / values are recomputed into the tensors the captured graphs read, the sequence starts over.  What
        stays: the per-position table of conditioning projections, the stacked weights, the captured step graph
        and the search's buffers (sampling keeps such caches per model: `sampling.decode_cache`).  False when
        the encoder memory's shape differs (the caller builds a new cache)."""
        if (None if enc is None else tuple(enc.shape)) != self._enc_shape:
            return False
        for old, new in zip(self.cross, self._cross_kv(enc)):
            if old is not None:
                old[0].copy_(new[0])
                old[1].copy_(new[1])
        self.ctl.zero_()
        return True

    def _capture(self):
        """Static input buffers, one eager warm-up (sizes the GEMM workspaces outside the
        capture; it only touches cache row 0, which the first real step rewrites), capture."""
        dev = self.kv.device
        self._ids = torch.zeros(self.batch, dtype=torch.int64, device=dev)
        self._pos = torch.zeros(self.batch, dtype=torch.float32, device=dev)
        self.ctl.zero_()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            self._forward(self._ids, self._pos, 0, self.ctl)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            self._out = self._forward(self._ids, self._pos, 0, self.ctl)
        self._graph = graph

    @torch.no_grad()
    def step(self, ids, pos, length):
        """ids (B,) int64: the token at window index `length`; pos (B,) fp32 `pos_cond` of
        that token or None (ignored when the cache was built with `positions`).  Appends the
        token's keys/values and returns logits (B, V)."""
        if not 0 <= length < self.max_len:
            raise IndexError(f"cache position {length} outside [0, {self.max_len})")
        if self._graph is None and self._search is not None:      # the search's captured step serves single steps too
            s = self._search
            s.ids.copy_(ids.reshape(self.batch))
            self.ctl[0:1].fill_(length)
            s.g_step.replay()
            return s.logits.clone()
        if self._graph is None:
            return self._forward(ids.reshape(self.batch), pos, length, None)
        self._ids.copy_(ids.reshape(self.batch))
        if pos is not None and self._table is None:
            self._pos.copy_(pos.reshape(self.batch))
        self.ctl[0:1].fill_(length)
        self._graph.replay()
        return self._out.clone()

    # -- launch-count reduction ----------------------------------------------------------------
    # The step time is the number of launches on its dependent chain.  Two groups of small GEMMs are
    # issued as one grouped launch each from weights stacked once at construction: every projection of
    # `cond` (AdaLN scale/shift and residual scale layers of ALL decoder layers: 9 per enc-dec layer;
    # with `positions` they leave the step altogether) and the q/k/v MLPs of a self-attention layer
    # (2 launches instead of 6); the residual layer's x * scale(cond) rides in the attention kernel's output.
    # (Issuing the independent work on side streams as parallel graph branches was measured
    # slower than the serial chain: cross-branch dependencies cost more than they hide.)
    def _stack_weights(self):
        model, D = self.model, self.dim
        ws, bs, self._proj_idx, self._proj_lin = [], [], [], []

        def add(lin):
            ws.append(lin.weight)
            bs.append(lin.bias)
            self._proj_lin.append(lin)
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

    @torch.no_grad()
    def _build_table(self, positions):
        """(max_len, P * D): row L holds every projection of cond(positions[L]); the views of `_proj_row`
        (the row of the step in flight) are what the launches of a step read."""
        dev, D, L = self.kv.device, self.dim, self.max_len
        pos = torch.as_tensor(list(positions), dtype=torch.float32, device=dev)
        if pos.shape != (L,):
            raise ValueError(f"positions: {tuple(pos.shape)} values for {L} cache rows")
        cond = _mlp2_forward(self.model.pos_cond_layer, ops.posemb(pos, D).reshape(1, L, D)).reshape(L, D)
        P = len(self._proj_lin)
        if self._stacked and L <= 512:
            allp = ops.gemm_grouped_skinny(cond.contiguous(), self._proj_w, self._proj_b, shared_a=True)   # (P, L, D)
        else:
            allp = torch.stack([QF.linear_act(cond.reshape(1, L, D), lin.weight, lin.bias).reshape(L, D)
                                for lin in self._proj_lin])
        self._table = allp.permute(1, 0, 2).reshape(L, P * D).contiguous()
        self._proj_row = torch.zeros((P, D), dtype=torch.float32, device=dev)
        pick = lambda v: (self._proj_row[v[0]], self._proj_row[v[1]]) if isinstance(v, tuple) else self._proj_row[v]
        self._row_proj = [{k: pick(v) for k, v in idx.items()} for idx in self._proj_idx]

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
                    proj[name + "_norm"] = (norm.scale_layer(cond).reshape(B, D), norm.shift_layer(cond).reshape(B, D))
                if res.use_scale_layer:
                    proj[name + "_scale"] = res.scale_layer(cond).reshape(B, D)
            out.append(proj)
        return out

    def _norm(self, norm, x, proj, key, use_adaln0):
        if use_adaln0:
            scale, shift = proj[key]
            B = self.batch
            return QF.layernorm_mod(x, _rows(scale, B).reshape(x.shape), _rows(shift, B).reshape(x.shape),
                                    norm.norm.eps)
        return QF.layernorm_affine(x, norm.weight, norm.bias, norm.eps)

    def _ln_mlp(self, norm, x, proj, key, use_adaln0, seq, mul=None, stacked=None):
        """The block's two-layer MLP on LayerNorm(x): the norm rides in the first Linear's launch and `mul`
        -- the residual layer's scale(cond) -- in the second's.  stacked: (w1, b1, w2, b2, act1, act2) with a
        leading group dimension (the q/k/v MLPs).  Returns (B, D), or (G, B, D) for a stacked MLP; None when the
        shapes do not fit the fused launches."""
        B, D = self.batch, self.dim
        if stacked is not None:
            w1, b1, w2, b2, act1, act2 = stacked
        else:
            (w1, b1), (w2, b2) = _lin_params(seq[0]), _lin_params(seq[1])
            act1, act2 = seq[0]._act, seq[1]._act
        if D % 256 or B > 512 or w1.shape[-1] != D or w2.shape[-1] % 256 or b1 is None or b2 is None:
            return None
        x2 = x.reshape(B, D)
        H = w1.shape[-2]
        if B <= 16 and ops.decode_linear_supported(B, H, D, True) and ops.decode_linear_supported(B, D, H, False):
            if use_adaln0:
                scale, shift = proj[key]
                hid = ops.decode_linear(x2, w1, b1, act1, scale=scale, shift=shift, eps=norm.norm.eps)
            else:
                hid = ops.decode_linear(x2, w1, b1, act1, gamma=norm.weight, beta=norm.bias, eps=norm.eps)
            return ops.decode_linear(hid, w2, b2, act2, mul=mul)
        if use_adaln0:
            scale, shift = proj[key]
            hid = ops.gemm_skinny_ln(x2, w1, b1, act1, scale=_rows(scale, B), shift=_rows(shift, B),
                                     eps=norm.norm.eps)
        else:
            hid = ops.gemm_skinny_ln(x2, w1, b1, act1, gamma=norm.weight, beta=norm.bias, eps=norm.eps)
        if stacked is not None:
            return ops.gemm_grouped_skinny(hid, w2, b2, act=act2)
        return ops.gemm_skinny_ln(hid, w2, b2, act2, mul=None if mul is None else _rows(mul, B))

    def _residual(self, res, x, x_skip, proj, key, scaled=False):
        """ResidualLinearLayer.forward with the scale projection supplied (`scaled`: the
        producer of x already applied it)."""
        if res.use_scale_layer and not scaled:
            x = QF.mul(x, _rows(proj[key], self.batch).reshape(x.shape))
        w, b = _lin_params(res.linear)
        return QF.linear_act(x, w, b, residual=res.skip_linear(x_skip), act=res._act)

    def _forward(self, ids, pos, length, len_dev, out=None):
        """One decoder step on one new row per sequence.  len_dev: the control words (graph replay:
        the kernels read the cache length from ctl[0]) or None (the host value `length`).
        out: optional (B, V) buffer for the logits."""
        model, B, D = self.model, self.batch, self.dim
        if D % 4 == 0:
            tab = self._table
            x = ops.decode_embed(ids, model.dec_embedding.weight, self.pe, ctl=len_dev, length=length,
                                 proj_table=tab, proj_row=self._proj_row if tab is not None else None)
            x = x.reshape(B, 1, D)
        else:
            x = QF.embedding_pos(ids.reshape(B, 1), model.dec_embedding.weight, self.pe[length:length + 1])
            assert len_dev is None, "graph replay needs a model width that is a multiple of 4"
        if self._table is not None:
            projections = self._row_proj
        else:
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
            o_mul = proj["self_scale"] if "self_scale" in proj else None
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
                o_mul = proj["cross_scale"] if "cross_scale" in proj else None
                o = ops.attention_decode(q, None, None, ck, cv, ck.shape[2], at.heads, o_mul=o_mul)
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
        (w1, b1), (w2, b2) = _lin_params(model.classifier[0]), _lin_params(model.classifier[1])
        if out is not None and B <= 16 and b1 is not None and b2 is not None and \
                ops.decode_linear_supported(B, w1.shape[0], D, False) and \
                ops.decode_linear_supported(B, w2.shape[0], w2.shape[1], False):
            hid = ops.decode_linear(x.reshape(B, D), w1, b1, model.classifier[0]._act)
            ops.decode_linear(hid, w2, b2, model.classifier[1]._act, out=out)      # straight into the caller's buffer
            return out
        logits = _mlp2_forward(model.classifier, x).reshape(B, -1)
        if out is not None:
            out.copy_(logits)
            return out
        return logits

    def rows(self, lo, hi):
        """View of cache rows [lo, hi) of every layer: (layer, 2, B, H, hi-lo, d)."""
        return self.kv[:, :, :, :, lo:hi]

    # -- device-resident chunk search ----------------------------------------------------------
    @torch.no_grad()
    def begin_search(self, first_ids, images, beams, beam_width, temperature, end_token, shift, generate_mode,
                     max_chunks, candidates, forced=None, log_probs=False, generator=None, reference_order=False):
        """Prepares the search of generate_images.py:256-345 on `images` x `beams` cache rows (beams > 1:
        the candidate chunks of an image as rows of one batch; beams == 1: `candidates` chunks one after the
        other): evaluates the first token (window index 0), captures the step's graph, draws the uniforms of
        every draw the stage can make (ONE call of the device generator).  forced: optional (draws, columns)
        int64, entries >= 0 replace the draw (tests); log_probs: keep every probability row sampled from.

        reference_order (beams > 1): the candidates of a chunk are independent given the kept prefix, so they
        run as rows of one batch, but every draw keeps the number the reference's candidate-after-candidate
        loop gives it -- draw (chunk * beams + candidate) * beam_width + slot, one column per image -- and the
        first best candidate wins as in the reference's `>` comparison: the same tokens from the same draws as
        beams == 1 with `candidates` = beams, at the cost of the batched search."""
        N, NB, bw = int(images), int(beams), int(beam_width)
        B, D, dev = self.batch, self.dim, self.kv.device
        if N * NB != B or self.dim % 4:
            raise ValueError("begin_search: images * beams must equal the cache's batch (and the width be 4-aligned)")
        if self.model.use_pos_cond and self._table is None:
            raise ValueError("begin_search: a position-conditioned model needs the cache built with `positions`")
        V = _lin_params(self.model.classifier[1])[0].shape[0]
        ordered = bool(reference_order) and NB > 1
        if ordered and int(candidates) != 1:
            raise ValueError("begin_search: reference_order runs every candidate as a row (candidates must be 1)")
        per_chunk = (NB if ordered else int(candidates)) * bw       # draw rows one chunk position consumes
        cols = N if ordered else B
        draws = max(1, int(max_chunks) * per_chunk)
        # The buffers the captured step graph reads and writes, and the graph itself, are kept across searches of
        # the same geometry (a cache that sampling.decode_cache handed out again): no capture, no eager warm-up.
        s = self._search
        fresh = s is None or (s.N, s.NB, s.bw, s.V) != (N, NB, bw, V)
        if fresh:
            s = SimpleNamespace(N=N, NB=NB, bw=bw, V=V)
            s.ids = torch.zeros(B, dtype=torch.int64, device=dev)
            s.comb = torch.ones(B, dtype=torch.float32, device=dev)
            s.chunk = torch.zeros((B, bw), dtype=torch.int64, device=dev)
            s.best_p = torch.zeros(N, dtype=torch.float32, device=dev)
            s.best_chunk = torch.zeros((N, bw), dtype=torch.int64, device=dev)
            s.take = torch.zeros(N, dtype=torch.int32, device=dev)
            R = max(bw - 1, 1)
            s.staged = torch.zeros((self.kv.shape[0], 2, N, self.heads, R, D // self.heads), dtype=torch.float32,
                                   device=dev)
            s.tokens = torch.zeros((N, self.max_len + bw), dtype=torch.int64, device=dev)
            s.last = torch.zeros((B, V), dtype=torch.float32, device=dev)
            s.logits = torch.zeros((B, V), dtype=torch.float32, device=dev)
            s.g_step = None
        else:
            s.comb.fill_(1.0)
            s.tokens.zero_()
        s.draws, s.used, s.chunks, s.candidates = draws, 0, 0, int(candidates)
        s.beams, s.per_set = (NB if ordered else 0), (NB * bw if ordered else bw)
        s.uniforms = torch.rand((draws, cols), device=dev, generator=generator)
        s.forced = None
        if forced is not None:
            s.forced = torch.full((draws, cols), -1, dtype=torch.int64, device=dev)
            f = torch.as_tensor(forced, dtype=torch.int64, device=dev).reshape(-1, cols)[:draws]
            s.forced[:f.shape[0]] = f
        s.probs = torch.zeros((draws, cols, V), dtype=torch.float32, device=dev) if log_probs else None
        first = first_ids.reshape(N).to(torch.int64)
        s.tokens[:, 0] = first
        s.ids.copy_(first.repeat_interleave(NB))
        self.ctl.zero_()
        s.gen, s.T, s.end, s.shift = bool(generate_mode), float(temperature), int(end_token), int(shift)
        if s.g_step is None:
            # A capture records launches without running them, and the first launch of a kernel in a process (code
            # object load) or a workspace that has to grow cannot happen inside one: the first step of a given
            # shape in this process runs eagerly once; later stages of that shape go straight to the capture and
            # evaluate their first token by replaying it.
            sig = (B, D, V, self.heads, len(self.model.decoder_layers), tuple(c is not None for c in self.cross),
                   self._table is not None, self._stacked, FUSE_NORMS,
                   tuple(k.shape[2] for c in self.cross if c is not None for k in c[:1]))
            if sig not in _WARM_SHAPES:
                self._forward(s.ids, None, 0, self.ctl, out=s.logits)
                self.ctl.zero_()
                _WARM_SHAPES.add(sig)
            s.g_step = torch.cuda.CUDAGraph()
            with torch.cuda.graph(s.g_step):
                self._forward(s.ids, None, 0, self.ctl, out=s.logits)
        s.g_step.replay()                      # the first token, window index 0
        s.last.copy_(s.logits)
        # chunk search starts behind the first token: window index 1
        self.ctl.zero_()
        self.ctl[0:2].fill_(1)
        self._search = s
        return s

    def _draw(self, slot, src, inc):
        s = self._search
        ops.decode_sample(src, s.T, s.end, s.gen, s.shift, s.uniforms, self.ctl, slot, s.bw, s.ids, s.chunk, s.comb,
                          forced=s.forced, probs_log=s.probs, inc_len=inc, beams=s.beams)

    @torch.no_grad()
    def run_chunk(self, last=False):
        """One chunk position: `candidates` candidate chunks, then the kept one joins the sequence (and, unless
        `last`, its final token is evaluated for the next chunk's first draw).  Enqueues only -- the decoder step
        as one graph replay, sampling and bookkeeping as single launches -- and never synchronises."""
        s = self._search
        N, NB, bw = s.N, s.NB, s.bw
        for _ in range(s.candidates):
            self._draw(0, s.last, False)                    # every candidate's first draw: the kept prefix's logits
            for _ in range(1, bw):
                s.g_step.replay()                           # the token just drawn, at window index ctl[0]
                self._draw(-1, s.logits, True)              # slot = steps since the candidate began (ctl[4])
            ops.decode_decide(self.ctl, N, NB, bw, s.comb, s.chunk, s.best_p, s.best_chunk, s.take, draws=s.per_set)
            if bw > 1:
                ops.decode_rows(self.ctl, self.kv, s.staged, s.take, N, NB, restore=False)
        if bw > 1:
            ops.decode_rows(self.ctl, self.kv, s.staged, s.take, N, NB, restore=True)
        ops.decode_commit(self.ctl, N, NB, bw, s.best_chunk, s.tokens, s.ids)
        if not last:
            s.g_step.replay()
            s.last.copy_(s.logits)
        ops.decode_advance(self.ctl, bw)
        s.used += s.candidates * s.per_set
        s.chunks += 1

    def finish_search(self):
        """(N, 1 + chunks * beam_width) tokens of the sequences, first token included."""
        s = self._search
        return s.tokens[:, :1 + s.chunks * s.bw].clone()
