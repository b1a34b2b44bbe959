"""Autoregressive token sampling as the reference does it, on the device:
 - best-of-`num_beam` chunks of `beam_width` tokens with temperature sampling and the
   <end> probability zeroed (generate_images.py:256-345)           -> mode "generate";
 - plain ancestral sampling with the <end> -> 0 hack (the periodic sample inside
   train_quantized_transformer.py:581-656)                          -> mode "train".
The encoder half runs ONCE per stage (its input is constant across decode steps; the
reference recomputes it on every call).  Sampling uses torch.multinomial on the device
generator exactly like the reference on --device cuda, so a given --seed reproduces the
reference's draw sequence as long as the logits agree.

Reference quirks kept on purpose (drop-in behaviour):
 - "generate" numbers the appended window positions cur_len + tok + 1 (position 1 is
   never used: 0, 2, 3, ...), "train" numbers them 0, 1, 2, ...;
 - "generate" loops while len < total_seq, so beam_width must divide total_seq."""
import os

import torch

from . import ops
from .kvcache import DecodeCache


def _sample(logits, temperature, end_token, mode, rows, comb):
    """One sampling draw as the reference makes it; returns (next ids (B,1), comb)."""
    probs = torch.softmax(logits / temperature, dim=1)
    if mode == "generate":
        probs[:, end_token] = 0.0              # <end> removed from consideration
    nxt = torch.multinomial(probs, 1)
    comb = comb * probs[rows, nxt.squeeze(1)]
    if mode == "train":
        nxt[nxt == end_token] = 0              # reference HACK: <end> -> index 0
    return nxt, comb


def _cacheable(model, hr_input, use_sliding_window):
    if not (hr_input.shape[1] == 1 and hasattr(model, "decoder_layers")
            and bool(model.use_pos_cond) == bool(use_sliding_window)
            and all(l.self_attn_block.self_attn.use_masked_attn for l in model.decoder_layers)):
        return False
    # head dims that run zero-padded (QF.attention) keep the full-window loop: the cache kernel takes the
    # instantiated head dims only
    from . import ops
    return all(m.head_dim in ops.DECODE_HEAD_DIMS for m in model.modules() if hasattr(m, "head_dim"))


# Which sampler the cached loop uses: "fused" (default) draws inside the captured graphs
# (csrc/decode.hip qarig_decode_sample: inverse CDF from uniforms of the device generator, ONE generator call
# per stage; the whole chunk search stays on the device), "torch" makes one torch.multinomial call per token as
# the reference does on --device cuda (same generator stream as the reference for a given seed, ~8 extra
# launches and a graph boundary per token).  QARIG_SAMPLER overrides the default.
DEFAULT_SAMPLER = "fused"
ORDERED_ROWS = 512       # images x candidates the reference-order search runs as rows of one batch (0: one by one)
DECODE_ROWS = 16         # rows the single-token kernels take (csrc/decode.hip); more rows run the general kernels
GROUP_BELOW_ROWS = 64    # fused sampler: 16 < images x candidates < 64 is generated in groups of 16 rows (measured:
                         # 16 images x 4 candidates in groups 0.47 s per 256-token stage, 25 x 4 = 100 rows as one
                         # batch on the general kernels 0.51 s -- from about 64 rows up one batch is the faster form)
GROUP_IMAGES = True

# Test hook of the fused sampler: {"forced": (draws, rows) int64 tensor or None, "log": bool}; after a stage
# "probs" holds the (draws, rows, V) probability rows it sampled from and "draws" their number.
FUSED_DEBUG = None


class _Timer:
    """QARIG_GEN_TIMING=1: wall time of the phases of a stage on stderr (synchronises; diagnosis only)."""

    def __init__(self):
        self.on = os.environ.get("QARIG_GEN_TIMING") == "1"
        self.marks = []
        if self.on:
            import time
            torch.cuda.synchronize()
            self.clock, self.t = time.perf_counter, time.perf_counter()

    def mark(self, name):
        if self.on:
            torch.cuda.synchronize()
            t = self.clock()
            self.marks.append(f"{name} {(t - self.t) * 1e3:.2f} ms")
            self.t = t

    def report(self):
        if self.on:
            import sys
            print("[qarig generate] " + ", ".join(self.marks), file=sys.stderr)


# Decode caches kept per model between generations (a process that generates more than once with the same
# weights: a server, a CLI run over several batches): what a cache holds beyond the sequence itself -- the
# per-position table of the conditioning projections, the stacked weights, the captured step graph, the search's
# buffers -- depends on the weights, the batch and the window only.  Keyed by the model object and the state of
# every parameter (data pointer, torch's version counter, the owning FlatAdam's step count); a write into the
# weights that none of those sees (p.data arithmetic) needs decode_cache_clear().
_DECODE_CACHES = {}
DECODE_CACHE_SLOTS = 8


def decode_cache_clear():
    _DECODE_CACHES.clear()


def _weights_key(model):
    key = []
    for p in model.parameters():
        owner = getattr(p, "_qarig_owner", None)
        key.append((p.data_ptr(), p._version, owner.step_count if owner is not None else -1))
    return tuple(key)


def decode_cache(model, enc, batch, limit, positions):
    """A DecodeCache(model, enc, batch, limit, positions=positions) -- a kept one re-bound to `enc` when the
    model's weights have not changed since it was built, a new one otherwise."""
    import weakref
    slot = (id(model), batch, limit, None if positions is None else tuple(positions),
            None if enc is None else tuple(enc.shape))
    wkey = _weights_key(model)
    hit = _DECODE_CACHES.get(slot)
    if hit is not None and hit[0]() is model and hit[1] == wkey and hit[2].rebind(enc):
        return hit[2]
    cache = DecodeCache(model, enc, batch, limit, graph=False, positions=positions)
    if len(_DECODE_CACHES) >= DECODE_CACHE_SLOTS:
        _DECODE_CACHES.pop(next(iter(_DECODE_CACHES)))
    _DECODE_CACHES[slot] = (weakref.ref(model), wkey, cache)
    return cache


def _generate_fused(model, hr_input, enc, total_seq, temperature, use_sliding_window,
                    sliding_window, end_token, shift, num_beam, beam_width, mode, progress,
                    stop_len, pos_off, batch_beams):
    """The cached search with sampling, candidate bookkeeping and the decoder steps replayed from
    captured graphs (kvcache.DecodeCache.begin_search); nothing is read back before the stage ends.
    Returns (hr_input, pos, cache), or None when the model does not fit the fused kernels; the cache (rows of
    the kept tokens but the last) serves the evaluations of the next chunk that come before the window slides."""
    device = hr_input.device
    N = hr_input.shape[0]
    # The candidate chunks of a position are independent given the kept prefix: they run as rows of one batch
    # either way.  Without --batch-beams every draw keeps the number the reference's candidate loop gives it
    # (begin_search reference_order: same draws -> same tokens as one candidate after the other); ORDERED_ROWS = 0
    # runs the candidates literally one after the other (tests, A/B).
    ordered = not batch_beams and num_beam > 1 and N * num_beam <= ORDERED_ROWS
    B = num_beam if (batch_beams or ordered) and num_beam > 1 else 1
    cap = stop_len + beam_width
    limit = min(cap, sliding_window) if use_sliding_window else cap
    pos = torch.zeros((N, 1), device=device) if use_sliding_window else None
    cur = hr_input.shape[1]
    chunks = 0
    while cur + chunks * beam_width < stop_len and cur + (chunks + 1) * beam_width <= limit:
        chunks += 1
    if chunks == 0:
        return hr_input, pos, None
    positions = [0.0] + [float(L + pos_off) for L in range(1, limit)] if use_sliding_window else None
    enc_b = enc.repeat_interleave(B, dim=0) if (enc is not None and B > 1) else enc
    tm = _Timer()
    cache = decode_cache(model, enc_b, N * B, limit, positions)
    if cache.dim % 4 or (model.use_pos_cond and cache._table is None):
        return None
    tm.mark("cache")
    dbg = FUSED_DEBUG or {}
    after = cur + chunks * beam_width                      # tokens (first included) once the cached chunks are in
    tail_chunks = max(0, -(-(stop_len - after) // beam_width))          # chunks _fused_tail draws for
    cache.begin_search(hr_input[:, 0], N, B, beam_width, temperature, end_token, shift, mode == "generate",
                       chunks + tail_chunks, 1 if B > 1 else num_beam, forced=dbg.get("forced"),
                       log_probs=bool(dbg.get("log")), reference_order=ordered)
    tm.mark("capture")
    for c in range(chunks):
        cache.run_chunk(last=c == chunks - 1)
        if progress is not None:
            progress(cur + (c + 1) * beam_width - 1, total_seq)
    hr_input = cache.finish_search()
    tm.mark(f"{chunks} chunks")
    tm.report()
    if use_sliding_window:
        new = torch.tensor([float(L + pos_off) for L in range(cur, hr_input.shape[1])], device=device)
        pos = torch.cat((pos, new[None].expand(N, -1)), dim=1)
    return hr_input, pos, cache


def _fused_tail(model, cache, hr_input, pos, enc, use_sliding_window, sliding_window, beam_width, progress,
                total_seq, stop_len, pos_off, pos_bound):
    """The chunks behind the cached ones (the window slides inside them: every evaluation from there on is the
    reference's full-window evaluation, generate_images.py:275-286) on the state of the fused search: the candidates
    stay rows of one batch, every draw is made by the sampling kernel from the search's uniforms under the
    reference's draw numbers, the kept chunk is chosen by the decide kernel.  Evaluations that still fit the
    window (the chunk in which it starts to slide: all but its last) are replays of the cache's step graph.
    Enqueues only; nothing is read back."""
    s = cache._search
    N, NB, bw, ctl = s.N, s.NB, s.bw, cache.ctl
    B = N * NB
    device = hr_input.device
    enc_eval = enc.repeat_interleave(NB, dim=0) if (enc is not None and NB > 1) else enc
    whole = hasattr(model, "_cond")        # a Transformer: whole-number positions as int64 (one conditioning row each)
    start = 0
    while hr_input.shape[1] < stop_len:
        cur = hr_input.shape[1]
        for _ in range(s.candidates):
            t_in = hr_input.repeat_interleave(NB, dim=0) if NB > 1 else hr_input
            t_pos = (pos.repeat_interleave(NB, dim=0) if NB > 1 else pos) if use_sliding_window else None
            t_start = start
            for tok in range(bw):
                if use_sliding_window and t_in.shape[1] >= sliding_window:
                    t_start += 1
                    t_pos = t_pos[:, 1:]
                n = t_in.shape[1]
                if n - 1 < cache.max_len and (not use_sliding_window or n < sliding_window):
                    s.ids.copy_(t_in[:, -1])
                    ctl[0:1].fill_(n - 1)
                    s.g_step.replay()
                    logits = s.logits
                else:
                    # A slid window is window - 1 tokens: rows x 255 is no whole number of GEMM tiles and runs the
                    # guarded kernels.  One more token behind the last makes it one (16 x 256 = 4,096 rows) and
                    # changes nothing before it -- self-attention is causal, everything else is per token -- so the
                    # logits wanted are those of the last token but one.
                    win, wpos = t_in[:, t_start:], t_pos
                    rows_ = win.shape[0] * win.shape[1]
                    pad = rows_ % 128 != 0 and (rows_ + win.shape[0]) % 128 == 0
                    if pad:
                        win = torch.cat((win, win[:, -1:]), dim=1)
                        wpos = None if wpos is None else torch.cat((wpos, wpos[:, -1:]), dim=1)
                    if wpos is None or not whole:
                        logits = model.decode(win.contiguous(), enc_eval, wpos)
                    else:
                        logits = model.decode(win.contiguous(), enc_eval, wpos.long(), pos_bound=pos_bound)
                    logits = logits[:, -2 if pad else -1, :]
                ops.decode_sample(logits, s.T, s.end, s.gen, s.shift, s.uniforms, ctl, tok, bw, s.ids, s.chunk,
                                  s.comb, forced=s.forced, probs_log=s.probs, inc_len=False, beams=s.beams)
                t_in = torch.cat((t_in, s.ids[:, None]), dim=1)
                if use_sliding_window:
                    t_pos = torch.cat((t_pos, torch.full((B, 1), float(cur + tok + pos_off), device=device)), dim=1)
            ops.decode_decide(ctl, N, NB, bw, s.comb, s.chunk, s.best_p, s.best_chunk, s.take, draws=s.per_set)
        ops.decode_advance(ctl, bw)
        s.used += s.candidates * s.per_set
        hr_input = torch.cat((hr_input, s.best_chunk), dim=1)
        start = t_start
        if use_sliding_window:
            pos = t_pos[::NB] if NB > 1 else t_pos
        if progress is not None:
            progress(hr_input.shape[1] - 1, total_seq)
    return hr_input


def _generate_cached(model, hr_input, enc, total_seq, temperature, use_sliding_window,
                     sliding_window, end_token, shift, num_beam, beam_width, mode, progress,
                     stop_len, pos_off, batch_beams):
    """The same search as the loops below, evaluating one token per model call from a
    `DecodeCache` for as long as no evaluation of the next chunk would slide the window.
    Sampling draws are made in the reference's order (same shapes, same generator), so the
    sequential variant reproduces the full-window loop whenever the logits round alike.
    Returns (hr_input, pos) for the windowed loops to continue from."""
    device = hr_input.device
    N = hr_input.shape[0]
    B = num_beam if batch_beams and num_beam > 1 else 1
    # the loop overshoots stop_len by up to beam_width - 1 tokens (the reference's `while len < total`):
    # the last chunk of a sequence shorter than the window is cached like the others, instead of falling to
    # the full-window loop (whose ragged row counts run the guarded GEMM kernels: 14 % of a 3-stage cascade)
    cap = stop_len + beam_width
    limit = min(cap, sliding_window) if use_sliding_window else cap
    pos = torch.zeros((N, 1), device=device) if use_sliding_window else None
    cur = hr_input.shape[1]
    if cur + beam_width > limit:
        return hr_input, pos
    enc_b = enc.repeat_interleave(B, dim=0) if (enc is not None and B > 1) else enc
    cache = DecodeCache(model, enc_b, N * B, limit)
    rows = torch.arange(N * B, device=device)

    def positions(value):
        return torch.full((N * B,), float(value), device=device) if use_sliding_window else None

    last = cache.step(hr_input[:, 0].repeat_interleave(B), positions(0.0), 0)
    while cur < stop_len and cur + beam_width <= limit:
        best_chunk = best_p = best_rows = None
        for _ in range(1 if B > 1 else num_beam):
            comb = torch.ones(N * B, device=device)
            logits, new = last, []
            for tok in range(beam_width):
                nxt, comb = _sample(logits, temperature, end_token, mode, rows, comb)
                new.append(nxt + shift)
                if tok < beam_width - 1:
                    logits = cache.step(new[-1].squeeze(1), positions(cur + tok + pos_off),
                                        cur + tok)
            chunk = torch.cat(new, dim=1)
            if B > 1 or num_beam == 1:
                best_chunk, best_p = chunk, comb
                continue
            saved = cache.rows(cur, cur + beam_width - 1).clone() if beam_width > 1 else None
            if best_p is None:
                best_chunk, best_p, best_rows = chunk, comb, saved
            else:   # per sample, keep the candidate chunk with the larger probability product
                keep = best_p >= comb
                best_p = torch.where(keep, best_p, comb)
                best_chunk = torch.where(keep[:, None], best_chunk, chunk)
                if saved is not None:
                    best_rows = torch.where(keep[None, None, :, None, None, None], best_rows, saved)
        if B > 1:       # first beam with the maximal product wins, as in _generate_batched
            pick = torch.arange(N, device=device) * B + best_p.view(N, B).argmax(dim=1)
            best_chunk = best_chunk[pick]
            if beam_width > 1:
                r = cache.rows(cur, cur + beam_width - 1)
                r.copy_(r[:, :, pick].repeat_interleave(B, dim=2))
        elif best_rows is not None:
            cache.rows(cur, cur + beam_width - 1).copy_(best_rows)
        hr_input = torch.cat((hr_input, best_chunk.long()), dim=1)
        if use_sliding_window:
            pos = torch.cat([pos] + [torch.full((N, 1), float(cur + tok + pos_off), device=device)
                                     for tok in range(beam_width)], dim=1)
        cur += beam_width
        if progress is not None:
            progress(cur - 1, total_seq)
        if cur < stop_len and cur + beam_width <= limit:   # another cached chunk follows
            last = cache.step(best_chunk[:, -1].repeat_interleave(B),
                              positions(cur - 1 + pos_off), cur - 1)
    return hr_input, pos


@torch.no_grad()
def generate_tokens(model, hr_input, lr_input, total_seq, temperature, use_sliding_window,
                    sliding_window, end_token, shift=0, num_beam=1, beam_width=1, mode="generate",
                    progress=None, batch_beams=False, use_kv_cache=True, sampler=None):
    """hr_input: (N, S0) int64 conditioning/start tokens.  Returns the extended (N, S) tensor
    (first tokens included; callers strip them and undo `shift`).  use_kv_cache: evaluate one
    token per step from a key/value cache until the window starts to slide (same logits up
    to fp32 summation order); False re-runs the window for every token like the reference.
    sampler: "fused" / "torch" for the cached loop (DEFAULT_SAMPLER)."""
    assert mode in ("generate", "train")
    device = hr_input.device
    N = hr_input.shape[0]
    fused = (sampler or os.environ.get("QARIG_SAMPLER", DEFAULT_SAMPLER)) == "fused"
    # The single-token kernels behind the fused search take up to 16 rows (images x candidates).  A somewhat
    # larger batch is generated in groups of that many, one after the other through the model's kept decode
    # cache -- each group at the step time of 16 rows; from GROUP_BELOW_ROWS rows up the whole batch on the general
    # kernels is faster (the weights stream once per step for all rows).  The images are independent
    # (generate_images.py:256-345 loops over them only through the batch dimension).
    group = max(1, DECODE_ROWS // max(1, num_beam))
    rows_all = N * max(1, num_beam)
    if fused and GROUP_IMAGES and use_kv_cache and N > group and DECODE_ROWS < rows_all < GROUP_BELOW_ROWS and \
            _cacheable(model, hr_input, use_sliding_window):
        outs = []
        for g0 in range(0, N, group):
            sub = slice(g0, min(N, g0 + group))
            sub_progress = progress if g0 + group >= N else None
            outs.append(generate_tokens(model, hr_input[sub], None if lr_input is None else lr_input[sub], total_seq,
                                        temperature, use_sliding_window, sliding_window, end_token, shift, num_beam,
                                        beam_width, mode, sub_progress, batch_beams, use_kv_cache, sampler))
        return torch.cat(outs, dim=0)
    enc = model.encode(lr_input) if model.use_encoder else None
    pos = torch.zeros((N, 1), device=device) if use_sliding_window else None
    start = 0
    rows = torch.arange(N, device=device)
    stop_len = total_seq if mode == "generate" else hr_input.shape[1] + total_seq
    pos_off = 1 if mode == "generate" else 0
    cache = None
    if use_kv_cache and _cacheable(model, hr_input, use_sliding_window):
        args = (model, hr_input, enc, total_seq, temperature, use_sliding_window, sliding_window, end_token,
                shift, num_beam, beam_width, mode, progress, stop_len, pos_off, batch_beams)
        done = None
        if fused:
            done = _generate_fused(*args)
        if done is not None:
            hr_input, pos, cache = done
        else:
            hr_input, pos = _generate_cached(*args)

    pos_bound = stop_len + beam_width + pos_off + 1
    if cache is not None and cache._search is not None:
        tail = _Timer()
        hr_input = _fused_tail(model, cache, hr_input, pos, enc, use_sliding_window, sliding_window, beam_width,
                               progress, total_seq, stop_len, pos_off, pos_bound)
        tail.mark("windowed tail (fused draws)")
        tail.report()
        if FUSED_DEBUG is not None:
            FUSED_DEBUG["probs"] = cache._search.probs
            FUSED_DEBUG["draws"] = cache._search.used
        return hr_input

    def last_logits(t_in, t_start, t_pos):
        """Logits of the window's last token: the reference's full-window evaluation."""
        # the loop's positions are whole numbers (cur + tok + pos_off): as int64 the model evaluates the
        # conditioning path once per POSITION instead of once per token (Transformer._cond; same values)
        if t_pos is None or not hasattr(model, "_cond"):
            return model.decode(t_in[:, t_start:].contiguous(), enc_eval, t_pos)[:, -1, :]
        return model.decode(t_in[:, t_start:].contiguous(), enc_eval, t_pos.long(), pos_bound=pos_bound)[:, -1, :]

    tail = _Timer()
    enc_eval = enc
    if batch_beams and num_beam > 1:
        enc_eval = enc.repeat_interleave(num_beam, dim=0) if enc is not None else None
        out = _generate_batched(model, hr_input, last_logits, total_seq, temperature, use_sliding_window,
                                sliding_window, end_token, shift, num_beam, beam_width, mode,
                                progress, stop_len, pos_off, pos)
        tail.mark("windowed tail (batched beams)")
        tail.report()
        return out
    while hr_input.shape[1] < stop_len:
        cur = hr_input.shape[1]
        best_in = best_p = None
        for _ in range(num_beam):
            comb = torch.ones(N, device=device)
            t_start, t_in, t_pos = start, hr_input, pos
            for tok in range(beam_width):
                if use_sliding_window and t_in.shape[1] >= sliding_window:
                    t_start += 1
                    t_pos = t_pos[:, 1:]
                logits = last_logits(t_in, t_start, t_pos)
                probs = torch.softmax(logits / temperature, dim=1)
                if mode == "generate":
                    probs[:, end_token] = 0.0          # <end> removed from consideration
                nxt = torch.multinomial(probs, 1)
                comb = comb * probs[rows, nxt.squeeze(1)]
                if mode == "train":
                    nxt[nxt == end_token] = 0          # reference HACK: <end> -> index 0
                t_in = torch.cat((t_in, nxt + shift), dim=1)
                if use_sliding_window:
                    t_pos = torch.cat((t_pos, torch.full((N, 1), float(cur + tok + pos_off),
                                                         device=device)), dim=1)
            if best_p is None:
                best_in, best_p = t_in, comb
            else:  # keep, per sample, the candidate chunk with the larger probability product
                keep = best_p >= comb
                best_p = torch.where(keep, best_p, comb)
                best_in = torch.where(keep[:, None], best_in, t_in)
        start = t_start
        hr_input = best_in.long()
        if use_sliding_window:
            pos = t_pos
        if progress is not None:
            progress(hr_input.shape[1] - 1, total_seq)
    tail.mark("windowed tail")
    tail.report()
    return hr_input


def _generate_batched(model, hr_input, last_logits, total_seq, temperature, use_sliding_window,
                      sliding_window, end_token, shift, num_beam, beam_width, mode, progress,
                      stop_len, pos_off, pos):
    """Same search, with the `num_beam` independent candidate chunks evaluated as ONE batch
    of N*num_beam sequences per model call (the reference runs them one after the other).
    Additive option: 1/num_beam of the model calls; the device generator is consumed in a
    different order, so samples differ from the sequential loop for a given seed."""
    device = hr_input.device
    N, B = hr_input.shape[0], num_beam
    start = 0
    rows = torch.arange(N * B, device=device)
    while hr_input.shape[1] < stop_len:
        cur = hr_input.shape[1]
        t_in = hr_input.repeat_interleave(B, dim=0)                 # (N*B, S): beams of image n adjacent
        t_pos = pos.repeat_interleave(B, dim=0) if use_sliding_window else None
        comb = torch.ones(N * B, device=device)
        t_start = start
        for tok in range(beam_width):
            if use_sliding_window and t_in.shape[1] >= sliding_window:
                t_start += 1
                t_pos = t_pos[:, 1:]
            logits = last_logits(t_in, t_start, t_pos)
            probs = torch.softmax(logits / temperature, dim=1)
            if mode == "generate":
                probs[:, end_token] = 0.0
            nxt = torch.multinomial(probs, 1)
            comb = comb * probs[rows, nxt.squeeze(1)]
            if mode == "train":
                nxt[nxt == end_token] = 0
            t_in = torch.cat((t_in, nxt + shift), dim=1)
            if use_sliding_window:
                t_pos = torch.cat((t_pos, torch.full((N * B, 1), float(cur + tok + pos_off),
                                                     device=device)), dim=1)
        # first beam with the maximal product wins (the sequential loop keeps the earlier one on ties)
        best = comb.view(N, B).argmax(dim=1)
        pick = torch.arange(N, device=device) * B + best
        hr_input = t_in[pick].long()
        start = t_start
        if use_sliding_window:
            pos = t_pos[pick]
        if progress is not None:
            progress(hr_input.shape[1] - 1, total_seq)
    return hr_input
