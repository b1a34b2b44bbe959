"""The hot loop of train_quantized_transformer.py (reference :404-514) as reusable
steps: BMU tokenisation -> token assembly -> sliding window -> Transformer -> CE ->
backward -> (DP all-reduce) -> Adam.  Everything stays on the device; the only host
value is the window offset drawn from the CPU generator, as the reference does."""
import torch

from . import functional as QF
from . import parallel


def tokenize(feature_map, lr_codebook, hr_codebook, train_base_model):
    """(hr_input, lr_input, hr_target) int64 on the device.
    reference train_quantized_transformer.py:410-455"""
    N = feature_map.shape[0]
    lr_idx = lr_codebook.get_patches_bmu(feature_map, reshape=True)
    hr_idx = hr_codebook.get_patches_bmu(feature_map, reshape=True)
    k_lr, k_hr = lr_codebook.num_embeddings, hr_codebook.num_embeddings
    dev = feature_map.device
    if train_base_model:
        # LR token(s) act as <start>; HR ids are shifted into the combined vocabulary
        hr_input = torch.cat((lr_idx, hr_idx + k_lr), dim=1)
        lr_input = None
    else:
        start = torch.full((N, 1), k_hr, dtype=torch.int64, device=dev)
        hr_input = torch.cat((start, hr_idx), dim=1)
        lr_input = lr_idx
    end = torch.full((N, 1), k_hr, dtype=torch.int64, device=dev)
    hr_target = torch.cat((hr_idx, end), dim=1)
    return hr_input, lr_input, hr_target


def slide(hr_input, hr_target, window, rand_indices):
    """One window of `window` tokens per sample starting at rand_indices[n], plus the
    absolute positions used as conditioning.  reference :458-484 (unfold + gather)."""
    dev = hr_input.device
    offs = rand_indices.to(dev).unsqueeze(1) + torch.arange(window, device=dev).unsqueeze(0)
    return hr_input.gather(1, offs), hr_target.gather(1, offs), offs


def tokenize_window(feature_map, lr_codebook, hr_codebook, train_base_model, window, rand_indices):
    """tokenize + slide fused: BMU indices -> (hr_input, lr_input, hr_target, pos) of ONE window
    per sample, assembled by one kernel (no cat / full / gather / arange launches and no
    full-length sequences in HBM).  rand_indices None: no sliding window (pos None)."""
    from . import ops
    lr_idx = lr_codebook.get_patches_bmu(feature_map, reshape=True)
    hr_idx = hr_codebook.get_patches_bmu(feature_map, reshape=True)
    offs = rand_indices
    if offs is not None and not offs.is_cuda:
        # pinned + non-blocking: a copy from pageable memory would make the host wait for the device
        offs = offs.pin_memory().to(feature_map.device, non_blocking=True)
    hr_in, hr_tg, pos = ops.assemble_tokens(lr_idx, hr_idx, train_base_model, lr_codebook.num_embeddings,
                                            hr_codebook.num_embeddings, offs, window)
    return hr_in, (None if train_base_model else lr_idx), hr_tg, pos


def num_windows(seq_len, window):
    return seq_len - window + 1


def train_step(model, optim, hr_input, lr_input, hr_target, pos_idx, dp=True, pos_bound=None):
    """forward + CE + backward + gradient all-reduce + Adam.  Returns the loss tensor
    (device scalar; callers decide when to .item()).  pos_bound: exclusive upper bound of
    pos_idx (the un-windowed sequence length) -- sizes the model's position table without
    a device read-back."""
    optim.zero_grad()
    logits = model(x_dec=hr_input, x_enc=lr_input, pos_cond=pos_idx, pos_bound=pos_bound)
    loss = QF.cross_entropy(logits.view(-1, logits.shape[-1]), hr_target.flatten())
    loss.backward()
    w = parallel.world_size() if dp else 1
    if dp and not optim.finish_allreduce() and w > 1:   # overlapped buckets, else one exchange
        parallel.allreduce_flat(optim.flat_grad)
    optim.step(grad_scale=1.0 / w)
    return loss


class PinnedFeed:
    """Small per-step host values (window offsets, Adam's step scalars) on their way into the static
    device buffers of a replayed graph: through a ring of PINNED host slots and non-blocking copies,
    so that the host never waits for the device (a copy from pageable memory is synchronous: with it
    the host could not run ahead of a replayed step at all).  A slot is reused only after the copy
    that last read it has completed (one event per slot)."""

    def __init__(self, dev_tensor, slots=8):
        self.dev = dev_tensor
        self.slots = [torch.empty(dev_tensor.shape, dtype=dev_tensor.dtype, pin_memory=True) for _ in range(slots)]
        self.events = [None] * slots
        self.i = 0

    def push(self, host_values):
        i = self.i
        self.i = (i + 1) % len(self.slots)
        if self.events[i] is not None:
            self.events[i].synchronize()
        self.slots[i].copy_(host_values)
        self.dev.copy_(self.slots[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[i] = ev


class _SegmentedCapture:
    """forward + backward of one step recorded as a CHAIN of HIP graphs, cut wherever the backward
    pass completes a gradient bucket of the overlapped data-parallel all-reduce
    (FlatAdam.enable_allreduce_overlap): replay launches segment k, then -- eagerly, on RCCL's own
    stream, ordered behind the segment by the usual stream event -- the all-reduce of the bucket
    that segment completed, then segment k+1, so the exchange overlaps the replayed rest of
    backward exactly as it overlaps the eager one, and the host issues a dozen graph launches
    instead of ~1,200 kernels.  The collectives themselves are not captured (nothing about RCCL's
    capture support is assumed).  All segments allocate from one private memory pool and are
    replayed in capture order.  A segment is cut at the first parameter report AFTER a bucket has
    completed (never behind the last one), so no segment is empty."""

    def __init__(self, optim):
        self.optim = optim
        self.graphs = []
        self.bucket_after = []      # buckets whose all-reduce follows segment k

    def capture(self, body):
        import torch.distributed as dist
        opt = self.optim
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        pool = torch.cuda.graph_pool_handle()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())

        from . import _lib
        calls_at_begin = [0]

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=pool, capture_error_mode=mode)
            self.graphs.append(g)
            calls_at_begin[0] = _lib.N_CALLS

        ready = []

        def cut(bucket, params_left):
            """FlatAdam._grad_done: (None, parameters still to report) on entry, (b, 0) when bucket b
            has just become final."""
            if bucket is not None:
                ready.append(bucket)
            elif ready and params_left > 1:
                self.graphs[-1].capture_end()
                self.bucket_after.append(list(ready))
                del ready[:]
                begin()

        # backward on THIS thread: the cuts end / begin captures from the gradient hooks, and a
        # thread-local capture may only be ended by the thread that began it
        with torch.cuda.stream(side), torch.autograd.set_multithreading_enabled(False):
            begin()
            opt._capture_cut = cut if opt._overlap else None
            try:
                out = body()
            finally:
                opt._capture_cut = None
            if _lib.N_CALLS == calls_at_begin[0]:
                # every remaining parameter was reported by the launch that closed the previous
                # segment: give the trailing segment one (no-op) node rather than an empty graph
                opt._dev_step_buffer().mul_(1.0)
            self.graphs[-1].capture_end()
            self.bucket_after.append(list(ready))
        torch.cuda.current_stream().wait_stream(side)
        if opt._overlap:
            opt.reset_overlap_bookkeeping()
        return out

    def replay(self):
        opt = self.optim
        for g, buckets in zip(self.graphs, self.bucket_after):
            g.replay()
            for b in buckets:
                opt.launch_bucket(b)


class GraphedTrainStep:
    """The whole training step -- BMU tokenisation, window slicing, forward, cross-entropy,
    backward, Adam -- captured ONCE into a HIP graph (torch.cuda.CUDAGraph over the library's
    stream launches and torch's own small kernels) and replayed per batch.

    For per-GPU batches of a few thousand tokens (BASELINE config 4 hands each GPU 8 sequences)
    the step is ~1,700 launches of 20-60 us kernels and the Python issue rate, not the GPU, sets
    the step time; replay removes the host from the loop.  At config 2's 16 k tokens the step is
    GPU-bound already and this changes nothing.

    What varies between steps lives in static device buffers refreshed before each replay: the
    latent batch, the per-sample window offsets, and Adam's step size / bias correction
    (FlatAdam.advance_captured).  The first `warmup` calls run eagerly (they are real training
    steps; they also size every workspace and the allocator pools).

    Data parallel (more than one rank, or FlatAdam's overlapped all-reduce enabled): the step is a
    chain of graph segments cut at the gradient buckets (_SegmentedCapture) with the bucket
    all-reduces issued between them, and Adam (which must wait for the last collective) is one eager
    launch behind them."""

    def __init__(self, model, optim, lr_codebook, hr_codebook, train_base_model, window, warmup=2):
        self.segmented = parallel.world_size() > 1 or optim._overlap
        self.model, self.optim = model, optim
        self.lr_cb, self.hr_cb = lr_codebook, hr_codebook
        self.base, self.window = train_base_model, window
        self.warmup = warmup
        self.calls = 0
        self.graph = None
        self._rand_feed = None

    def _body(self, z, rand, captured):
        hr_in, lr_in, hr_tg, pos = tokenize_window(z, self.lr_cb, self.hr_cb, self.base, self.window,
                                                   rand if self.window is not None else None)
        lr_seq = (z.shape[2] // self.lr_cb.patch_dim[0]) * (z.shape[3] // self.lr_cb.patch_dim[1])
        seq = (z.shape[2] // self.hr_cb.patch_dim[0]) * (z.shape[3] // self.hr_cb.patch_dim[1]) + \
            (lr_seq if self.base else 1)
        self.optim.zero_grad()
        logits = self.model(x_dec=hr_in, x_enc=lr_in, pos_cond=pos, pos_bound=seq)
        loss = QF.cross_entropy(logits.view(-1, logits.shape[-1]), hr_tg.flatten())
        loss.backward()
        if self.segmented:
            if not captured:
                self._exchange_and_step()
        elif captured:
            self.optim.step_captured()
        else:
            self.optim.step()
        return loss.detach()

    def _exchange_and_step(self):
        w = parallel.world_size()
        if not self.optim.finish_allreduce() and w > 1:
            parallel.allreduce_flat(self.optim.flat_grad)
        self.optim.step(grad_scale=1.0 / w)

    def __call__(self, z, rand):
        """z: latent batch on the device; rand: int64 window offsets (host or device).
        Returns the loss (device scalar; a view of the graph's static output after capture)."""
        self.calls += 1
        if rand is None:                       # no sliding window: nothing to slice
            rand = torch.zeros(z.shape[0], dtype=torch.int64)
        if self.calls <= self.warmup:
            return self._body(z, rand.to(z.device), captured=False)
        if self.graph is None:
            self._z = z.clone()
            self._rand = rand.to(z.device).clone()
            self.optim._dev_step_buffer()      # must exist BEFORE capture: created inside, its
            torch.cuda.synchronize()           # zero-fill would be replayed ahead of every Adam
            if self.segmented:
                self.graph = _SegmentedCapture(self.optim)
                self._loss = self.graph.capture(lambda: self._body(self._z, self._rand, captured=True))
            else:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self._loss = self._body(self._z, self._rand, captured=True)
        else:
            if z.shape != self._z.shape:
                raise ValueError(f"GraphedTrainStep was captured for batches of shape {tuple(self._z.shape)}, "
                                 f"got {tuple(z.shape)}")
            self._z.copy_(z)
            if rand.is_cuda:
                self._rand.copy_(rand)
            else:
                if self._rand_feed is None:
                    self._rand_feed = PinnedFeed(self._rand)
                self._rand_feed.push(rand)
        if self.segmented:
            self.graph.replay()
            self._exchange_and_step()
            return self._loss
        self.optim.advance_captured()
        self.graph.replay()
        return self._loss
