"""Adam over ONE flat fp32 buffer.

All parameters are re-pointed (as views) into a single allocation, and so are their
gradients and both moments: the optimiser step is one kernel launch, zero_grad is one
memset, and the flat gradient buffer is exactly what the data-parallel all-reduce
moves over RCCL (qarig.parallel).  Semantics and state_dict layout follow
torch.optim.Adam as the reference uses it (betas=(0.5, 0.999), eps 1e-8, no weight
decay; train_quantized_transformer.py:317-334), so "model_optimizer" entries of
reference checkpoints load and save unchanged.
"""
import math

import torch
import torch.distributed as dist

from . import ops

# Gradient buckets for the overlapped data-parallel all-reduce: 16 Mi floats = 64 MiB keeps
# each xGMI link busy with few, large messages (ring collectives are per-link bound).
BUCKET_ELEMS = 16 * 1024 * 1024


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs the parameters on the GPU (call model.to(device) first)")
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False)
        self.param_groups = [dict(self.defaults, params=list(range(len(self.params))))]
        sizes = [p.numel() for p in self.params]
        # 32-B aligned slices: every fp32 view can be read with float4 loads, and the same offsets are 16-B
        # aligned in the bf16 image of the buffer (flat_shadow: LDS-DMA rows of the reduced-precision GEMMs)
        offs, total = [], 0
        for s in sizes:
            offs.append(total)
            total += (s + 7) // 8 * 8
        self.offsets, self.total = offs, total
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_count = 0
        self._overlap = False
        self._works = []
        self._capture_cut = None    # set by pipeline._SegmentedCapture while it records a step
        with torch.no_grad():
            for p, o, s in zip(self.params, offs, sizes):
                view = self.flat_param[o:o + s].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + s].view(p.shape)
                p._qarig_owner = self      # caches of derived data (ops.bmu_image) follow step_count

    # -- data-parallel overlap ----------------------------------------------------------
    def enable_allreduce_overlap(self, force=False):
        """Launch the RCCL sum all-reduce of each gradient bucket as soon as the backward
        pass has written every gradient in it (buckets are contiguous runs of the flat
        buffer in parameter order, i.e. whole layers; backward completes them last layer
        first), so the exchange overlaps the rest of backward.  Completion is tracked
        exactly: the fused weight/bias-gradient kernels report each parameter they finish
        (qarig.functional), every other parameter through a post-accumulate-grad hook.
        Assumes each parameter receives its gradient once per step (true for the models
        of this repository)."""
        if self._overlap or not (dist.is_available() and dist.is_initialized()) \
                or (dist.get_world_size() == 1 and not force):
            return
        self._overlap = True
        self._bucket_of, self._bucket_range, self._pending0 = {}, [], []
        lo, count, start_i = 0, 0, 0
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            end = o + (p.numel() + 3) // 4 * 4
            self._bucket_of[id(p)] = len(self._bucket_range)
            count += 1
            last = i == len(self.params) - 1
            if end - lo >= BUCKET_ELEMS or last:
                self._bucket_range.append((lo, end))
                self._pending0.append(count)
                lo, count = end, 0
        self._pending = list(self._pending0)
        self._done = set()
        self._launched = 0
        for p in self.params:
            p._qarig_grad_done = self._grad_done
            p.register_post_accumulate_grad_hook(self._grad_done)

    def _grad_done(self, p):
        # idempotent per step: a parameter may be reported both by the fused gradient
        # kernel path and by autograd's post-accumulate hook
        if id(p) in self._done:
            return
        if self._capture_cut is not None:
            self._capture_cut(None, len(self.params) - len(self._done))
        self._done.add(id(p))
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            if self._capture_cut is not None:
                # a step is being recorded into graph segments: the bucket's all-reduce is not part
                # of the recording, the segment ends here and replay issues the collective behind it
                self._capture_cut(b, 0)
            else:
                self.launch_bucket(b)

    def launch_bucket(self, b):
        lo, hi = self._bucket_range[b]
        self._works.append(dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self._launched += 1

    def reset_overlap_bookkeeping(self):
        """After a recording pass (no collective was issued, every parameter reported)."""
        self._pending = list(self._pending0)
        self._done = set()
        self._launched = 0

    def finish_allreduce(self):
        """Waits for the overlapped bucket all-reduces (no-op without overlap).  Returns
        True if this step's gradients have been reduced by the overlapped path."""
        if not self._overlap:
            return False
        # a replayed step (pipeline._SegmentedCapture) runs no hooks: its buckets were launched by
        # the replay loop, one per recorded cut
        replayed = not self._done and self._launched == len(self._bucket_range)
        if not replayed and any(n != 0 for n in self._pending):
            raise RuntimeError("overlapped all-reduce: some parameters never reported a gradient "
                               f"(pending per bucket: {self._pending})")
        if self._launched != len(self._bucket_range):
            raise RuntimeError(f"overlapped all-reduce: {self._launched} of {len(self._bucket_range)} "
                               "gradient buckets were exchanged")
        for w in self._works:
            w.wait()
        self._works = []
        self.reset_overlap_bookkeeping()
        return True

    # -- torch.optim API subset the reference's training loops use ------------------
    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def _step_scalars(self, step):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        return g["lr"] / (1 - b1 ** step), math.sqrt(1 - b2 ** step)

    # -- reduced precision: the bf16 copies of the weights come out of the Adam pass --------------------
    def _shadow_buffer(self):
        """Flat bf16 image of flat_param, written by the Adam kernel of every step while a reduced-precision
        mode is on (173 per-weight cast launches per step of the config-5 model otherwise)."""
        if not ops.lp_mode():
            return None
        if getattr(self, "flat_shadow", None) is None:
            self.flat_shadow = torch.empty(self.total, dtype=torch.bfloat16, device=self.flat_param.device)
            self._shadow_views = {}
        return self.flat_shadow

    def _shadow_written(self):
        # the kernel writes through raw pointers: torch's version counters stay where they are, so a later
        # in-place change of a parameter by anyone else (load_state_dict, a test) shows as a version mismatch
        self._shadow_versions = [p._version for p in self.params]

    def shadow_of(self, p):
        """bf16 view of parameter p out of the last Adam pass, or None (no pass yet in this mode, or p was
        modified since)."""
        if getattr(self, "_shadow_versions", None) is None or not ops.lp_mode():
            return None
        i = self._index.get(id(p)) if hasattr(self, "_index") else None
        if i is None:
            self._index = {id(q): k for k, q in enumerate(self.params)}
            i = self._index.get(id(p))
        if i is None or self._shadow_versions[i] != p._version:
            return None
        v = self._shadow_views.get(i)
        if v is None:
            o = self.offsets[i]
            v = self._shadow_views[i] = self.flat_shadow[o:o + p.numel()].view(p.shape)
        return v

    def step(self, grad_scale=1.0):
        self.step_count += 1
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        step_size, bc2_sqrt = self._step_scalars(self.step_count)
        shadow = self._shadow_buffer()
        ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, b1, b2,
                      g["eps"], step_size, bc2_sqrt, grad_scale, shadow=shadow)
        ops.lp_invalidate()      # cached bf16 / e4m3 copies made from the old weights are stale now
        if shadow is not None:
            self._shadow_written()
        else:
            self._shadow_versions = None     # (a step outside the mode leaves the bf16 image behind)

    # -- graph replay (qarig.pipeline.GraphedTrainStep) -----------------------------------
    def _dev_step_buffer(self):
        if not hasattr(self, "_dev_step"):
            self._dev_step = torch.zeros(2, dtype=torch.float32, device=self.flat_param.device)
        return self._dev_step

    def step_captured(self, grad_scale=1.0):
        """The Adam launch as it is recorded into a captured training-step graph: step size and
        bias correction are read from a device buffer that `advance_captured` refreshes on the
        host before every replay (learning-rate changes included)."""
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        shadow = self._shadow_buffer()
        ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, b1, b2,
                      g["eps"], 0.0, 1.0, grad_scale, dev_step=self._dev_step_buffer(), shadow=shadow)
        ops.lp_invalidate()
        if shadow is not None:
            self._shadow_written()
        else:
            self._shadow_versions = None

    def advance_captured(self):
        """Called before every replay of a captured step: the replayed Adam kernel rewrites the
        parameters without touching torch's version counters, so the reduced-precision weight
        shadows an eager forward may have cached (per-checkpoint sampling between replays) are stale
        from here on."""
        self.step_count += 1
        step_size, bc2_sqrt = self._step_scalars(self.step_count)
        if not hasattr(self, "_dev_step_feed"):
            from .pipeline import PinnedFeed
            self._dev_step_feed = PinnedFeed(self._dev_step_buffer())
        self._dev_step_feed.push(torch.tensor([step_size, bc2_sqrt], dtype=torch.float32))
        ops.lp_invalidate()

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        groups = [{k: v for k, v in g.items()} for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        for k in ("lr", "betas", "eps"):
            if k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]
        for i, st in sd["state"].items():
            i = int(i)
            o, n = self.offsets[i], self.params[i].numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(float(st["step"]))
        ops.lp_invalidate()
