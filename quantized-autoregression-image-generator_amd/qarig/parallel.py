"""Data parallelism: one process per GPU, replicas of all weights, the minibatch
sharded across ranks, and ONE exchange per step -- a sum all-reduce of the flat fp32
gradient buffer over RCCL/xGMI (backend "nccl" on ROCm), divided by world size inside
the Adam kernel (grad_scale).  Nothing else crosses GPUs (BASELINE.json north_star).

The reference has no distributed code at all; equal shards + mean-CE make this
reproduce the single-process gradient on the concatenated batch.
On CPU (tests) the same code runs over gloo.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, force=False):
    """Initialises torch.distributed from the torchrun environment (no-op at world 1 unless
    `force`, which tests use to drive the RCCL code path with a single rank)."""
    world, rank, local = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:     # QARIG_DIST_BACKEND=gloo: several ranks on one GPU (tests; RCCL wants a GPU per rank)
            backend = os.environ.get("QARIG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)     # RCCL: one GPU per rank
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


# xGMI is point-to-point (7 links per GPU): a few large messages keep every link busy;
# 64 MiB buckets let the tail of one overlap the head of the next.
BUCKET_ELEMS = 16 * 1024 * 1024


def allreduce_flat(flat, async_op=False):
    """Sum all-reduce of a flat tensor in large buckets.  Returns the work handles when
    async_op (caller waits before the optimiser step)."""
    if world_size() == 1:
        return []
    works = []
    n = flat.numel()
    for o in range(0, n, BUCKET_ELEMS):
        w = dist.all_reduce(flat[o:min(n, o + BUCKET_ELEMS)], op=dist.ReduceOp.SUM,
                            async_op=async_op)
        if async_op:
            works.append(w)
    return works


def shard(t, dim=0):
    """This rank's equal slice of a globally drawn tensor (RNG drawn once, globally,
    then sliced -- SURVEY.md 7-9)."""
    w, r = world_size(), rank()
    if w == 1:
        return t
    per = t.shape[dim] // w
    return t.narrow(dim, r * per, per)


def shard_range(n):
    """[lo, hi) of this rank's contiguous share of n independent items (images to generate, files to encode):
    ceil(n / world) each, the last ranks possibly fewer or none."""
    w, r = world_size(), rank()
    per = -(-n // w)
    lo = min(n, r * per)
    return lo, min(n, lo + per)


def gather_rows(t, n):
    """The shards of shard_range(n) -- this rank's (hi - lo, ...) tensor t -- as one (n, ...) tensor on every rank
    (an all-gather of equal-sized, zero-padded pieces: the only exchange of sharded generation, a few KB of
    token ids per stage)."""
    w = world_size()
    if w == 1:
        return t
    per = -(-n // w)
    pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(parts, pad)
    return torch.cat(parts, dim=0)[:n]


def broadcast_params(optim, src=0):
    """Rank `src`'s parameters to every rank.  optim: the FlatAdam that owns them (or its flat parameter
    buffer).  The broadcast writes the flat buffer behind torch's per-parameter version counters, so every
    copy derived from the old values -- the bf16 image of the reduced-precision mode, cached bf16 / e4m3 /
    re-ordered conv weights -- is declared stale here rather than left to a key that did not change."""
    flat = getattr(optim, "flat_param", optim)
    if world_size() > 1:
        dist.broadcast(flat, src=src)
    from . import ops
    ops.lp_invalidate()
    ops.bmu_invalidate()
    if hasattr(optim, "_shadow_versions"):
        optim._shadow_versions = None


def broadcast_host_tensor(t, src=0):
    """Makes a host-side (CPU) tensor identical on every rank: drawn on rank `src`, sent
    through the device when the backend is RCCL (which moves device buffers only)."""
    if world_size() == 1:
        return t
    if dist.get_backend() == "nccl":
        d = t.cuda()
        dist.broadcast(d, src=src)
        return d.cpu()
    dist.broadcast(t, src=src)
    return t
