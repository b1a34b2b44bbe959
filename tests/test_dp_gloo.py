"""Data-parallel path on CPU: world_size 2 over gloo (the GPU run uses the same code over
RCCL).  DP correctness = gradients after the sum all-reduce and the 1/world scale equal
the single-process gradients on the concatenated batch (CE is a mean over tokens and the
shards are equal).  Gradients come from the oracle model here -- the collective,
sharding and bucket logic under test are the product's qarig.parallel."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from qarig import parallel
    from oracle import ref_models as rm
    w, r, _ = parallel.init(backend="gloo")
    assert (w, r) == (world, rank) and parallel.world_size() == world
    g = load_golden("transformer_base_pos")
    cfg = dict(use_encoder=False, use_pos_cond=True, num_dec_layers=2, self_attn_heads=4,
               hidden_activation="silu")
    gen = torch.Generator().manual_seed(0)      # drawn globally, identically on every rank
    N = 4
    x = torch.randint(0, 40, (N, 12), generator=gen)
    t = torch.randint(0, 33, (N, 12), generator=gen)
    pos = torch.randint(0, 50, (N, 1), generator=gen) + torch.arange(12)[None]

    def grads(xs, ts, ps):
        sd = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
        rm.cross_entropy(rm.transformer_forward(sd, cfg, xs, None, ps), ts).backward()
        return torch.cat([sd[k].grad.reshape(-1) for k in sorted(sd)])

    local = grads(parallel.shard(x), parallel.shard(t), parallel.shard(pos))
    parallel.BUCKET_ELEMS = 10_000            # force several buckets
    works = parallel.allreduce_flat(local, async_op=True)
    assert len(works) > 1
    for wk in works:
        wk.wait()
    local.mul_(1.0 / world)
    p = torch.arange(8.0)
    parallel.broadcast_params(p if rank == 0 else p.zero_())
    assert torch.equal(p, torch.arange(8.0))
    # independent items sharded over the ranks (images of generate_images.py) and their rows gathered back:
    # 5 items -> 3 + 2, 1 item -> 1 + 0 (a rank with nothing still takes part)
    for n in (5, 1, 4):
        lo, hi = parallel.shard_range(n)
        per = -(-n // world)
        assert (lo, hi) == (min(n, rank * per), min(n, rank * per + per))
        mine = torch.arange(lo, hi)[:, None] * 10 + torch.arange(3)[None]          # (hi - lo, 3)
        allr = parallel.gather_rows(mine, n)
        assert torch.equal(allr, torch.arange(n)[:, None] * 10 + torch.arange(3)[None])
    if rank == 0:
        full = grads(x, t, pos)
        q.put((float((local - full).abs().max()), float(full.abs().max()),
               tuple(parallel.shard(x).shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_allreduce_equals_single_process_gradient():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err, scale, shp = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert shp == (2, 12)
    assert err <= 2e-6 * max(scale, 1e-6) + 1e-9


def test_world_one_is_a_noop():
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    from qarig import parallel
    t = torch.arange(6.0)
    assert parallel.world_size() == 1 and parallel.rank() == 0
    assert parallel.allreduce_flat(t) == [] and parallel.shard(t) is t
