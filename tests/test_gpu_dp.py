"""Data-parallel train step on the GPU with world_size 2 (two processes sharing the one
GPU of the test box, gloo transport for CUDA tensors; production uses the same code over
RCCL): after one DP step on two half-batches -- with the bucketed all-reduce overlapped
with backward, and without -- the weights equal those of a single-process step on the
whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _position_table_on():
    """The workers force the position-table conditioning (production shapes take it by
    themselves); the in-process reference step must run the same form."""
    from qarig import functional as QF
    old = QF.COND_TABLE_MIN_RATIO
    QF.COND_TABLE_MIN_RATIO = 0
    yield
    QF.COND_TABLE_MIN_RATIO = old


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WIDE = False     # True: D = 256 -> the position-table projections run as grouped launches


def _build():
    from models.Transformer import Transformer
    torch.manual_seed(11)
    if WIDE:
        m = Transformer(use_encoder=True, use_pos_cond=True, num_enc_layers=1, num_dec_layers=2,
                        num_enc_embedding=24, num_dec_embedding=40, self_attn_heads=32,
                        cross_attn_heads=32, transformer_in_dim=256, transformer_out_dim=33,
                        transformer_hidden_dim=512)
    else:
        m = Transformer(use_encoder=True, use_pos_cond=True, num_enc_layers=1, num_dec_layers=2,
                        num_enc_embedding=24, num_dec_embedding=40, self_attn_heads=4, cross_attn_heads=2,
                        transformer_in_dim=32, transformer_out_dim=33, transformer_hidden_dim=64)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return m


def _data():
    g = torch.Generator().manual_seed(0)
    N, S = 4, 12
    return (torch.randint(0, 40, (N, S), generator=g), torch.randint(0, 24, (N, 5), generator=g),
            torch.randint(0, 33, (N, S), generator=g),
            torch.randint(0, 50, (N, 1), generator=g) + torch.arange(S)[None])


def _worker(rank, world, port, overlap, q, backend="gloo", wide=False):
    global WIDE
    WIDE = wide
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    from qarig import functional as QF
    from qarig import optim as qoptim
    from qarig import parallel, pipeline
    QF.COND_TABLE_MIN_RATIO = 0        # position-table conditioning on, as in production shapes
    parallel.init(backend=backend, force=True)
    torch.cuda.set_device(0)
    m = _build().cuda()
    qoptim.BUCKET_ELEMS = 400_000 if wide else 20_000    # several buckets on these small models
    opt = qoptim.FlatAdam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
    parallel.broadcast_params(opt.flat_param)
    if overlap:
        opt.enable_allreduce_overlap(force=True)
        assert len(opt._bucket_range) > 2
    x, e, t, pos = (parallel.shard(v).cuda() for v in _data())
    for _ in range(2):
        loss = pipeline.train_step(m, opt, x, e, t, pos)
    if rank == 0:
        # by value (numpy): a torch tensor travels as a file descriptor that dies with this process
        q.put((opt.flat_param.detach().cpu().numpy(), float(loss.detach())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,wide,world", [(False, False, 2), (True, False, 2), (True, True, 2),
                                                # more than two ranks (one sample each): bucket ordering and completion
                                                # tracking with world > 2, the per-layer grouping of the projections
                                                (True, False, 4), (True, True, 4)])
def test_dp2_step_equals_single_process(overlap, wide, world):
    """wide: the workers' position-table projections are grouped per decoder layer (what
    torch.distributed with more than one rank selects) and report their parameters to the overlapped
    all-reduce from the grouped backward; the reference step groups them in one launch."""
    global WIDE
    from qarig import pipeline
    from qarig.optim import FlatAdam
    WIDE = wide
    m = _build().cuda()
    opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
    x, e, t, pos = (v.cuda() for v in _data())
    for _ in range(2):
        pipeline.train_step(m, opt, x, e, t, pos, dp=False)
    want = opt.flat_param.detach().cpu()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, overlap, q, "gloo", wide)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    got = None
    for _ in range(180):                      # poll: a crashed worker must not cost minutes
        try:
            got, _ = q.get(timeout=1)
            break
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    assert got is not None and all(p.exitcode == 0 for p in procs)
    got = torch.from_numpy(got)
    # two Adam steps: the second one sees weights that already differ by summation-order noise
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())


def test_rccl_single_rank_overlapped_allreduce_path():
    """The production transport: backend "nccl" (= RCCL).  The test box has one GPU, so the
    group has one rank; the step still goes through RCCL init, the bucketed asynchronous
    all-reduces issued from the backward thread and their stream ordering against the
    fused gradient kernels and the Adam launch.  A sum over one rank is the identity, so the
    weights must equal a step without any exchange bit for bit."""
    global WIDE
    from qarig import pipeline
    from qarig.optim import FlatAdam
    WIDE = False
    m = _build().cuda()
    opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
    x, e, t, pos = (v.cuda() for v in _data())
    for _ in range(2):
        pipeline.train_step(m, opt, x, e, t, pos, dp=False)
    want = opt.flat_param.detach().cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), True, q, "nccl"))
    p.start()
    import queue
    got = None
    for _ in range(180):
        try:
            got, _ = q.get(timeout=1)
            break
        except queue.Empty:
            if p.exitcode not in (None, 0):
                break
    p.join(timeout=30)
    if p.is_alive():
        p.kill()
    assert got is not None and p.exitcode == 0
    assert torch.equal(torch.from_numpy(got), want)


def _graph_build(base=True):
    from models.Codebook import Codebook
    from models.Transformer import Transformer
    from qarig.optim import FlatAdam
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(4)
    lr_cb = Codebook(patch_dim=(8, 8) if base else (4, 4), image_dim=(8, 8), image_channel=4, num_embeddings=16).cuda()
    hr_cb = Codebook(patch_dim=(2, 2), image_dim=(8, 8), image_channel=4, num_embeddings=32).cuda()
    with torch.no_grad():
        lr_cb.codebook.weight.copy_(torch.tanh(torch.randn(lr_cb.codebook.weight.shape, generator=g)))
        hr_cb.codebook.weight.copy_(torch.tanh(torch.randn((32, 16), generator=g)))
    m = Transformer(use_encoder=not base, use_pos_cond=True, num_enc_layers=None if base else 1, num_dec_layers=2,
                    num_enc_embedding=None if base else 16, num_dec_embedding=48 if base else 33,
                    self_attn_heads=8, cross_attn_heads=None if base else 8, transformer_in_dim=64,
                    transformer_out_dim=33, transformer_hidden_dim=128).cuda()
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().max() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return lr_cb, hr_cb, m, FlatAdam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))


def _graph_batches(n=5):
    g = torch.Generator().manual_seed(9)
    return [(torch.tanh(torch.randn((6, 4, 8, 8), generator=g)), torch.randint(0, 17 - 12 + 1, (6,), generator=g))
            for _ in range(n)]


def _eager_reference(base=True):
    from qarig import pipeline
    lr_cb, hr_cb, m, opt = _graph_build(base)
    losses = []
    for z, rand in _graph_batches():
        hr_in, lr_in, hr_tg = pipeline.tokenize(z.cuda(), lr_cb, hr_cb, base)
        hr_in, hr_tg, pos = pipeline.slide(hr_in, hr_tg, 12, rand)
        losses.append(float(pipeline.train_step(m, opt, hr_in, lr_in, hr_tg, pos, dp=False, pos_bound=17)))
    return opt.flat_param.detach().clone(), losses


def test_graphed_train_step_matches_eager_steps():
    """pipeline.GraphedTrainStep: after its eager warm-up the step is replayed from a captured
    HIP graph (tokenisation, forward, loss, backward, Adam with device-side step scalars);
    five steps on changing batches must leave the weights of five eager steps."""
    from qarig import pipeline
    want, losses_e = _eager_reference()
    lr_cb, hr_cb, m, opt = _graph_build()
    step = pipeline.GraphedTrainStep(m, opt, lr_cb, hr_cb, True, 12, warmup=2)
    losses_g = [float(step(z.cuda(), rand)) for z, rand in _graph_batches()]
    assert step.graph is not None and opt.step_count == 5
    assert torch.equal(opt.flat_param, want)
    assert losses_g == losses_e


def _graph_worker(rank, world, port, q, backend, base):
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    from qarig import functional as QF
    from qarig import optim as qoptim
    from qarig import parallel, pipeline
    QF.COND_TABLE_MIN_RATIO = 0
    parallel.init(backend=backend, force=True)
    torch.cuda.set_device(0)
    lr_cb, hr_cb, m, _ = _graph_build(base)
    qoptim.BUCKET_ELEMS = 40_000                     # several buckets -> several graph segments
    opt = qoptim.FlatAdam(m.parameters(), lr=1e-3, betas=(0.5, 0.999))
    parallel.broadcast_params(opt.flat_param)
    opt.enable_allreduce_overlap(force=True)
    nb = len(opt._bucket_range)
    step = pipeline.GraphedTrainStep(m, opt, lr_cb, hr_cb, base, 12, warmup=2)
    assert step.segmented
    for z, rand in _graph_batches():
        loss = step(parallel.shard(z).cuda(), parallel.shard(rand))
    segs = len(step.graph.graphs)
    launched = sorted(b for bs in step.graph.bucket_after for b in bs)
    if rank == 0:
        q.put((opt.flat_param.detach().cpu().numpy(), float(loss), segs, nb, launched))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,backend,base", [(1, "nccl", True), (2, "gloo", True), (2, "gloo", False)])
def test_segmented_graph_step_with_bucket_allreduces(world, backend, base):
    """GraphedTrainStep under data parallelism: forward + backward replayed as HIP-graph SEGMENTS cut
    at the gradient buckets, the bucket all-reduces issued between the segments (over RCCL with one
    rank: bit-identical to eager steps; over gloo with two ranks on half batches: the weights of
    single-process steps on the whole batches up to summation order), Adam behind the last one."""
    want, _ = _eager_reference(base)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, q, backend, base)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    got = None
    for _ in range(180):
        try:
            got = q.get(timeout=1)
            break
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    assert got is not None and all(p.exitcode == 0 for p in procs)
    flat, loss, segs, nb, launched = got
    flat = torch.from_numpy(flat)
    assert nb > 2 and segs > 1 and launched == list(range(nb)), (segs, nb, launched)
    if world == 1:
        assert torch.equal(flat, want.cpu())
    else:
        assert float((flat - want.cpu()).abs().max()) < 5e-5 * float(want.abs().max())
