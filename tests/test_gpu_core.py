"""GPU parity of the contraction core (GEMM) and the BMU search against the oracle.
Everything goes through the C ABI (qarig.ops -> ctypes -> libqarig_hip.so)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

GEMM_TOL = 2e-6  # |err| / max|ref| ; fp32 fma chain vs fp64 reference


def _ref_gemm(A, B, ak, bk):
    Ad = A.double().cpu() if ak else A.double().cpu().t()
    Bd = B.double().cpu() if bk else B.double().cpu().t()
    return Ad @ Bd.t()


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (257, 130, 70), (64, 513, 512), (1000, 96, 33),
                                   (2048, 512, 2048)])
@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, False), (False, True)])
def test_gemm_layouts(M, N, K, ak, bk):
    from qarig import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K) if ak else (K, M), generator=g).cuda()
    B = torch.randn((N, K) if bk else (K, N), generator=g).cuda()
    C = ops.gemm(A, B, ak, bk)
    ref = _ref_gemm(A, B, ak, bk)
    assert rel_err(C, ref) < GEMM_TOL * max(1, K / 512) ** 0.5


def test_gemm_epilogues():
    from qarig import ops
    from oracle import ref_models as rm
    g = torch.Generator().manual_seed(0)
    M, N, K = 300, 200, 96
    A = torch.randn((M, K), generator=g)
    W = torch.randn((N, K), generator=g) * 0.2
    b = torch.randn((N,), generator=g)
    R = torch.randn((M, N), generator=g)
    Z = torch.randn((M, N), generator=g)
    for act_name, act in (("silu", 1), ("tanh", 2), ("sigmoid", 3), (None, 0)):
        C, pre = ops.gemm(A.cuda(), W.cuda(), bias=b.cuda(), residual=R.cuda(), want_preact=True,
                          act=act)
        t = A.double() @ W.double().t() + b.double() + R.double()
        assert rel_err(pre, t) < GEMM_TOL
        assert rel_err(C, rm.activation(t, act_name)) < 5e-6
        # backward fusion: C = acc * act'(Z)
        Zd = Z.double().requires_grad_(True)
        rm.activation(Zd, act_name).sum().backward()
        G = ops.gemm(A.cuda(), W.cuda(), gradz=Z.cuda(), gact=act)
        assert rel_err(G, (A.double() @ W.double().t()) * Zd.grad) < 5e-6


def test_gemm_splitk_and_colsum():
    from qarig import ops
    g = torch.Generator().manual_seed(1)
    M, N, K = 200, 96, 5000
    A = torch.randn((K, M), generator=g).cuda()
    B = torch.randn((K, N), generator=g).cuda()
    ref = _ref_gemm(A, B, False, False)
    for s in (1, 3, 8):
        C = ops.gemm(A, B, False, False, splitk=s)
        assert rel_err(C, ref) < 1e-5
    C2 = ops.gemm(A, B, False, False, splitk=8)
    assert torch.equal(C, C2)  # deterministic slab order
    X = torch.randn((3001, 130), generator=g).cuda()
    assert rel_err(ops.colsum(X), X.double().sum(0)) < 2e-6


@pytest.mark.parametrize("M,N,K", [(1, 16, 256), (4, 2048, 512), (16, 512, 2048), (17, 513, 512),
                                   (33, 100, 768), (64, 2048, 512), (48, 7, 256), (65, 300, 512),
                                   (100, 513, 2048), (256, 64, 256), (384, 512, 512)])
def test_gemm_skinny_decode_shapes(M, N, K):
    """M <= 512, K % 256 == 0, both operands reduction-contiguous: the weight-streaming
    16x16x4-MFMA kernel (decode steps), every epilogue option, strided A, ragged N."""
    from qarig import ops
    from oracle import ref_models as rm
    g = torch.Generator().manual_seed(M + N + K)
    big = torch.randn((M, K + 4), generator=g).cuda()
    A = big[:, 4:]                      # 16-B aligned rows, ld = K + 4
    W = (torch.randn((N, K), generator=g) * 0.1).cuda()
    b = torch.randn((N,), generator=g).cuda()
    R = torch.randn((M, N), generator=g).cuda()
    Z = torch.randn((M, N), generator=g).cuda()
    t = A.double().cpu() @ W.double().cpu().t()
    assert rel_err(ops.gemm(A, W), t) < GEMM_TOL * max(1, K / 512) ** 0.5
    for act_name, act in (("silu", 1), (None, 0)):
        C, pre = ops.gemm(A, W, bias=b, residual=R, want_preact=True, act=act)
        tt = t + b.double().cpu() + R.double().cpu()
        assert rel_err(pre, tt) < GEMM_TOL * max(1, K / 512) ** 0.5
        assert rel_err(C, rm.activation(tt, act_name)) < 5e-6 * max(1, K / 512) ** 0.5
    Zd = Z.double().cpu().requires_grad_(True)
    rm.activation(Zd, "silu").sum().backward()
    G = ops.gemm(A, W, gradz=Z, gact=1)
    assert rel_err(G, t * Zd.grad) < 5e-6 * max(1, K / 512) ** 0.5
    assert torch.equal(ops.gemm(A, W), ops.gemm(A, W, splitk=4))    # own K split, deterministic


@pytest.mark.parametrize("M,N,K,ak,bk,splitk", [
    (2048, 2048, 512, True, True, 1),      # 256 tiles: Linear 512->2048 of an 8-sequence shard
    (2048, 2048, 512, True, False, 1),     # input gradient of 2048->512... operand order (kc, xc)
    (2048, 512, 2048, True, True, 4),      # 64 tiles x 4 slabs
    (2048, 512, 2048, False, False, 4),    # weight gradient (xc, xc) with the A row sums riding on it
    (1024, 1024, 256, True, True, 1),      # 64 tiles, 8 k-tiles per team
    (512, 640, 1024, False, False, 2),     # ragged tile count (4 x 5), two slabs
])
def test_gemm_paired_teams(M, N, K, ak, bk, splitk):
    """Launches of at most one workgroup per CU run the paired kernel (two 4-wave teams per tile,
    each reducing half of the k-range, accumulators swapped through LDS): every epilogue option
    against fp64, the bias-gradient row sums, determinism, and agreement with the one-team
    kernels up to the changed summation order."""
    from qarig import ops
    from oracle import ref_models as rm
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + splitk)
    A = torch.randn((M, K) if ak else (K, M), generator=g).cuda()
    B = (torch.randn((N, K) if bk else (K, N), generator=g) * 0.1).cuda()
    ref = _ref_gemm(A, B, ak, bk)
    tol = GEMM_TOL * max(1, K / 512) ** 0.5
    C = ops.gemm(A, B, ak, bk, splitk=splitk)
    assert rel_err(C, ref) < tol
    assert torch.equal(C, ops.gemm(A, B, ak, bk, splitk=splitk))
    if splitk == 1:
        b = torch.randn((N,), generator=g).cuda()
        R = torch.randn((M, N), generator=g).cuda()
        Z = torch.randn((M, N), generator=g).cuda()
        for act_name, act in (("silu", 1), ("tanh", 2), (None, 0)):
            Y, pre = ops.gemm(A, B, ak, bk, bias=b, residual=R, want_preact=True, act=act)
            t = ref + b.double().cpu() + R.double().cpu()
            assert rel_err(pre, t) < tol
            # |t| reaches 14 here: its 5e-7 relative error is 7e-6 absolute, which tanh passes
            # through unchanged near 0 where |tanh| <= 1 is the scale of the comparison
            assert rel_err(Y, rm.activation(t, act_name)) < (2e-5 if act_name == "tanh" else 5e-6)
        Zd = Z.double().cpu().requires_grad_(True)
        rm.activation(Zd, "silu").sum().backward()
        assert rel_err(ops.gemm(A, B, ak, bk, gradz=Z, gact=1), ref * Zd.grad) < 5e-6
        out = torch.randn((M, N), generator=g).cuda()
        want = out.double().cpu() + ref
        ops.gemm(A, B, ak, bk, out=out, accumulate=True)
        assert rel_err(out, want) < tol
    if not ak:
        rs = torch.zeros(M, device="cuda")
        C2 = ops.gemm(A, B, ak, bk, splitk=splitk, a_rowsum=rs)
        assert torch.equal(C2, C)
        assert rel_err(rs, A.double().cpu().sum(0)) < 2e-6 * max(1, K / 512) ** 0.5


@pytest.mark.parametrize("M,N,K,ak,bk,splitk", [
    (2048, 512, 512, True, True, 1),       # the 512 -> 512 residual Linear of an 8-sequence shard: 256 tiles of 64 x 64
    (2048, 512, 512, True, False, 1),      # its input gradient
    (512, 512, 2048, False, False, 4),     # its weight gradient: 64 tiles x 4 slabs, bias row sums riding on it
    (1024, 512, 2048, True, True, 2),      # a 4 x 255-row window evaluation's second MLP layer
    (64, 64, 16, True, True, 1),           # one tile, one k-tile
    (192, 320, 48, False, True, 1),        # tile-contiguous A against reduction-contiguous B
    (4096, 2048, 512, True, True, 1),      # forced onto 64-tiles although 128-tiles would fill the chip
])
def test_gemm_64_tiles(M, N, K, ak, bk, splitk):
    """csrc/gemm64.hip: products whose 128 x 128 tiling leaves CUs idle run on 64 x 64 tiles -- every operand
    layout, every epilogue option, split reductions with the bias row sums, against fp64, and against the
    128-tile kernels (option gemm_tile64 = 0) up to the summation order."""
    from qarig import ops, _lib
    from oracle import ref_models as rm
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + splitk)
    A = torch.randn((M, K) if ak else (K, M), generator=g).cuda()
    B = (torch.randn((N, K) if bk else (K, N), generator=g) * 0.1).cuda()
    b = torch.randn((N,), generator=g).cuda()
    R = torch.randn((M, N), generator=g).cuda()
    Z = torch.randn((M, N), generator=g).cuda()
    ref = _ref_gemm(A, B, ak, bk)
    tol = GEMM_TOL * max(1, K / 512) ** 0.5

    def run():
        out = [ops.gemm(A, B, ak, bk, splitk=splitk)]
        Y, pre = ops.gemm(A, B, ak, bk, bias=b, residual=R, want_preact=True, act=1, splitk=splitk)
        out += [Y, pre, ops.gemm(A, B, ak, bk, gradz=Z, gact=1, splitk=splitk)]
        acc = R.clone()
        ops.gemm(A, B, ak, bk, out=acc, accumulate=True, splitk=splitk)
        out.append(acc)
        if not ak:
            rs = torch.zeros(M, device="cuda")
            out += [ops.gemm(A, B, ak, bk, splitk=splitk, a_rowsum=rs), rs]
        return out
    old = _lib.set_option("gemm_tile64", 1)
    try:
        assert _lib.load().qarig_gemm_tile64(M, N, K) == 1
        got = run()
        assert torch.equal(got[0], ops.gemm(A, B, ak, bk, splitk=splitk))      # deterministic
        _lib.set_option("gemm_tile64", 0)
        assert _lib.load().qarig_gemm_tile64(M, N, K) == 0
        other = run()
    finally:
        _lib.set_option("gemm_tile64", old)
    t = ref + b.double().cpu() + R.double().cpu()
    Zd = Z.double().cpu().requires_grad_(True)
    rm.activation(Zd, "silu").sum().backward()
    want = [ref, rm.activation(t, "silu"), t, ref * Zd.grad, R.double().cpu() + ref]
    if not ak:
        want += [ref, A.double().cpu().sum(0)]
    for i, (x, y, w) in enumerate(zip(got, other, want)):
        assert rel_err(x, w) < (5e-6 if i in (1, 3) else tol) * (2 if i == 6 else 1), i
        assert rel_err(x, y) < 2 * tol, i


def test_gemm_tile64_is_the_default_for_small_grids():
    from qarig import _lib
    lib = _lib.load()
    assert lib.qarig_gemm_tile64(2048, 512, 512) == 1 and lib.qarig_gemm_tile64(1020, 512, 512) == 0
    assert lib.qarig_gemm_tile64(16384, 2048, 512) == 0          # 128-tiles fill the chip
    from qarig import ops
    assert lib.qarig_gemm_tile64(2048, 512, 2048) == 0           # long reduction on 64 tiles of 128: the split-K ring
    assert ops.auto_splitk(2048, 512, 512) == 1 and ops.auto_splitk(1024, 512, 512) == 2
    assert ops.pick_splitk(512, 512, 2048) == 4


def test_gemm_xcd_split_mapping():
    """Split reductions whose split count is a multiple of 8 run one split per XCD on a flat grid (placement
    only): the weight-gradient shapes of the bench step with 8 and 16 splits, row sums riding, accumulate --
    against fp64 and bit-identical to the tiles-per-XCD mapping."""
    from qarig import _lib, ops
    g = torch.Generator().manual_seed(21)
    for (M, N, K, sk) in ((2048, 512, 16384, 8), (512, 2048, 4096, 16), (256, 384, 2048, 8)):
        A = torch.randn((K, M), generator=g).cuda()
        B = (torch.randn((K, N), generator=g) * 0.1).cuda()
        ref = _ref_gemm(A, B, False, False)
        rs = torch.zeros(M, device="cuda")
        C = ops.gemm(A, B, False, False, splitk=sk, a_rowsum=rs)
        assert rel_err(C, ref) < GEMM_TOL * (K / 512) ** 0.5
        assert rel_err(rs, A.double().cpu().sum(0)) < 2e-6 * (K / 512) ** 0.5
        old = _lib.set_option("gemm_xcd_splits", 0)
        try:
            rs0 = torch.zeros(M, device="cuda")
            C0 = ops.gemm(A, B, False, False, splitk=sk, a_rowsum=rs0)
        finally:
            _lib.set_option("gemm_xcd_splits", old)
        assert torch.equal(C, C0) and torch.equal(rs, rs0)
        out = torch.randn((M, N), generator=g).cuda()
        want = out.double().cpu() + ref
        ops.gemm(A, B, False, False, splitk=sk, out=out, accumulate=True)
        assert rel_err(out, want) < GEMM_TOL * (K / 512) ** 0.5
        Af = torch.randn((M, K), generator=g).cuda()          # a forward-shaped product with 8 splits
        Bf = (torch.randn((N, K), generator=g) * 0.1).cuda()
        assert rel_err(ops.gemm(Af, Bf, splitk=8), _ref_gemm(Af, Bf, True, True)) < GEMM_TOL * (K / 512) ** 0.5


def test_gemm_strided_views():
    from qarig import ops
    g = torch.Generator().manual_seed(2)
    big = torch.randn((300, 513), generator=g).cuda()
    A = big[:, 1:101]           # ld 513, not 16-B aligned -> scalar path
    W = torch.randn((64, 100), generator=g).cuda()
    C = ops.gemm(A, W)
    assert rel_err(C, A.double().cpu() @ W.double().cpu().t()) < GEMM_TOL


BMU_CASES = ["trained_p1", "trained_p2", "trained_p4", "trained_p8", "trained_full", "ragged",
             "fresh_p4", "ties", "direct"]


@pytest.mark.parametrize("case", BMU_CASES)
def test_bmu_vs_oracle_and_golden(case):
    """HIP BMU == C oracle bit-for-bit on every case (same fp32 fma order), and ==
    the reference's golden indices wherever the oracle is."""
    from qarig import ops
    from oracle import bmu as obmu
    g = load_golden("bmu")[case]
    p = int(g["p"])
    got = ops.bmu(g["x"].cuda(), g["w"].cuda(), (p, p)).cpu().numpy()
    want = obmu.bmu(g["x"].numpy(), g["w"].numpy(), (p, p))
    assert got.dtype == np.int64
    assert np.array_equal(got, want)
    if case != "fresh_p4":
        assert np.array_equal(got, g["idx"].numpy().reshape(-1))


@pytest.mark.parametrize("N,C,H,W,p,K", [(64, 4, 32, 32, 4, 512), (64, 4, 32, 32, 32, 512),
                                         (16, 4, 64, 64, 1, 8192), (64, 4, 32, 32, 2, 512),
                                         (5, 3, 18, 30, 3, 100),
                                         # few rows, long patches (the conditional codebook's kernel): chunk counts that
                                         # are not whole, a patch width that is not a multiple of 4
                                         (5, 3, 24, 24, 24, 77), (2, 1, 20, 20, 20, 30), (9, 2, 18, 18, 18, 130)])
def test_bmu_seeded_vs_oracle(N, C, H, W, p, K):
    from qarig import ops
    from oracle import bmu as obmu
    g = torch.Generator().manual_seed(N + p + K)
    x = torch.tanh(torch.randn((N, C, H, W), generator=g))
    w = torch.tanh(torch.randn((K, C * p * p), generator=g))
    got = ops.bmu(x.cuda(), w.cuda(), (p, p)).cpu().numpy()
    want = obmu.bmu(x.numpy(), w.numpy(), (p, p))
    assert np.array_equal(got, want)
    # torch's own cdist + argmin (reference models/Codebook.py:86-94; MKL's summation order, not the oracle's):
    # it may differ from the kernel only on rows whose two candidates fp32 cannot tell apart -- each such row is
    # checked in fp64: both candidates within the fp32 noise of the mm-form distance
    # d^2 = |x|^2 + |w|^2 - 2 x.w of the exact minimum, err(d^2) <= 4 sqrt(D + 2) eps32 (|x|^2 + max|w|^2),
    # err(d) = err(d^2) / 2d
    xp = obmu.patchify(x.numpy(), (p, p))
    ref = torch.argmin(torch.cdist(torch.from_numpy(xp), w), -1).numpy()
    differ = np.nonzero(ref != got)[0]
    assert differ.size <= 1e-4 * got.size
    if differ.size:
        i64, gap = obmu.bmu_f64(x.numpy(), w.numpy(), (p, p))
        wd = w.numpy().astype(np.float64)
        w2max = float((wd ** 2).sum(1).max())
        D = xp.shape[1]
        for r in differ:
            xr = xp[r].astype(np.float64)
            dmin = np.sqrt(((xr - wd[i64[r]]) ** 2).sum())
            bound = 4 * np.sqrt(D + 2) * np.finfo(np.float32).eps * ((xr ** 2).sum() + w2max) / max(2 * dmin, 1e-30)
            for cand in (got[r], ref[r]):
                d = np.sqrt(((xr - wd[cand]) ** 2).sum())
                assert d - dmin <= bound, (r, cand, d - dmin, bound)
            assert gap[r] <= bound, (r, gap[r], bound)


def test_bmu_random_shapes_bit_exact():
    """Seeded sweep over ragged geometries (rows and K off the 128 tile, odd patch sizes,
    D from 1 to 768, duplicated codebook rows and duplicated patches, K straddling the
    25-row direct-formula switch of torch.cdist): the HIP indices equal the C oracle's
    bit for bit on every case."""
    from qarig import ops
    from oracle import bmu as obmu
    rng = np.random.default_rng(1234)
    g = torch.Generator().manual_seed(77)
    cases = 0
    for _ in range(40):
        p = int(rng.choice([1, 2, 3, 4, 8]))
        C = int(rng.choice([1, 3, 4, 12]))
        H, W = p * int(rng.integers(1, 7)), p * int(rng.integers(1, 7))
        N = int(rng.integers(1, 6))
        K = int(rng.choice([1, 2, 7, 24, 25, 26, 64, 100, 129, 300, 513]))
        x = torch.tanh(torch.randn((N, C, H, W), generator=g))
        w = torch.tanh(torch.randn((K, C * p * p), generator=g))
        if K > 3 and rng.random() < 0.5:
            w[K // 2] = w[0]                      # duplicated unit: the lower index must win
            w[K - 1] = w[1]
        if rng.random() < 0.3:
            x[:, :, -p:, -p:] = x[:, :, :p, :p].clone()   # duplicated patches
        got = ops.bmu(x.cuda(), w.cuda(), (p, p)).cpu().numpy()
        want = obmu.bmu(x.numpy(), w.numpy(), (p, p))
        assert np.array_equal(got, want), (N, C, H, W, p, K)
        cases += 1
    assert cases == 40


def test_bmu_full_size_properties():
    """BASELINE-size run (65,536 rows x K=512): properties that need no oracle pass --
    the chosen unit's exact distance is within fp32 noise of the exact minimum, and
    quantising codebook rows themselves returns their own index."""
    from qarig import ops
    g = torch.Generator().manual_seed(9)
    x = torch.tanh(torch.randn((64, 4, 64, 64), generator=g)).cuda()
    w = torch.tanh(torch.randn((512, 16), generator=g)).cuda()
    idx = ops.bmu(x, w, (2, 2))
    from oracle import ref_models as rm
    xp = rm.patchify(x, (2, 2)).reshape(-1, 16).double()
    d = torch.cdist(xp, w.double())
    chosen = d.gather(1, idx[:, None]).squeeze(1)
    assert float((chosen - d.min(1).values).max()) < 1e-5
    # identity property: patches that ARE codebook rows map to themselves
    img = rm.unpatchify(w[:256].reshape(1, 256, 16), (32, 32), (2, 2))
    assert torch.equal(ops.bmu(img, w, (2, 2)).cpu(), torch.arange(256))


@pytest.mark.parametrize("case", ["trained_p1", "trained_p2", "ties"])
def test_bmu_coarse_pass_on_reference_goldens(case):
    """The coarse-pass form (bf16 MFMA on three-way operand splits + certificate + exact re-scan) on
    the reference-golden cases it applies to (D <= 16, K a multiple of 32): bit-identical to the C
    oracle and to the reference's own indices; `ties` (duplicated patches / units) must come out
    through the exact re-scan with the first-index rule."""
    from qarig import ops
    from oracle import bmu as obmu
    g = load_golden("bmu")[case]
    p = int(g["p"])
    got, cnt = ops.bmu_coarse(g["x"].cuda(), g["w"].cuda(), (p, p))
    want = obmu.bmu(g["x"].numpy(), g["w"].numpy(), (p, p))
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(got.cpu().numpy(), g["idx"].numpy().reshape(-1))
    if case == "ties":
        assert int(cnt.item()) > 0


@pytest.mark.parametrize("N,C,H,W,p,K,kind", [
    (64, 4, 64, 64, 2, 512, "trained"),      # the metric's own launch: 65,536 rows x 512 x 16
    (64, 4, 32, 32, 1, 512, "trained"),      # D = 4
    (16, 2, 32, 32, 2, 1024, "trained"),     # D = 8, the largest resident codebook
    (3, 4, 34, 26, 2, 96, "trained"),        # ragged row count (663 rows: a partial last block)
    (16, 4, 32, 32, 2, 512, "fresh"),        # degenerate fresh-init codebook: (nearly) every row uncertified
    (8, 4, 32, 32, 2, 256, "dups"),          # duplicated units and patches that ARE units (d = 0, clamp)
    (8, 4, 32, 32, 2, 256, "tiny"),          # denormal-range data: pieces that do not add up -> memory path
    (8, 4, 32, 32, 2, 64, "wide"),           # large dynamic range
])
def test_bmu_coarse_pass_bit_exact(N, C, H, W, p, K, kind):
    from qarig import ops
    from oracle import bmu as obmu
    g = torch.Generator().manual_seed(N + 3 * p + K + len(kind))
    D = C * p * p
    x = torch.tanh(torch.randn((N, C, H, W), generator=g))
    w = torch.tanh(torch.randn((K, D), generator=g))
    if kind == "fresh":
        w = (torch.rand((K, D), generator=g) * 2 - 1) / K
    elif kind == "dups":
        w[K // 2] = w[0]
        w[K - 1] = w[1]
        w[7] = w[300 % K]
        img = w[:64].reshape(1, 64, C, p, p)            # patches that are codebook rows
        x[0, :, :8 * p, :8 * p] = img.reshape(8, 8, C, p, p).permute(2, 0, 3, 1, 4).reshape(C, 8 * p, 8 * p)
        x[1] = x[0]
    elif kind == "tiny":
        x = x * 1e-39
        w = w * 1e-39
    elif kind == "wide":
        x = x * torch.exp(4 * torch.randn((N, 1, H, W), generator=g))
        w = w * torch.exp(4 * torch.randn((K, 1), generator=g))
    got, cnt = ops.bmu_coarse(x.cuda(), w.cuda(), (p, p))
    want = obmu.bmu(x.numpy(), w.numpy(), (p, p))
    assert np.array_equal(got.cpu().numpy(), want), (kind, int((got.cpu().numpy() != want).sum()))
    # the same through a prepared codebook image (what ops.bmu does for a frozen codebook), twice: the
    # second call takes the cached image
    wc = w.cuda()
    for _ in range(2):
        got_p, cnt_p = ops.bmu_coarse(x.cuda(), wc, (p, p), prepared=True)
        assert np.array_equal(got_p.cpu().numpy(), want), kind
        assert int(cnt_p.item()) == int(cnt.item())
    wc.mul_(-1.0)                                   # the codebook changes: the image must follow
    got_n, _ = ops.bmu_coarse(x.cuda(), wc, (p, p), prepared=True)
    assert np.array_equal(got_n.cpu().numpy(), obmu.bmu(x.numpy(), (-w).numpy(), (p, p))), kind
    rows = want.size
    if kind == "trained":
        assert int(cnt.item()) <= max(4, rows // 200), (int(cnt.item()), rows)   # the certificate carries the load
    if kind in ("tiny",):
        assert int(cnt.item()) == rows
    # the dispatcher takes the same path by itself on large launches and agrees
    assert np.array_equal(ops.bmu(x.cuda(), w.cuda(), (p, p)).cpu().numpy(), want)


def test_bmu_takes_the_prepared_image_of_a_stable_parameter_and_follows_its_updates():
    """ops.bmu on an nn.Parameter codebook (what models/Codebook.py passes): the second search of an unchanged
    codebook builds its prepared image, later ones DMA it; an in-place update through torch (version counter) or
    through an optimiser that owns the flat buffer (step count) drops it.  Indices equal the oracle's throughout."""
    from qarig import ops
    from oracle import bmu as obmu
    g = torch.Generator().manual_seed(77)
    x = torch.tanh(torch.randn((32, 4, 64, 64), generator=g))           # 32,768 rows: the coarse-pass kernel
    w = torch.nn.Parameter(torch.tanh(torch.randn((512, 16), generator=g)).cuda(), requires_grad=False)
    xc = x.cuda()
    ops.bmu_invalidate()
    want = obmu.bmu(x.numpy(), w.detach().cpu().numpy(), (2, 2))
    for call in range(3):
        assert np.array_equal(ops.bmu(xc, w, (2, 2)).cpu().numpy(), want), call
        assert (id(w) in ops._bmu_images) == (call >= 1)                 # built by the second search
    img = ops._bmu_images[id(w)][2]
    with torch.no_grad():
        w.mul_(-0.5)                                                     # through torch: the version counter moves
    want2 = obmu.bmu(x.numpy(), w.detach().cpu().numpy(), (2, 2))
    assert not np.array_equal(want, want2)
    assert np.array_equal(ops.bmu(xc, w, (2, 2)).cpu().numpy(), want2)   # (unprepared: first search of this state)
    assert np.array_equal(ops.bmu(xc, w, (2, 2)).cpu().numpy(), want2)
    assert ops._bmu_images[id(w)][2] is not img

    class Owner:                                                         # FlatAdam's contract: p._qarig_owner.step_count
        step_count = 3
    w._qarig_owner = Owner()
    for _ in range(2):
        assert np.array_equal(ops.bmu(xc, w, (2, 2)).cpu().numpy(), want2)
    w.data.mul_(-1.0)                                                    # behind the version counter's back ...
    w._qarig_owner.step_count = 4                                        # ... as the optimiser's step does
    want3 = obmu.bmu(x.numpy(), w.detach().cpu().numpy(), (2, 2))
    assert np.array_equal(ops.bmu(xc, w, (2, 2)).cpu().numpy(), want3)
    assert np.array_equal(ops.bmu(xc, w, (2, 2)).cpu().numpy(), want3)
    # a plain tensor (a detached alias, unknown provenance) never gets an image
    t = w.detach()
    for _ in range(3):
        assert np.array_equal(ops.bmu(xc, t, (2, 2)).cpu().numpy(), want3)
    assert id(t) not in ops._bmu_images


def test_bmu_coarse_pass_on_constructed_near_ties():
    """Adversarial input for the coarse pass's certificate: the codebook is made of PAIRS w, w + s e_j with s
    swept over 1e-6 ... 3e-3, so every patch's best code has a twin whose squared distance differs by anything
    from far below the certificate's eps to well above 4 eps (eps = 1e-5 (|x|^2 + 2 max|w|^2): rows that must be
    re-scanned, rows just certified).  Indices must equal the C oracle's on every row, whichever side they fall."""
    from qarig import ops
    from oracle import bmu as obmu
    g = torch.Generator().manual_seed(321)
    N, C, H, W, p, K = 32, 4, 32, 32, 2, 256
    D = C * p * p
    base = torch.tanh(torch.randn((K // 2, D), generator=g))
    s = torch.exp(torch.empty(K // 2).uniform_(float(np.log(1e-6)), float(np.log(3e-3)), generator=g))
    s = s * (torch.randint(0, 2, (K // 2,), generator=g) * 2 - 1)
    twin = base.clone()
    twin[torch.arange(K // 2), torch.randint(0, D, (K // 2,), generator=g)] += s
    w = torch.stack((base, twin), 1).reshape(K, D).contiguous()
    x = torch.tanh(torch.randn((N, C, H, W), generator=g))
    # a third of the patches sit close to a code, where the pair's gap is smallest relative to eps
    xp = torch.from_numpy(obmu.patchify(x.numpy(), (p, p)).copy())
    near = torch.arange(0, xp.shape[0], 3)
    xp[near] = base[torch.randint(0, K // 2, (near.numel(),), generator=g)] + 0.05 * torch.randn((near.numel(), D), generator=g)
    from oracle import ref_models as rm
    x = rm.unpatchify(xp.reshape(N, -1, D), (H, W), (p, p)).contiguous()
    got, cnt = ops.bmu_coarse(x.cuda(), w.cuda(), (p, p))
    want = obmu.bmu(x.numpy(), w.numpy(), (p, p))
    assert np.array_equal(got.cpu().numpy(), want), int((got.cpu().numpy() != want).sum())
    # the sweep really straddles the certificate: some rows certified, some re-scanned
    assert 0 < int(cnt.item()) < want.size
    _, gap = obmu.bmu_f64(x.numpy(), w.numpy(), (p, p))
    assert (gap < 1e-7).any() and (gap > 1e-3).any()


def test_cpu_tensor_is_refused():
    from qarig import ops
    with pytest.raises(RuntimeError):
        ops.bmu(torch.zeros(1, 4, 4, 4), torch.zeros(8, 16), (2, 2))


def test_gemm_skinny_splitk_with_epilogue():
    """Decode-shaped GEMMs (M = a few dozen rows): the reduction is split over the chip and
    bias / residual / activation run in the reduce pass; same results as the one-pass path."""
    from qarig import ops
    from oracle import ref_models as rm
    g = torch.Generator().manual_seed(4)
    M, N, K = 16, 512, 2048
    A = torch.randn((M, K), generator=g).cuda()
    W = (torch.randn((N, K), generator=g) * 0.05).cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn((M, N), generator=g).cuda()
    assert ops.auto_splitk(M, N, K) > 1
    y_auto, pre_auto = ops.gemm(A, W, bias=b, residual=R, want_preact=True, act=1)
    y_one, pre_one = ops.gemm(A, W, bias=b, residual=R, want_preact=True, act=1, splitk=1)
    t = A.double().cpu() @ W.double().cpu().t() + b.double().cpu() + R.double().cpu()
    assert rel_err(pre_auto, t) < 2e-6 and rel_err(pre_one, t) < 2e-6
    assert rel_err(y_auto, rm.activation(t, "silu")) < 4e-6
    assert rel_err(y_auto, y_one) < 2e-6


@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, False)])
def test_gemm_interior_dma_path(ak, bk):
    """Interior shapes take the LDS-DMA kernel (swizzled k-contiguous tiles, permuted
    summation order): all epilogues, split-K, accumulate and the bias row-sum hook."""
    from qarig import ops
    from oracle import ref_models as rm
    g = torch.Generator().manual_seed(17)
    M, N, K = 256, 384, 160
    A = torch.randn((M, K) if ak else (K, M), generator=g).cuda()
    B = (torch.randn((N, K) if bk else (K, N), generator=g) * 0.3).cuda()
    ref = _ref_gemm(A, B, ak, bk)
    assert rel_err(ops.gemm(A, B, ak, bk), ref) < GEMM_TOL
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn((M, N), generator=g).cuda()
    Z = torch.randn((M, N), generator=g).cuda()
    y, pre = ops.gemm(A, B, ak, bk, bias=b, residual=R, want_preact=True, act=1, splitk=1)
    t = ref + b.double().cpu() + R.double().cpu()
    assert rel_err(pre, t) < GEMM_TOL and rel_err(y, rm.activation(t, "silu")) < 4e-6
    Zd = Z.double().cpu().requires_grad_(True)
    rm.activation(Zd, "silu").sum().backward()
    assert rel_err(ops.gemm(A, B, ak, bk, gradz=Z, gact=1, splitk=1), ref * Zd.grad) < 4e-6
    # split-K (K = 160 -> 5 splits of 32), accumulate into an existing buffer, row sums of A
    C = torch.randn((M, N), generator=g).cuda()
    want = C.double().cpu() + ref
    rs = torch.randn(M, generator=g).cuda()
    rs_want = rs.double().cpu() + (A.double().cpu().sum(1) if ak else A.double().cpu().sum(0))
    ops.gemm(A, B, ak, bk, splitk=5, out=C, accumulate=True, a_rowsum=rs)
    assert rel_err(C, want) < GEMM_TOL
    assert rel_err(rs, rs_want) < GEMM_TOL
    # determinism
    y2, _ = ops.gemm(A, B, ak, bk, bias=b, residual=R, want_preact=True, act=1, splitk=1)
    assert torch.equal(y, y2)
