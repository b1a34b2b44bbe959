"""Every kernel-selection option (qarig_set_option) against the parity tests of the kernels it
chooses between: an option only picks among kernels that must give the same results, so the same
assertions must hold under each value.  (Round 2 ran this as tools/switch_matrix.sh outside pytest;
the options are settable in-process now, so the driver's `pytest -m gpu` covers every surviving path.)
Superseded forms were deleted instead of switched: the 2/3-stage GEMM ring, the first-round stagger,
the resident-off BMU path for D <= 16, the four-chain BMU scan."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def option():
    from qarig import _lib
    saved = []

    def set_(name, value):
        saved.append((name, _lib.set_option(name, value)))

    yield set_
    for name, old in reversed(saved):
        _lib.set_option(name, old)


def test_unknown_option_is_refused():
    from qarig import _lib
    with pytest.raises(KeyError, match="unknown option"):
        _lib.set_option("no_such_switch", 1)
    assert _lib.set_option("gemm_dma", 1) == 1      # default, and the call returns the previous value


@pytest.mark.parametrize("name,value", [("gemm_dma", 0), ("gemm_pair", 0), ("gemm_pair", 1), ("gemm_xcd_splits", 0),
                                        ("gemm_tile64", 0), ("gemm_tile64", 1), ("gemm_x3", 1)])
def test_gemm_options(option, name, value):
    import test_gpu_core as core
    import test_gpu_grouped as grouped
    option(name, value)
    for ak, bk in ((True, True), (True, False), (False, False)):
        core.test_gemm_layouts(2048, 512, 2048, ak, bk)
        core.test_gemm_interior_dma_path(ak, bk)
    core.test_gemm_epilogues()
    core.test_gemm_splitk_and_colsum()
    core.test_gemm_xcd_split_mapping()
    core.test_gemm_paired_teams(2048, 2048, 512, True, True, 1)
    core.test_gemm_paired_teams(2048, 512, 2048, False, False, 4)
    grouped.test_grouped_mlp_node_matches_separate_mlps_and_fp64(3, 256, 128, 256, 0, True)


def test_gemm_x3_takes_the_products_and_keeps_the_fp32_tolerances(option):
    """Option gemm_x3: the interior products run as six bf16-MFMA products of exact three-way operand splits
    (csrc/gemm_x3.hip).  Every layout, split and epilogue against fp64 at the fp32 kernels' own tolerance
    (2e-6 sqrt(K / 512)); the results differ from the fp32-MFMA kernel's in the last bits (another summation), so
    the kernel did run; row sums riding on a weight-gradient product; a whole train step's parity follows in
    test_gpu_pipeline_golden under the same option."""
    from conftest import rel_err
    from qarig import _lib, ops
    import test_gpu_pipeline_golden as golden
    g = torch.Generator().manual_seed(12)
    lib = _lib.load()
    assert lib.qarig_gemm_x3_ok(2048, 512, 512, 1) == 1 and lib.qarig_gemm_x3_ok(2048, 512, 520, 1) == 0
    assert lib.qarig_gemm_x3_ok(2000, 512, 512, 1) == 0 and lib.qarig_gemm_x3_ok(512, 2048, 16384, 8) == 1
    # (products of fewer than 192 tiles of 128 x 128 take the 64 x 64-tile form, the others the 128 x 128 one;
    #  the last: too few workgroups, not taken)
    for (M, N, K, ak, bk, sk) in ((2048, 512, 512, True, True, 1), (1024, 2048, 2048, True, False, 1),
                                  (512, 2048, 4096, False, False, 8), (1024, 512, 8192, False, True, 4),
                                  (4096, 2048, 512, True, True, 1), (2048, 4096, 1024, True, False, 1),
                                  (2048, 2048, 2048, False, False, 2), (192, 320, 96, False, False, 1),
                                  (128, 128, 32, True, True, 1)):
        A = torch.randn((M, K) if ak else (K, M), generator=g).cuda()
        B = (torch.randn((N, K) if bk else (K, N), generator=g) * 0.05).cuda()
        bias = torch.randn(N, generator=g).cuda()
        res = torch.randn((M, N), generator=g).cuda()
        want = (A.double() if ak else A.double().t()) @ (B.double().t() if bk else B.double())
        outs = {}
        for x3 in (0, 1):
            option("gemm_x3", x3)
            outs[x3] = ops.gemm(A, B, a_kcontig=ak, b_kcontig=bk, splitk=sk)
            tol = 2e-6 * max(1.0, K / 512) ** 0.5
            assert rel_err(outs[x3], want) < tol, (M, N, K, ak, bk, sk, x3)
            if sk == 1:
                y = ops.gemm(A, B, a_kcontig=ak, b_kcontig=bk, bias=bias, residual=res, act=ops.act_id("silu"))
                assert rel_err(y, torch.nn.functional.silu(want + bias.double() + res.double())) < tol
        t128 = -(-M // 128) * -(-N // 128)
        half = t128 * sk < 192 and (M // 64) * (N // 64) * sk >= 32
        full = M % 128 == 0 and N % 128 == 0 and t128 >= 32
        assert torch.equal(outs[0], outs[1]) == (not (half or full)), (M, N, K)
    # the bias gradient riding on a weight-gradient product (tile-contiguous A)
    option("gemm_x3", 1)
    dT = torch.randn((4096, 512), generator=g).cuda()
    X = torch.randn((4096, 2048), generator=g).cuda()
    rs = torch.zeros(512, device="cuda")
    dW = ops.gemm(dT, X, a_kcontig=False, b_kcontig=False, splitk=8, a_rowsum=rs)
    assert rel_err(dW, dT.double().t() @ X.double()) < 2e-6 * 8 ** 0.5
    assert rel_err(rs, dT.double().sum(0)) < 2e-6 * 8 ** 0.5
    dT2 = torch.randn((2048, 2048), generator=g).cuda()             # ... and on the 128 x 128-tile form (256 tiles)
    X2 = torch.randn((2048, 2048), generator=g).cuda()
    rs2 = torch.zeros(2048, device="cuda")
    dW2 = ops.gemm(dT2, X2, a_kcontig=False, b_kcontig=False, splitk=2, a_rowsum=rs2)
    assert rel_err(dW2, dT2.double().t() @ X2.double()) < 2e-6 * 4 ** 0.5
    assert rel_err(rs2, dT2.double().sum(0)) < 2e-6 * 4 ** 0.5
    # grouped launches (q/k/v MLPs of a 2,048-row shard, their summed input gradient, their weight gradients with row
    # sums): every epilogue against fp64, as test_gpu_grouped runs them at small sizes
    import test_gpu_grouped as grouped
    for args in ((3, 2048, 512, 512, True, True, 1, True), (3, 2048, 2048, 512, True, False, 1, False),
                 (3, 512, 2048, 2048, False, False, 4, False), (14, 128, 128, 192, True, False, 3, False)):
        grouped.test_grouped_gemm_every_epilogue_vs_fp64(*args)
    # one full train step of the golden pipeline under the option
    golden.test_train_step_matches_reference_step("base", False)
    golden.test_train_step_matches_reference_step("encdec", True)


@pytest.mark.parametrize("name,value", [("bmu_cs", 1), ("bmu_cs", 2), ("bmu_cs", 4), ("bmu_groups", 0),
                                        ("bmu_groups", 1), ("bmu_coarse", 0), ("bmu_coarse", 1)])
def test_bmu_options(option, name, value):
    import test_gpu_core as core
    option(name, value)
    for case in ("trained_p1", "trained_p2", "ragged", "fresh_p4", "ties"):
        core.test_bmu_vs_oracle_and_golden(case)
    core.test_bmu_seeded_vs_oracle(64, 4, 32, 32, 2, 512)
    core.test_bmu_seeded_vs_oracle(16, 4, 64, 64, 1, 8192)
    core.test_bmu_random_shapes_bit_exact()


@pytest.mark.parametrize("name,value", [("attn_qw", 1), ("attn_qw", 4), ("attn_bw", 1)])
def test_attention_options(option, name, value):
    from conftest import rel_err
    from qarig import ops
    option(name, value)
    g = torch.Generator().manual_seed(3)
    for (N, Sq, Sk, H, d, causal) in ((2, 256, 256, 64, 8, True), (1, 130, 77, 6, 16, False), (2, 64, 64, 4, 64, True)):
        q, k, v, do = (torch.randn(s, generator=g) for s in ((N, Sq, H * d), (N, Sk, H * d), (N, Sk, H * d),
                                                              (N, Sq, H * d)))
        qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
        s_ = torch.einsum("nqhd,nkhd->nhqk", qd.view(N, Sq, H, d), kd.view(N, Sk, H, d)) / d ** 0.5
        if causal:
            s_ = s_.masked_fill(torch.ones(Sq, Sk).triu(1).bool(), float("-inf"))
        od = torch.einsum("nhqk,nkhd->nqhd", torch.softmax(s_, -1), vd.view(N, Sk, H, d)).reshape(N, Sq, H * d)
        od.backward(do.double())
        o, lse = ops.attention_fwd(q.cuda(), k.cuda(), v.cuda(), H, causal)
        dq, dk, dv = ops.attention_bwd(q.cuda(), k.cuda(), v.cuda(), o, do.cuda(), lse, H, causal)
        assert rel_err(o, od) < 2e-6
        for got, want in ((dq, qd.grad), (dk, kd.grad), (dv, vd.grad)):
            assert rel_err(got, want) < 5e-6


@pytest.mark.parametrize("name,value", [("lp_big", 0), ("lp_big", 1), ("lp_mfma16", 0)])
def test_reduced_precision_options(option, name, value):
    import test_gpu_bf16 as bf
    from qarig import ops
    option(name, value)
    for layout in (0, 1, 2):
        bf.test_lp_fragment_maps_on_exact_integer_data(layout, 4096, 4096, 192)
        bf.test_lp_fragment_maps_on_exact_integer_data(layout, 256, 384, 192)
    old, ops.PRECISION = ops.PRECISION, "bf16"
    try:
        for ak, bk in ((True, True), (True, False), (False, False)):
            bf.test_bf16_gemm_is_exact_on_rounded_operands(ops, 256, 384, 512, 1, ak, bk)
        bf.test_bf16_gemm_epilogues_and_accumulate(ops)
    finally:
        ops.PRECISION = old
    bf.test_lp_big_tile_kernel_epilogues()


@pytest.mark.parametrize("name,value", [("convt_pair", 0), ("conv_ring", 0)])
def test_conv_options(option, name, value):
    """conv_ring = 0 sends every convolution to the gather kernels: the ring kernels' parity tests then
    pin the gather kernels on the same shapes (the cross-check of round 2's switch matrix)."""
    import inspect

    import test_gpu_conv as conv
    option(name, value)
    ran = 0
    for fname, fn in sorted(vars(conv).items()):
        if not fname.startswith("test_") or not callable(fn):
            continue
        marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
        if marks or inspect.signature(fn).parameters:
            continue                      # parametrised / fixture-taking tests keep to their own runs
        fn()
        ran += 1
    assert ran >= 5, ran
    # the ring kernels' own shapes, now on the other kernel family
    conv.test_conv3x3_ring_kernel_fwd_bwd_vs_fp64(2, 32, 8, 8, 128, 1)
    conv.test_conv3x3_ring_kernel_fwd_bwd_vs_fp64(1, 48, 16, 24, 256, 2)
    conv.test_conv_transpose2d_ring_kernel_vs_fp64(2, 32, 8, 8, 128)
    conv.test_conv2d_stride2_input_gradient_ring_vs_fp64(2, 128, 16, 16, 32)
    conv.test_conv2d_stride2_forward_ring_vs_fp64(2, 32, 16, 16, 128, 1)
    conv.test_conv_transpose2d_input_gradient_ring_vs_fp64(8, 128, 4, 4, 16)
    conv.test_conv2d_readme_channels_fwd_bwd_vs_fp64(256, 512, 2)
    conv.test_conv_transpose2d_readme_channels_fwd_bwd_vs_fp64(512, 256)
