"""GPU parity of the product Codebook class (BMU, Gaussian-neighbourhood quantise,
gather, unpatchify) against the reference's goldens."""
import pytest
import torch

from conftest import grad_err, load_golden, rel_err

pytestmark = pytest.mark.gpu


def _cb(g, rng=4):
    from models.Codebook import Codebook
    cb = Codebook(patch_dim=(2, 2), image_dim=(8, 8), image_channel=4, num_embeddings=48,
                  init_neighbour_range=rng)
    cb.custom_load_state_dict({"codebook.weight": g["w"]})
    return cb.cuda()


def test_codebook_class_vs_reference_golden():
    g = load_golden("codebook")
    cb = _cb(g)
    x = g["x"].cuda()
    assert torch.equal(cb.get_patches_bmu(x, reshape=True).cpu(), g["bmu"])
    assert cb.get_patches_bmu(x).shape == (3 * 16,)
    q = cb(x, use_gaussian=True)
    assert rel_err(q, g["fwd_gauss"]) < 2e-6
    loss = ((q - x) ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss_gauss"])) < 1e-6
    assert grad_err(cb.codebook.weight.grad, g["w_grad_gauss"]) < 1e-5
    assert rel_err(cb.get_quantized_patches(x, use_gaussian=True), g["patches_gauss"]) < 2e-6
    cb.codebook.weight.grad = None
    q2 = cb(x, use_gaussian=False)
    assert torch.equal(q2.cpu(), g["fwd_hard"])
    (q2 ** 2).mean().backward()
    assert grad_err(cb.codebook.weight.grad, g["w_grad_hard"]) < 1e-6
    idx = g["idx"].cuda()
    assert torch.equal(cb.get_quantized_image(idx).cpu(), g["quant_image"])
    assert torch.equal(cb.get_quantized_image(idx, unpatchify_input=False).cpu(), g["quant_patches"])
    seq = []
    for _ in range(6):
        cb.decrease_neighbourhood()
        seq.append(cb.neighbourhood_range)
    assert seq == g["neighbourhood_seq"].tolist()
    cb.neighbourhood_range = 2
    assert rel_err(cb(x, use_gaussian=True), g["fwd_gauss_r2"]) < 2e-6
    with pytest.raises(Exception):
        cb.decrease_neighbourhood(0)


def test_patchify_roundtrip_and_goldens():
    from models.layers import patchify, unpatchify
    g = load_golden("layers")
    for p in (1, 2, 4):
        pt = patchify(g["x"].cuda(), (p, p))
        assert torch.equal(pt.cpu(), g[f"patch_p{p}"])
        assert torch.equal(unpatchify(pt, (4, 4), (p, p)).cpu(), g["x"])
    assert torch.equal(patchify(g["xr"].cuda(), (2, 3)).cpu(), g["patch_rect"])
    # idempotence at full size: unpatchify(patchify(z)) == z for every README patch size
    z = torch.randn(64, 4, 32, 32, device="cuda")
    for p in (1, 2, 4, 8, 32):
        assert torch.equal(unpatchify(patchify(z, (p, p)), (32, 32), (p, p)), z)


def test_gather_out_of_range_is_reported():
    from qarig import ops
    g = load_golden("codebook")
    cb = _cb(g)
    bad = g["idx"].clone()
    bad[0, 0] = 48
    cb.get_quantized_image(bad.cuda())
    with pytest.raises(IndexError):
        ops.check_index_flag(torch.device("cuda", 0), "gather")


@pytest.mark.parametrize("K,D,R,rng", [(512, 16, 8192, 256), (512, 16, 4099, 1.0), (512, 4, 2048, 37),
                                       (8192, 4, 32768, 4096), (48, 16, 100, 4), (7, 3, 5, 256)])
def test_som_quantise_band_form_vs_fp64_and_dense(K, D, R, rng, monkeypatch):
    """The Gaussian neighbourhood as a band over the code axis (no rows x K matrix): forward rows and
    the codebook gradient against an fp64 evaluation of the reference expression
    (Codebook.py:112-130) over ALL K codes, and against the literal matrix form (QARIG_SOM_DENSE=1).
    Dropped weights are < 2^-40 of the centre weight: the bound is the fp32 summation's own."""
    import math

    from models.Codebook import _SomQuantize
    from qarig import ops
    g = torch.Generator().manual_seed(K + D + R)
    w = ((torch.rand((K, D), generator=g) * 2 - 1) / K).cuda()
    bmu = torch.randint(0, K, (R,), generator=g).cuda()
    bmu[0], bmu[-1] = 0, K - 1          # both clipped ends of the band
    dq = torch.randn((R, D), generator=g).cuda()
    two_var = 2 * -(rng / (2 * math.log(0.1)))
    assert ops.som_reach(two_var) ** 2 / two_var >= 40 * math.log(2)

    def run():
        wl = w.clone().requires_grad_(True)
        q = _SomQuantize.apply(wl, bmu, two_var)
        (q * dq).sum().backward()
        return q.detach(), wl.grad

    q, dw = run()
    j = torch.arange(K, dtype=torch.float64)
    # the reference's weights are fp32 values (int64 squared / python float -> fp32, exp in fp32)
    d2 = ((j[None, :] - bmu.cpu().double()[:, None]) ** 2).float()
    gw = torch.exp(-(d2 / two_var)).double()
    assert rel_err(q, gw @ w.cpu().double()) < 2e-6
    assert grad_err(dw, gw.t() @ dq.cpu().double()) < 1e-5
    if R * K <= (1 << 24):
        monkeypatch.setenv("QARIG_SOM_DENSE", "1")
        qd, dwd = run()
        assert rel_err(q, qd) < 2e-6 and grad_err(dw, dwd) < 1e-5


@pytest.mark.parametrize("V,D,M", [(48, 16, 100), (512, 16, 65536), (512, 64, 1000), (4099, 4, 32768 + 17),
                                   (8192, 4, 63), (1024, 3, 64 * 32 + 1)])
def test_embedding_bwd_narrow_tables(V, D, M):
    """Per-code sums of gradient rows (nn.Embedding backward, Codebook.py:132; the band form's
    gradient) on the narrow-table kernel: vs an fp64 index_add, and deterministic."""
    from qarig import ops
    g = torch.Generator().manual_seed(V + D + M)
    ids = torch.randint(0, V, (M,), generator=g)
    ids[0], ids[-1] = V - 1, 0
    dy = torch.randn((M, D), generator=g)
    want = torch.zeros((V, D), dtype=torch.float64).index_add_(0, ids, dy.double())
    a = ops.embedding_bwd(ids.cuda(), dy.cuda(), V)
    assert grad_err(a, want) < 2e-6
    assert torch.equal(a, ops.embedding_bwd(ids.cuda(), dy.cuda(), V))
    assert torch.equal(a[want.abs().sum(1) == 0].cpu(), torch.zeros_like(a[want.abs().sum(1) == 0].cpu()))
