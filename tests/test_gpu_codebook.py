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
