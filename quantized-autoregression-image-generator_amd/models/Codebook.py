"""Drop-in `models.Codebook.Codebook` (reference models/Codebook.py:18-164): SOM-style
codebook over latent patches.  BMU search, gather and the Gaussian neighbourhood
quantisation run on HIP (fused patchify; no (rows x K) distance matrix)."""
import math

import torch
import torch.nn as nn

from qarig import ops

from ._loading import load_matching


def _som_dense():
    """QARIG_SOM_DENSE=1: the reference's literal form, an (rows x K) weight matrix through the GEMM."""
    import os
    return os.environ.get("QARIG_SOM_DENSE", "0") == "1"


class _SomQuantize(torch.autograd.Function):
    """quant = g @ W with g the (constant) Gaussian index-neighbourhood weights; gradient reaches
    the codebook only (reference Codebook.py:112-130).  g[r][j] depends on r only through bmu[r]:
    quant = band(W)[bmu] and dW = band(per-code sums of dq), band = the Gaussian along the code axis
    over the indices whose weight is >= 2^-40 (csrc/codebook.hip som_band_kernel) -- no (rows x K) matrix."""

    @staticmethod
    def forward(ctx, weight, bmu, two_var):
        ctx.two_var = two_var
        if _som_dense():
            g = ops.som_weights(bmu, weight.shape[0], two_var)
            ctx.save_for_backward(g)
            ctx.dense = True
            return ops.gemm(g, weight, a_kcontig=True, b_kcontig=False)
        ctx.dense = False
        ctx.K = weight.shape[0]
        ctx.save_for_backward(bmu)
        return ops.gather_rows(bmu, ops.som_band(weight.detach(), two_var))

    @staticmethod
    def backward(ctx, dq):
        dq = dq.contiguous()
        if ctx.dense:
            (g,) = ctx.saved_tensors
            R, K = g.shape
            return ops.gemm(g, dq, a_kcontig=False, b_kcontig=False,
                            splitk=ops.pick_splitk(K, dq.shape[1], R)), None, None
        (bmu,) = ctx.saved_tensors
        return ops.som_band(ops.embedding_bwd(bmu, dq, ctx.K), ctx.two_var), None, None


class _HardQuantize(torch.autograd.Function):
    """codebook[bmu] (nn.Embedding semantics, reference Codebook.py:132)."""

    @staticmethod
    def forward(ctx, weight, bmu):
        ctx.save_for_backward(bmu)
        ctx.K = weight.shape[0]
        return ops.gather_rows(bmu, weight)

    @staticmethod
    def backward(ctx, dq):
        (bmu,) = ctx.saved_tensors
        return ops.embedding_bwd(bmu, dq.contiguous(), ctx.K), None


class _Unpatchify(torch.autograd.Function):
    @staticmethod
    def forward(ctx, patches, image_dim, patch_dim):
        ctx.patch_dim = patch_dim
        return ops.unpatchify(patches, image_dim, patch_dim)

    @staticmethod
    def backward(ctx, dimg):
        return ops.patchify(dimg.contiguous(), ctx.patch_dim), None, None


class Codebook(nn.Module):
    def __init__(self, patch_dim=(2, 2), image_dim=(32, 32), image_channel=4, num_embeddings=512,
                 init_neighbour_range=256):
        super().__init__()
        # reference Codebook.py:27-28 (a check that can never fire; kept verbatim in effect)
        if init_neighbour_range > num_embeddings and init_neighbour_range < 1:
            raise Exception("Invalid value for init_neighbour_range.")
        self.neighbourhood_range = init_neighbour_range
        self.patch_dim = patch_dim
        self.image_dim = image_dim
        patch_H, patch_W = self.patch_dim
        self.embedding_dim = image_channel * patch_H * patch_W
        self.num_embeddings = num_embeddings
        self.codebook = nn.Embedding(self.num_embeddings, self.embedding_dim)
        self.codebook.weight.data.uniform_(-1 / self.num_embeddings, 1 / self.num_embeddings)

    def custom_load_state_dict(self, state_dict, ignore_msgs=False):
        load_matching(self, state_dict, ignore_msgs=ignore_msgs)

    # reference Codebook.py:68-74 (`steps` is validated but otherwise unused)
    def decrease_neighbourhood(self, steps=1):
        if steps < 1:
            raise Exception("Invalid value for steps, should be > 1.")
        self.neighbourhood_range = 1.0 if self.neighbourhood_range <= 1 \
            else self.neighbourhood_range - 1

    # reference Codebook.py:77-99
    def get_patches_bmu(self, x, reshape=False):
        # (the Parameter itself, not a detached alias: ops.bmu follows its version counter and its optimiser's step
        #  count to know when a prepared image of the codebook is still the codebook)
        idx = ops.bmu(x, self.codebook.weight, self.patch_dim)
        if reshape:
            idx = idx.reshape(x.shape[0], -1)
        return idx

    # reference Codebook.py:102-135
    def get_quantized_patches(self, x, use_gaussian=True):
        bmu = self.get_patches_bmu(x)
        N = x.shape[0]
        if use_gaussian:
            variance = -(self.neighbourhood_range / (2 * math.log(0.1)))
            q = _SomQuantize.apply(self.codebook.weight, bmu, 2 * variance)
        else:
            q = _HardQuantize.apply(self.codebook.weight, bmu)
        return q.view(N, -1, self.embedding_dim)

    # reference Codebook.py:138-154
    def get_quantized_image(self, indices, unpatchify_input=True):
        N, Seq = indices.shape
        if unpatchify_input:
            return ops.codebook_gather_image(indices, self.codebook.weight.detach(), self.image_dim,
                                             self.patch_dim)
        return ops.gather_rows(indices, self.codebook.weight.detach()).view(
            N, Seq, self.embedding_dim)

    # reference Codebook.py:156-164
    def forward(self, x, use_gaussian=True):
        q = self.get_quantized_patches(x, use_gaussian=use_gaussian)
        return _Unpatchify.apply(q, self.image_dim, self.patch_dim)
