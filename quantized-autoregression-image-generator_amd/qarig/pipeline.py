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
