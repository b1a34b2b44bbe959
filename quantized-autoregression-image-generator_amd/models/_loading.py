"""Forgiving state-dict loader shared by the model classes: unknown names and shape
mismatches are skipped with a printed message and never raise (reference
models/Transformer.py:104-120, Codebook.py:48-66, FC_Encoder.py:62-83,
FC_Decoder.py:68-91, Autoencoder.py:46-61)."""
import torch


def load_matching(module, state_dict, rename=None, must_contain=None, ignore_msgs=False):
    own = module.state_dict()
    for name, param in state_dict.items():
        if rename is not None:
            name = name.replace(rename[0], rename[1])
        if must_contain is not None and must_contain not in name:
            if not ignore_msgs:
                print(f"Skipping: {name}")
            continue
        if name not in own:
            if not ignore_msgs:
                print(f"No Layer found: {name}, skipping")
            continue
        if own[name].shape != param.data.shape:
            if not ignore_msgs:
                print(f"Skipped: {name}")
            continue
        if isinstance(param, torch.nn.parameter.Parameter):
            param = param.data
        own[name].copy_(param)
