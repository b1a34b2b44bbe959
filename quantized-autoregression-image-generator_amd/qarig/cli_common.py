"""Shared plumbing of the re-hosted command lines (same flags, JSON keys, log lines,
checkpoint dicts and output tree as the reference's scripts)."""
import json
import logging
import os

import torch

from . import parallel


def require_gpu(device):
    if device != "cuda":
        raise SystemExit("this build runs on MI355X only: pass --device cuda "
                         "(PyTorch-ROCm names the HIP device 'cuda'); there is no CPU path")
    if not torch.cuda.is_available():
        raise SystemExit("--device cuda requested but no GPU is visible")
    world, rank, local = parallel.init()
    local %= torch.cuda.device_count()      # (more ranks than GPUs: gloo tests on a one-GPU box share the device)
    torch.cuda.set_device(local)
    return torch.device("cuda", local), world, rank


def read_config(path):
    with open(path, "r") as f:
        return json.loads(f.read())


def setup_logging(out_dir, project_name, rank=0):
    os.makedirs(out_dir, exist_ok=True)
    handlers = [logging.StreamHandler()]
    if rank == 0:
        handlers.insert(0, logging.FileHandler(os.path.join(out_dir, f"{project_name}.log")))
    logging.basicConfig(format="%(asctime)s %(message)s", encoding="utf-8", handlers=handlers,
                        level=logging.DEBUG if rank == 0 else logging.WARNING, force=True)


def load_decoder(path, device):
    """FC_Decoder rebuilt from an autoencoder/decoder checkpoint dict
    (reference train_quantized_transformer.py:186-208)."""
    from models.FC_Decoder import FC_Decoder
    from utils.model_utils import load_model
    ok, d = load_model(path)
    if not ok:
        raise Exception("An error occured while loading decoder model checkpoint!")
    dec = FC_Decoder(num_layers=d["num_layers"], image_channel=d["image_channel"],
                     min_channel=d["min_channel"], max_channel=d["max_channel"],
                     latent_channel=d["latent_channel"],
                     hidden_activation_type=d["hidden_activation_type"],
                     use_final_activation=d["use_final_dec_activation"],
                     final_activation_type=d["decoder_activation_type"]).to(device)
    dec.custom_load_state_dict(d["model"])
    return dec, d


def load_codebook(path, device, what="codebook"):
    """Codebook rebuilt from its checkpoint dict (reference :211-255)."""
    from models.Codebook import Codebook
    from utils.model_utils import load_model
    ok, d = load_model(path)
    if not ok:
        raise Exception(f"An error occured while loading {what} checkpoint!")
    cb = Codebook(patch_dim=d["patch_dim"], image_dim=d["image_dim"], image_channel=d["image_C"],
                  num_embeddings=d["num_embeddings"],
                  init_neighbour_range=d["neighbourhood_range"]).to(device)
    cb.custom_load_state_dict(d["checkpoint"])
    return cb, d


class ShardedLoader:
    """Batches of a Dataset with a shuffle drawn ONCE per epoch from the global CPU
    generator and sliced per rank (same permutation on every rank), so that a DP run on
    identical inputs reproduces the single-process order.  At world 1 it is
    torch's own DataLoader(shuffle=True), i.e. the reference's loader."""

    def __init__(self, dataset, batch_size, num_workers=4, shuffle=True, drop_last=False):
        self.dataset, self.batch_size = dataset, batch_size
        self.world, self.rank = parallel.world_size(), parallel.rank()
        self.num_workers, self.shuffle = num_workers, shuffle
        self._plain = None
        if self.world == 1:
            self._plain = torch.utils.data.DataLoader(dataset, batch_size=batch_size,
                                                      num_workers=num_workers, shuffle=shuffle,
                                                      drop_last=drop_last)

    def __len__(self):
        if self._plain is not None:
            return len(self._plain)
        return len(self.dataset) // (self.batch_size * self.world)

    def __iter__(self):
        if self._plain is not None:
            yield from self._plain
            return
        n = len(self.dataset)
        order = torch.randperm(n) if self.shuffle else torch.arange(n)
        order = parallel.broadcast_host_tensor(order)   # rank 0's draw, on every rank
        per = self.batch_size * self.world
        for b in range(n // per):
            idx = order[b * per + self.rank * self.batch_size:
                        b * per + (self.rank + 1) * self.batch_size].tolist()
            yield torch.utils.data.default_collate([self.dataset[i] for i in idx])


def halve_lr(optim):
    for g in optim.param_groups:
        g["lr"] = g["lr"] * 0.5
