#!/usr/bin/env python3
"""Drop codebook units used fewer than --prune-threshold times over a latent dataset, on
MI355X: same command line, log and `pruned_codebook.pt` dict as the reference's
prune_codebook.py; the usage count is a device histogram instead of a Python dict loop."""
import argparse
import logging
import pathlib

import torch

from models.Codebook import Codebook
from qarig import cli_common as cc
from qarig import ops, parallel
from dataset_loader.feature_map_dataset import FeatureMapDataset
from utils.model_utils import save_model


def main():
    project_name = "Prune Codebook"
    p = argparse.ArgumentParser(description=f"Train {project_name}.")
    p.add_argument("--device", choices=["cpu", "cuda"], type=str, default="cpu")
    p.add_argument("--dataset-path", required=True, type=pathlib.Path)
    p.add_argument("--codebook-path", required=True, type=pathlib.Path)
    p.add_argument("--batch-size", type=int, default=8)
    p.add_argument("--prune-threshold", type=int, default=10)
    p.add_argument("--out-dir", required=True, type=pathlib.Path)
    args = vars(p.parse_args())
    device, world, rank = cc.require_gpu(args["device"])
    out_dir = args["out_dir"]
    cc.setup_logging(out_dir, project_name, rank)
    info = logging.info
    codebook, d = cc.load_codebook(args["codebook_path"], device)
    codebook.eval()
    info(f"{project_name}")
    info(f"Output Dir: {out_dir}")
    info("#" * 100)
    info("Codebook Parameters.")
    info(f"Image dim: {d['image_dim']}")
    info(f"Image channel: {d['image_C']:,}")
    info(f"Patch size: {d['patch_dim']}")
    info(f"Num Embeddings: {d['num_embeddings']:,}")
    info(f"Neighbourhood range: {d['neighbourhood_range']:,}")
    info("#" * 100)
    dataset = FeatureMapDataset(dataset_path=args["dataset_path"], load_image=False,
                                return_filepaths=False)
    if world > 1:       # every rank counts a contiguous share of the files; ONE exchange: the K-bin histogram
        dataset = torch.utils.data.Subset(dataset, range(*parallel.shard_range(len(dataset))))
    loader = torch.utils.data.DataLoader(dataset, batch_size=args["batch_size"], num_workers=4,
                                         shuffle=True)
    counts = torch.zeros(d["num_embeddings"], dtype=torch.int64, device=device)
    for fm in loader:
        ops.index_histogram(codebook.get_patches_bmu(fm.to(device)), counts)
    ops.check_index_flag(device, "BMU histogram")
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        if rank != 0:
            return                      # rank 0 prints the histogram and writes the pruned codebook
    counts = counts.cpu().tolist()
    good = []
    for i, c in enumerate(counts):
        print(f"{i}: {c:,}")
        if c >= args["prune_threshold"]:
            good.append(i)
    info(f"Saved embeddings: {len(good)}")
    new_cb = Codebook(patch_dim=d["patch_dim"], image_dim=d["image_dim"], image_channel=d["image_C"],
                      num_embeddings=len(good), init_neighbour_range=d["neighbourhood_range"])
    with torch.no_grad():
        new_cb.codebook.weight.copy_(codebook.codebook.weight.detach().cpu()[good])
    out = {"patch_dim": d["patch_dim"], "image_dim": d["image_dim"], "image_C": d["image_C"],
           "num_embeddings": len(good), "neighbourhood_range": d["neighbourhood_range"],
           "global_steps": d["global_steps"], "checkpoint": new_cb.state_dict()}
    ok = save_model(model_dict=out, dest_path=out_dir, file_name="pruned_codebook.pt", logging=info)
    info("Successfully saved codebook." if ok else "Error occured saving codebook.")


if __name__ == "__main__":
    main()
