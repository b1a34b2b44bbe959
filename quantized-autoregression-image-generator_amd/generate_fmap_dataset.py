#!/usr/bin/env python3
"""Encode an image dataset into per-image latent .npy files + a TinyDB-format index on
MI355X: same command line and output layout (<out>/<folder>/<index> files,
<out>/all_dataset.json) as the reference's generate_fmap_dataset.py.

Under torchrun every rank encodes a contiguous share of the files (independent units, no collective on the
data path) under the file numbers a single process gives them; rank 0 writes the index of all of them."""
import argparse
import os
import pathlib

import numpy as np
import torch

from models.FC_Encoder import FC_Encoder
from qarig import cli_common as cc
from qarig import parallel
from dataset_loader._tinydb_json import write_all
from dataset_loader.image_dataset import ImageDataset
from utils.model_utils import load_model


def save_feature_maps(model, dataloader, out_dir, device, num_files_folder=1_000, first_index=0):
    """first_index: the dataset position of the loader's first item (a rank's share starts there): file numbers
    and folders are those of the whole dataset in order (reference generate_fmap_dataset.py:40-72)."""
    file_index, all_data = first_index, []
    print("#" * 100)
    print("Saving Feature Maps to disk...")
    for index, (image, image_paths) in enumerate(dataloader):
        with torch.no_grad():
            latent = model(image.to(device)).cpu()
        for fmap, image_path in zip(latent, image_paths):
            folder_name = file_index // num_files_folder
            folder = os.path.join(out_dir, str(folder_name))
            os.makedirs(folder, exist_ok=True)
            path = os.path.join(folder, f"{file_index}")
            with open(path, "wb") as f:
                np.save(f, fmap.numpy(), allow_pickle=False, fix_imports=False)
            file_index += 1
            all_data.append({"fmap_path": path, "image_path": image_path})
        print(f"{(index + 1):,} / {len(dataloader):,}")
    print("Finished saving feature maps.")
    if parallel.world_size() > 1:                       # the records of every rank's share, in dataset order
        import torch.distributed as dist
        shares = [None] * parallel.world_size()
        dist.all_gather_object(shares, all_data)
        all_data = [rec for share in shares for rec in share]
    if parallel.rank() == 0:
        write_all(os.path.join(out_dir, "all_dataset.json"), all_data)
        print("Finished saving json file.")
    print("#" * 100)


def main():
    p = argparse.ArgumentParser(description="Generate Feature Maps Dataset.")
    p.add_argument("--device", choices=["cpu", "cuda"], type=str, default="cpu")
    p.add_argument("--batch-size", type=int, default=8)
    p.add_argument("--num-files-folder", type=int, default=1_000)
    p.add_argument("--dataset-path", required=True, type=pathlib.Path)
    p.add_argument("--model-path", required=True, type=pathlib.Path)
    p.add_argument("--out-dir", required=True, type=pathlib.Path)
    args = vars(p.parse_args())
    device, _, _ = cc.require_gpu(args["device"])
    os.makedirs(args["out_dir"], exist_ok=True)
    ok, d = load_model(args["model_path"])
    if not ok:
        raise Exception("An error occured while loading Encoder model checkpoint!")
    # NOTE (reference behaviour kept): the encoder's final activation switch is read from
    # "use_final_dec_activation" (generate_fmap_dataset.py:136), not "use_final_enc_activation".
    encoder = FC_Encoder(num_layers=d["num_layers"], image_channel=d["image_channel"],
                         min_channel=d["min_channel"], max_channel=d["max_channel"],
                         latent_channel=d["latent_channel"],
                         hidden_activation_type=d["hidden_activation_type"],
                         use_final_activation=d["use_final_dec_activation"],
                         final_activation_type=d["encoder_activation_type"])
    encoder.custom_load_state_dict(d["model"], ignore_msgs=True)
    encoder = encoder.to(device).eval()
    dataset = ImageDataset(dataset_path=args["dataset_path"], return_filepaths=True)
    lo, hi = parallel.shard_range(len(dataset))
    if parallel.world_size() > 1:
        dataset = torch.utils.data.Subset(dataset, range(lo, hi))
    loader = torch.utils.data.DataLoader(dataset, batch_size=args["batch_size"], num_workers=4,
                                         shuffle=False)
    save_feature_maps(encoder, loader, str(args["out_dir"]), device, args["num_files_folder"], first_index=lo)


if __name__ == "__main__":
    main()
