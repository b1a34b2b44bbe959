#!/usr/bin/env python3
"""Train a SOM-style codebook on latent patches on MI355X: same command line, JSON keys,
checkpoint dict and log lines as the reference's train_codebook.py."""
import argparse
import logging
import pathlib

import torch

from models.Codebook import Codebook
from qarig import cli_common as cc
from qarig import functional as QF
from qarig import parallel
from qarig.optim import FlatAdam
from utils.image_utils import save_images
from utils.model_utils import load_model, save_model
from dataset_loader.feature_map_dataset import FeatureMapDataset


def parse_args():
    p = argparse.ArgumentParser(description="Train Codebook models.")
    p.add_argument("--device", choices=["cpu", "cuda"], type=str, default="cpu")
    p.add_argument("--dataset-path", required=True, type=pathlib.Path)
    p.add_argument("--decoder-path", required=True, type=pathlib.Path)
    p.add_argument("--codebook-path", default=None, required=False, type=pathlib.Path)
    p.add_argument("--batch-size", type=int, default=64)
    p.add_argument("--checkpoint-step", type=int, default=1_000)
    p.add_argument("--lr-step", type=int, default=50_000)
    p.add_argument("--max-epoch", type=int, default=1_000)
    p.add_argument("--config-path", required=True, type=pathlib.Path)
    p.add_argument("--out-dir", required=True, type=pathlib.Path)
    p.add_argument("--max-steps", type=int, default=None, help="(additive) stop after N steps.")
    return vars(p.parse_args())


def main():
    project_name = "Codebook"
    args = parse_args()
    cfg = cc.read_config(args["config_path"])
    device, world, rank = cc.require_gpu(args["device"])
    out_dir = args["out_dir"]
    cc.setup_logging(out_dir, project_name, rank)
    info = logging.info
    decoder_model, _ = cc.load_decoder(args["decoder_path"], device)
    decoder_model.eval()
    model_lr, neighbourhood_step = cfg["model_lr"], cfg["neighbourhood_step"]
    global_steps = 0
    if args["codebook_path"] is not None:
        ok, d = load_model(args["codebook_path"])
        if not ok:
            raise Exception("An error occured while loading codebook checkpoint!")
        patch_dim, image_dim, image_C = d["patch_dim"], d["image_dim"], d["image_C"]
        codebook = Codebook(patch_dim=patch_dim, image_dim=image_dim, image_channel=image_C,
                            num_embeddings=d["num_embeddings"],
                            init_neighbour_range=d["neighbourhood_range"])
        codebook.custom_load_state_dict(d["checkpoint"])
        global_steps = d["global_steps"]
    else:
        image_dim = (cfg["image_H"], cfg["image_W"])
        image_C = cfg["image_C"]
        patch_dim = (cfg["patch_H"], cfg["patch_W"])
        codebook = Codebook(patch_dim=patch_dim, image_dim=image_dim, image_channel=image_C,
                            num_embeddings=cfg["num_embeddings"],
                            init_neighbour_range=cfg["num_embeddings"] // 2)
    codebook = codebook.to(device)
    optim = FlatAdam(codebook.parameters(), lr=model_lr, betas=(0.5, 0.999))
    parallel.broadcast_params(optim)
    dataset = FeatureMapDataset(dataset_path=args["dataset_path"], load_image=False,
                                return_filepaths=False)
    loader = cc.ShardedLoader(dataset, args["batch_size"], num_workers=4, shuffle=True)
    info(f"{project_name}")
    info(f"Output Dir: {out_dir}")
    info(f"Patch dim: {patch_dim} | Image dim: {image_dim} | Num Embeddings: {codebook.num_embeddings:,}")
    info("#" * 100)

    done = False
    for epoch in range(0, args["max_epoch"]):
        iteration_count, total = 0, 0.0
        for index, fm in enumerate(loader):
            iteration_count += 1
            fm = fm.to(device)
            codebook.train()
            optim.zero_grad()
            quant = codebook(fm, use_gaussian=True)
            loss = QF.mse_loss(quant, fm)
            loss.backward()
            if world > 1:
                parallel.allreduce_flat(optim.flat_grad)
            optim.step(grad_scale=1.0 / world)
            lv = loss.item()
            if lv != lv:
                raise Exception("NaN encountered during training")
            total += lv
            if global_steps % args["lr_step"] == 0 and global_steps > 0:
                cc.halve_lr(optim)
            if global_steps % args["checkpoint_step"] == 0 and rank == 0:
                with torch.no_grad():
                    save_images(decoder_model(fm), f"image_plot_{global_steps}", out_dir, logging=info)
                    save_images(decoder_model(quant.detach()), f"quant_image_plot_{global_steps}",
                                out_dir, logging=info)
                d = {"patch_dim": patch_dim, "image_dim": image_dim, "image_C": image_C,
                     "num_embeddings": codebook.num_embeddings,
                     "neighbourhood_range": codebook.neighbourhood_range,
                     "global_steps": global_steps,
                     "checkpoint": {k: v.detach().clone() for k, v in codebook.state_dict().items()}}
                ok = save_model(model_dict=d, dest_path=out_dir, file_name=f"codebook_{global_steps}.pt",
                                logging=info)
                info("Successfully saved model." if ok else "Error occured saving model.")
            info("Cum. Steps: {:,} | Steps: {:,} / {:,} | L.R.: {:.8f} | Recon Loss: {:.5f} | "
                 "Neighbourhood Range: {}".format(global_steps + 1, index + 1, len(loader),
                                                  optim.param_groups[0]["lr"], total / iteration_count,
                                                  codebook.neighbourhood_range))
            global_steps += 1
            if global_steps % neighbourhood_step == 0:
                codebook.decrease_neighbourhood(steps=1)
            if args["max_steps"] is not None and iteration_count >= args["max_steps"]:
                done = True
                break
        if done:
            break


if __name__ == "__main__":
    main()
