#!/usr/bin/env python3
"""Train the fully-convolutional autoencoder on MI355X: same command line, JSON keys,
checkpoint dict and log lines as the reference's train_autoencoder.py."""
import argparse
import logging
import pathlib

from models.Autoencoder import Autoencoder
from qarig import cli_common as cc
from qarig import functional as QF
from qarig import parallel
from qarig.optim import FlatAdam
from utils.image_utils import save_images
from utils.model_utils import load_model, save_model
from dataset_loader.image_dataset import ImageDataset


def parse_args():
    p = argparse.ArgumentParser(description="Train Autoencoder models.")
    p.add_argument("--device", choices=["cpu", "cuda"], type=str, default="cpu")
    p.add_argument("--dataset-path", required=True, type=pathlib.Path)
    p.add_argument("--model-path", default=None, required=False, type=pathlib.Path)
    p.add_argument("--load-optim", action="store_true")
    p.add_argument("--batch-size", type=int, default=8)
    p.add_argument("--checkpoint-step", type=int, default=1_000)
    p.add_argument("--lr-step", type=int, default=50_000)
    p.add_argument("--max-epoch", type=int, default=1_000)
    p.add_argument("--config-path", required=True, type=pathlib.Path)
    p.add_argument("--out-dir", required=True, type=pathlib.Path)
    p.add_argument("--max-steps", type=int, default=None, help="(additive) stop after N steps.")
    return vars(p.parse_args())


def main():
    project_name = "Autoencoder"
    args = parse_args()
    cfg = cc.read_config(args["config_path"])
    device, world, rank = cc.require_gpu(args["device"])
    out_dir = args["out_dir"]
    cc.setup_logging(out_dir, project_name, rank)
    info = logging.info
    hp = dict(num_layers=cfg["num_layers"], image_channel=cfg["image_channel"],
              min_channel=cfg["min_channel"], max_channel=cfg["max_channel"],
              latent_channel=cfg["latent_channel"],
              hidden_activation_type=cfg["hidden_activation_type"],
              use_final_enc_activation=cfg["use_final_enc_activation"],
              encoder_activation_type="silu" if not cfg["use_final_enc_activation"]
              else cfg["encoder_activation_type"],
              use_final_dec_activation=cfg["use_final_dec_activation"],
              decoder_activation_type="tanh" if not cfg["use_final_dec_activation"]
              else cfg["decoder_activation_type"])
    model = Autoencoder(**hp).to(device)
    model_lr = cfg["model_lr"]
    saved = None
    if args["model_path"] is not None:
        ok, saved = load_model(args["model_path"])
        if not ok:
            raise Exception("An error occured while loading model checkpoint!")
        model.custom_load_state_dict(saved["model"])
    optim = FlatAdam(model.parameters(), lr=model_lr, betas=(0.5, 0.999))
    if saved is not None:
        if args["load_optim"]:
            optim.load_state_dict(saved["model_optimizer"])
        else:
            for g in optim.param_groups:
                g["lr"] = model_lr
    parallel.broadcast_params(optim)
    dataset = ImageDataset(dataset_path=args["dataset_path"])
    loader = cc.ShardedLoader(dataset, args["batch_size"], num_workers=4, shuffle=True)
    info(f"{project_name}")
    info(f"Output Dir: {out_dir}")
    info(f"Model size: {sum(p.numel() for p in model.parameters()):,}")
    info("#" * 100)

    global_steps, done = 0, False
    for _ in range(0, args["max_epoch"]):
        total, iteration_count = 0.0, 0
        for index, image in enumerate(loader):
            image = image.to(device)
            iteration_count += 1
            model.train()
            optim.zero_grad()
            recon = model(image)
            loss = QF.mse_loss(recon, image)
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
            if global_steps % args["checkpoint_step"] == 0 and global_steps >= 0 and rank == 0:
                d = dict(hp)
                d["model"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
                d["model_optimizer"] = optim.state_dict()
                ok = save_model(model_dict=d, dest_path=out_dir, file_name=f"model_{global_steps}.pt",
                                logging=info)
                info("Successfully saved model." if ok else "Error occured saving model.")
                save_images(image, f"ground_truth_{global_steps}", out_dir, logging=info)
                save_images(recon.detach(), f"recon_{global_steps}", out_dir, logging=info)
            info("Cum. Steps: {:,} | Steps: {:,} / {:,} | L.R.: {:.8f} | Recon Loss: {:.5f}".format(
                global_steps + 1, index + 1, len(loader), optim.param_groups[0]["lr"],
                total / iteration_count))
            global_steps += 1
            if args["max_steps"] is not None and global_steps >= args["max_steps"]:
                done = True
                break
        if done:
            break


if __name__ == "__main__":
    main()
