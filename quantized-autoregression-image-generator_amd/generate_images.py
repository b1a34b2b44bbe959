#!/usr/bin/env python3
"""Cascade image generation on MI355X: same command line, stage-config JSON and output
files as the reference's generate_images.py (base stage "0" + encoder-decoder stages,
best-of-`num_beam` chunks of `beam_width` tokens, sliding window).

Under torchrun (one process per GPU) the images are sharded across the ranks -- they are independent, no
collective on the data path -- and every stage's token ids are gathered (a few KB) so that rank 0 decodes and
writes the same image grids a single process writes; rank r seeds its generator with --seed + r."""
import argparse
import os
import pathlib

import torch

from models.Transformer import Transformer
from qarig import cli_common as cc
from qarig import ops, parallel, sampling
from utils.image_utils import save_images
from utils.model_utils import load_model


def parse_args():
    p = argparse.ArgumentParser(description="Generate Images.")
    p.add_argument("--device", choices=["cpu", "cuda"], type=str, default="cpu",
                   help="Which hardware device will model run on.")
    p.add_argument("--decoder-path", required=True, type=pathlib.Path,
                   help="File path to pre-trained decoder model.")
    p.add_argument("--num-images", type=int, default=25, help="Num of images to generate.")
    p.add_argument("--seed", type=int, default=None, help="Seed value.")
    p.add_argument("--config-path", required=True, type=pathlib.Path,
                   help="File path to load json config file.")
    p.add_argument("--out-dir", required=True, type=pathlib.Path, help="File path to output directory.")
    p.add_argument("--batch-beams", action="store_true",
                   help="(additive) number the draws of the num_beam candidate chunks by batch row instead of "
                        "candidate after candidate as the reference loop does (the candidates are evaluated as one "
                        "batch either way when images x num_beam <= 16; with --sampler torch this is what batches them).")
    p.add_argument("--sampler", choices=["fused", "torch"], default=None,
                   help="(additive) cached decoding: 'fused' (default) draws inside the decode loop's own kernel from "
                        "uniforms of the device generator -- the whole chunk search stays on the GPU; 'torch' makes one "
                        "torch.multinomial call per token like the reference (its generator stream for a given --seed).")
    p.add_argument("--no-kv-cache", action="store_true",
                   help="re-run the whole window for every token like the reference instead of "
                        "decoding one token per step from a key/value cache")
    return vars(p.parse_args())


def main():
    args = parse_args()
    device, world, rank = cc.require_gpu(args["device"])
    num_images, out_dir = args["num_images"], args["out_dir"]
    lo, hi = parallel.shard_range(num_images)
    n_local = hi - lo                                    # this rank's images
    log = print if rank == 0 else (lambda *a, **k: None)
    os.makedirs(out_dir, exist_ok=True)
    if args["seed"] is not None:
        torch.manual_seed(args["seed"] + rank)
    config = cc.read_config(args["config_path"])
    decoder_model, _ = cc.load_decoder(args["decoder_path"], device)
    decoder_model.eval()

    hr_input = None
    for index, data in config.items():          # stages "0", "1", "2" in file order
        log(f"Model: {int(index):,}")
        lr_codebook = None
        if data["lr_codebook_path"] is not None:
            lr_codebook, lr_d = cc.load_codebook(data["lr_codebook_path"], device)
        hr_codebook, hr_d = cc.load_codebook(data["hr_codebook_path"], device)
        k_hr = hr_d["num_embeddings"]
        img_H, img_W = hr_d["image_dim"]
        pH, pW = hr_d["patch_dim"]
        total_Seq = (img_H // pH) * (img_W // pW)
        if total_Seq % data["beam_width"] != 0:
            raise Exception("Invalid value for beam_width!")

        ok, md = load_model(data["model_path"])
        if not ok:
            raise Exception("An error occured while loading model checkpoint!")
        model = Transformer(
            use_encoder=not md["train_base_model"], use_pos_cond=md["use_sliding_window"],
            num_enc_layers=md["num_enc_layers"], num_dec_layers=md["num_dec_layers"],
            num_enc_embedding=md["num_enc_embedding"], num_dec_embedding=md["num_dec_embedding"],
            self_attn_heads=md["self_attn_heads"], cross_attn_heads=md["cross_attn_heads"],
            transformer_in_dim=md["transformer_in_dim"], transformer_out_dim=md["transformer_out_dim"],
            transformer_hidden_dim=md["transformer_hidden_dim"],
            hidden_activation=md["hidden_activation"])
        model.custom_load_state_dict(md["model"])
        model = model.to(device).eval()

        with torch.no_grad():
            shift = 0
            if index == "0":
                # base stage: a random LR-codebook token is the image-level condition
                k_lr = lr_d["num_embeddings"]
                lr_input = None
                hr_input = torch.randint(low=0, high=k_lr, size=(n_local, 1), device=device)
                all_cond = parallel.gather_rows(hr_input, num_images)
                if rank == 0:
                    cond = decoder_model(lr_codebook.get_quantized_image(indices=all_cond, unpatchify_input=True))
                    save_images(images=cond, file_name="recon_model_Cond", dest_path=out_dir, logging=print)
                shift = k_lr
            else:
                lr_input = hr_input                                   # previous stage's tokens
                hr_input = torch.full((n_local, 1), k_hr, dtype=torch.int64, device=device)

            if n_local:
                hr_input = sampling.generate_tokens(
                    model, hr_input, lr_input, total_Seq, data["temperature"], md["use_sliding_window"],
                    md["sliding_window"], end_token=k_hr, shift=shift, num_beam=data["num_beam"],
                    beam_width=data["beam_width"], mode="generate",
                    progress=lambda i, t: log(f"{i:,} / {t:,}"), batch_beams=args["batch_beams"],
                    use_kv_cache=not args["no_kv_cache"], sampler=args["sampler"])
                hr_input = hr_input[:, 1:] - shift
            else:                                        # more ranks than images: nothing to generate here
                hr_input = torch.zeros((0, total_Seq), dtype=torch.int64, device=device)
            ops.check_index_flag(device, f"stage {index} tokens")
            all_tokens = parallel.gather_rows(hr_input, num_images)
            if rank == 0:
                recon = decoder_model(hr_codebook.get_quantized_image(indices=all_tokens, unpatchify_input=True))
                ops.check_index_flag(device, f"stage {index} tokens")
                save_images(images=recon, file_name=f"recon_model_{index}", dest_path=out_dir, logging=print)
        sampling.decode_cache_clear()        # (the stage's decode caches hold its model: let both go, as the reference does)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
