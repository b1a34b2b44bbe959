#!/usr/bin/env python3
"""Train the quantized (cascaded) Transformer on MI355X.  Same command line, JSON config
keys, log lines, checkpoint dict and output tree as the reference's
train_quantized_transformer.py; the compute is the HIP path (models/, qarig/).

New, additive: run under `python -m torch.distributed.run --nproc-per-node N ...` for
data parallelism (--batch-size is then per GPU; one RCCL all-reduce of the flat gradient
buffer per step; rank 0 logs, checkpoints and samples)."""
import argparse
import logging
import pathlib

import torch

from models.Transformer import Transformer
from qarig import cli_common as cc
from qarig import ops, parallel, pipeline, sampling
from qarig.optim import FlatAdam
from utils.image_utils import save_images
from utils.model_utils import load_model, save_model
from dataset_loader.feature_map_dataset import FeatureMapDataset


def restricted_float(x):
    try:
        x = float(x)
    except ValueError:
        raise argparse.ArgumentTypeError("%r not a floating-point literal" % (x,))
    if x < 0.1:
        raise argparse.ArgumentTypeError("%r not in range > 0.1" % (x,))
    return x


def parse_args():
    p = argparse.ArgumentParser(description="Train Quantized Transformer models.")
    p.add_argument("--device", choices=["cpu", "cuda"], type=str, default="cpu",
                   help="Which hardware device will model run on.")
    p.add_argument("--dataset-path", required=True, type=pathlib.Path,
                   help="File path to feature map dataset json file.")
    p.add_argument("--train-base-model", action="store_true", help="Train Base Model, Decoder-only.")
    p.add_argument("--decoder-path", required=True, type=pathlib.Path,
                   help="File path to pre-trained decoder model.")
    p.add_argument("--lr-codebook-path", required=True, type=pathlib.Path,
                   help="File path to saved Low-Res codebook.")
    p.add_argument("--hr-codebook-path", required=True, type=pathlib.Path,
                   help="File path to saved High-Res codebook.")
    p.add_argument("--model-path", default=None, required=False, type=pathlib.Path,
                   help="File path to saved model checkpoint.")
    p.add_argument("--test-num-sample", type=int, default=25, help="Num samples for testing dataset.")
    p.add_argument("--load-optim", action="store_true", help="Load saved optim parameters with model.")
    p.add_argument("--batch-size", type=int, default=8, help="Batch size for dataset.")
    p.add_argument("--temperature", type=restricted_float, default=1.0,
                   help="Temperature for softmax sampling.")
    p.add_argument("--checkpoint-step", type=int, default=1_000,
                   help="Steps at which checkpoint takes place.")
    p.add_argument("--lr-step", type=int, default=50_000, help="Steps before halving learning rate.")
    p.add_argument("--max-epoch", type=int, default=1_000, help="Maximum epoch for training model.")
    p.add_argument("--use-activation-checkpoint", action="store_true",
                   help="Use Activation Checkpointing; trade-off memory footprint and compute.")
    p.add_argument("--config-path", required=True, type=pathlib.Path,
                   help="File path to load json config file.")
    p.add_argument("--out-dir", required=True, type=pathlib.Path, help="File path to output directory.")
    p.add_argument("--graph-step", action="store_true",
                   help="(additive) replay the training step from captured HIP graphs: removes the Python "
                        "launch overhead for small per-GPU batches; full batches only.  Under torchrun the "
                        "capture is a chain of segments cut at the gradient buckets, the bucket all-reduces "
                        "issued between them")
    p.add_argument("--gemm-x3", action="store_true",
                   help="(additive) run the eligible fp32 Linear products on the bf16 matrix pipe from exact three-way "
                        "bf16 splits of the fp32 operands (library option gemm_x3, csrc/gemm_x3.hip): fp32 operands, "
                        "results and tolerances, about 1.2 x the step throughput at README sizes")
    p.add_argument("--packed-dataset", default=None,
                   help="(additive) one (N,C,H,W) float32 .npy holding every latent of --dataset-path "
                        "(dataset_loader.prefetch.pack_feature_maps): read through one mmap instead of "
                        "one file per item; same items, same order")
    p.add_argument("--max-steps", type=int, default=None,
                   help="(additive) stop after this many optimiser steps.")
    return vars(p.parse_args())


def main():
    project_name = "Quantized Transformer"
    args = parse_args()
    cfg = cc.read_config(args["config_path"])
    device, world, rank = cc.require_gpu(args["device"])
    if args["gemm_x3"]:
        from qarig import _lib
        _lib.set_option("gemm_x3", 1)
    out_dir = args["out_dir"]
    cc.setup_logging(out_dir, project_name, rank)
    temperature = args["temperature"]
    train_base_model = args["train_base_model"]
    test_num_sample = args["test_num_sample"]

    decoder_model, dec_d = cc.load_decoder(args["decoder_path"], device)
    lr_codebook, lr_d = cc.load_codebook(args["lr_codebook_path"], device, "Low-Resolution codebook")
    hr_codebook, hr_d = cc.load_codebook(args["hr_codebook_path"], device, "Low-Resolution codebook")
    lr_num_embeddings, hr_num_embeddings = lr_d["num_embeddings"], hr_d["num_embeddings"]
    img_H, img_W = hr_d["image_dim"]
    hr_pH, hr_pW = hr_d["patch_dim"]
    total_hr_Seq = (img_H // hr_pH) * (img_W // hr_pW)

    # vocabularies (reference :258-296)
    if train_base_model:
        num_enc_layers = num_enc_embedding = cross_attn_heads = None
        num_dec_embedding = lr_num_embeddings + hr_num_embeddings
    else:
        num_enc_embedding = lr_num_embeddings
        num_enc_layers = cfg["num_enc_layers"]
        cross_attn_heads = cfg["cross_attn_heads"]
        num_dec_embedding = hr_num_embeddings + 1          # + <start>
    use_sliding_window = cfg["use_sliding_window"]
    sliding_window = cfg["sliding_window"] if use_sliding_window else None
    hp = dict(num_dec_layers=cfg["num_dec_layers"], self_attn_heads=cfg["self_attn_heads"],
              transformer_in_dim=cfg["in_dim"], transformer_out_dim=hr_num_embeddings + 1,
              transformer_hidden_dim=cfg["hidden_dim"], hidden_activation=cfg["hidden_activation"])

    model = Transformer(use_encoder=not train_base_model, use_pos_cond=use_sliding_window,
                        num_enc_layers=num_enc_layers, num_enc_embedding=num_enc_embedding,
                        num_dec_embedding=num_dec_embedding, cross_attn_heads=cross_attn_heads,
                        use_activation_checkpoint=args["use_activation_checkpoint"], **hp).to(device)
    model_lr = cfg["model_lr"]
    if args["model_path"] is not None:
        ok, saved = load_model(args["model_path"])
        if not ok:
            raise Exception("An error occured while loading model checkpoint!")
        model.custom_load_state_dict(saved["model"])
    optim = FlatAdam(model.parameters(), lr=model_lr, betas=(0.5, 0.999))
    if args["model_path"] is not None:
        if args["load_optim"]:
            optim.load_state_dict(saved["model_optimizer"])
        else:
            for g in optim.param_groups:
                g["lr"] = model_lr
    parallel.broadcast_params(optim)
    optim.enable_allreduce_overlap()     # no-op at world 1

    dataset = FeatureMapDataset(dataset_path=args["dataset_path"], load_image=False,
                                return_filepaths=False)
    train_set = dataset
    if args["packed_dataset"]:
        from dataset_loader.prefetch import PackedFeatureMapDataset
        train_set = PackedFeatureMapDataset(args["packed_dataset"])
        assert len(train_set) == len(dataset), "--packed-dataset does not match --dataset-path"
    # batches are copied to the device from pinned memory on a side stream, two ahead of the
    # step that consumes them (the reference copies each batch synchronously, :407)
    from dataset_loader.prefetch import DevicePrefetcher
    loader = DevicePrefetcher(cc.ShardedLoader(train_set, args["batch_size"], num_workers=4, shuffle=True),
                              device, depth=2)
    test_loader = torch.utils.data.DataLoader(dataset, batch_size=test_num_sample, num_workers=2,
                                              shuffle=True)

    info = logging.info
    info(f"{project_name}")
    info(f"Output Dir: {out_dir}")
    info(f"Model size: {sum(p.numel() for p in model.parameters()):,}")
    info("#" * 100)
    info("Decoder Parameters.")
    for k in ("num_layers", "image_channel", "min_channel", "max_channel", "latent_channel"):
        info(f"{k.replace('_', ' ').title()}: {dec_d[k]:,}")
    info(f"Hidden activation type: {dec_d['hidden_activation_type']}")
    info(f"Decoder activation type: {dec_d['decoder_activation_type']}")
    info("#" * 100)
    info("Codebook Parameters.")
    info(f"Low Res Patch size: {lr_d['patch_dim']}")
    info(f"Low Res Num Embeddings: {lr_num_embeddings:,}")
    info(f"High Res Patch size: {hr_d['patch_dim']}")
    info(f"High Res Num Embeddings: {hr_num_embeddings:,}")
    info("#" * 100)
    info("Transformer Parameters.")
    if use_sliding_window:
        info(f"Sliding Window: {sliding_window:,}")
    info(f"Num Encoder Embedding: {num_enc_embedding}")
    info(f"Num Encoder Layers: {num_enc_layers}")
    info(f"Num Decoder Embedding: {num_dec_embedding:,}")
    info(f"Num Decoder Layers: {hp['num_dec_layers']:,}")
    info(f"Self Attention Heads: {hp['self_attn_heads']:,}")
    info(f"Cross Attention Heads: {cross_attn_heads}")
    info(f"In Dim: {hp['transformer_in_dim']:,}")
    info(f"Out Dim: {hp['transformer_out_dim']:,}")
    info(f"Hidden Dim: {hp['transformer_hidden_dim']:,}")
    info(f"Hidden activation: {hp['hidden_activation']}")
    info("#" * 100)
    info("Training Parameters.")
    info(f"Max Epoch: {args['max_epoch']:,}")
    info(f"Batch Size: {args['batch_size']:,}" + (f" x {world} GPUs" if world > 1 else ""))
    info(f"Model LR Update size: {args['lr_step']:,}")
    info(f"Model Checkpoint step: {args['checkpoint_step']:,}")
    info("#" * 100)
    info("Sampling Parameters.")
    info(f"Temperature: {temperature:,}")
    info("#" * 100)

    def checkpoint(global_steps):
        model_dict = {"train_base_model": train_base_model, "use_sliding_window": use_sliding_window,
                      "sliding_window": sliding_window, "num_enc_embedding": num_enc_embedding,
                      "num_dec_embedding": num_dec_embedding, "num_enc_layers": num_enc_layers,
                      "num_dec_layers": hp["num_dec_layers"], "self_attn_heads": hp["self_attn_heads"],
                      "cross_attn_heads": cross_attn_heads,
                      "transformer_in_dim": hp["transformer_in_dim"],
                      "transformer_out_dim": hp["transformer_out_dim"],
                      "transformer_hidden_dim": hp["transformer_hidden_dim"],
                      "hidden_activation": hp["hidden_activation"],
                      "model": {k: v.detach().clone() for k, v in model.state_dict().items()},
                      "model_optimizer": optim.state_dict()}
        ok = save_model(model_dict=model_dict, dest_path=out_dir, file_name=f"model_{global_steps}.pt",
                        logging=info)
        info("Successfully saved model." if ok else "Error occured saving model.")
        # one full autoregressive sample per checkpoint (reference :536-677)
        model.eval()
        with torch.no_grad():
            fm = next(iter(test_loader)).to(device)
            n = fm.shape[0]
            save_images(decoder_model(fm), f"ground_truth_{global_steps}", out_dir, logging=info)
            save_images(decoder_model(lr_codebook(fm)), f"low_res_cond_{global_steps}", out_dir,
                        logging=info)
            save_images(decoder_model(hr_codebook(fm)), f"high_res_example_{global_steps}", out_dir,
                        logging=info)
            lr_idx = lr_codebook.get_patches_bmu(fm, reshape=True)
            if train_base_model:
                first, lr_in, shift = lr_idx, None, lr_num_embeddings
            else:
                first = torch.full((n, 1), hr_num_embeddings, dtype=torch.int64, device=device)
                lr_in, shift = lr_idx, 0
            toks = sampling.generate_tokens(
                model, first, lr_in, total_hr_Seq, temperature, use_sliding_window, sliding_window,
                end_token=hr_num_embeddings, shift=shift, mode="train",
                progress=lambda i, t: print(f"{i:,} / {t:,}"))
            toks = toks[:, first.shape[1]:] if train_base_model else toks[:, 1:]
            if train_base_model:
                toks = toks - lr_num_embeddings
                toks[toks == hr_num_embeddings] = lr_num_embeddings
            else:
                toks[toks == hr_num_embeddings] = 0
            save_images(decoder_model(hr_codebook.get_quantized_image(toks, unpatchify_input=True)),
                        f"high_res_recon_{global_steps}", out_dir, logging=info)
        model.train()
        torch.cuda.empty_cache()

    graphed = None
    batch_size = args["batch_size"]
    lr_pH, lr_pW = lr_d["patch_dim"]
    # tokens per sequence before windowing: the LR tokens (base model) or <start>, then the HR tokens
    graphed_seq = total_hr_Seq + ((img_H // lr_pH) * (img_W // lr_pW) if train_base_model else 1)
    if args["graph_step"]:
        graphed = pipeline.GraphedTrainStep(model, optim, lr_codebook, hr_codebook, train_base_model,
                                            sliding_window if use_sliding_window else None)
    global_steps = 0
    done = False
    for epoch in range(0, args["max_epoch"]):
        total_loss, iteration_count = 0.0, 0
        for index, feature_map in enumerate(loader):
            iteration_count += 1
            feature_map = feature_map.to(device)
            N = feature_map.shape[0]
            model.train()
            if graphed is not None and N == batch_size:
                rand = None
                if use_sliding_window:
                    rand = torch.randint(low=0, high=pipeline.num_windows(graphed_seq, sliding_window),
                                         size=(N * world,))             # CPU global RNG, drawn globally
                    rand = parallel.shard(parallel.broadcast_host_tensor(rand))
                loss = graphed(feature_map, rand)
            else:
                seq_total = graphed_seq
                rand = None
                if use_sliding_window:
                    nwin = pipeline.num_windows(seq_total, sliding_window)
                    rand = torch.randint(low=0, high=nwin, size=(N * world,))   # CPU global RNG
                    rand = parallel.shard(parallel.broadcast_host_tensor(rand))
                hr_in, lr_in, hr_tg, pos_idx = pipeline.tokenize_window(
                    feature_map, lr_codebook, hr_codebook, train_base_model,
                    sliding_window if use_sliding_window else None, rand)
                loss = pipeline.train_step(model, optim, hr_in, lr_in, hr_tg, pos_idx,
                                           pos_bound=seq_total)
            loss_val = loss.item()                                   # the reference's per-step sync
            ops.check_index_flag(device, "training batch")
            if loss_val != loss_val:
                raise Exception("NaN encountered during training.")
            total_loss += loss_val
            if global_steps % args["lr_step"] == 0 and global_steps > 0:
                cc.halve_lr(optim)
            if global_steps % args["checkpoint_step"] == 0 and global_steps >= 0 and rank == 0:
                checkpoint(global_steps)
            info("Cum. Steps: {:,} | Steps: {:,} / {:,} | L.R.: {:.8f} | Recon Loss: {:.5f}".format(
                global_steps + 1, index + 1, len(loader), optim.param_groups[0]["lr"],
                total_loss / iteration_count))
            global_steps += 1
            if args["max_steps"] is not None and global_steps >= args["max_steps"]:
                done = True
                break
        if done:
            break


if __name__ == "__main__":
    main()
