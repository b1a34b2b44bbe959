#!/usr/bin/env python3
"""BASELINE config 3: full cascade (base + 2 encoder-decoder stages) autoregressive
generation, num_beam = beam_width = 4, README model sizes, random weights, 1 MI355X.
Reports accepted tokens/s and model-evaluation tokens/s per stage, then codebook gather
+ conv decoder images/s.   python tools/bench_generate.py [--images 4] [--stages 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from models.Codebook import Codebook  # noqa: E402
from models.FC_Decoder import FC_Decoder  # noqa: E402
from models.Transformer import Transformer  # noqa: E402
from qarig import sampling  # noqa: E402


def build_stage_model(s, K, dev):
    base = s == 0
    return Transformer(use_encoder=not base, use_pos_cond=True,
                       num_enc_layers=None if base else 5, num_dec_layers=7,
                       num_enc_embedding=None if base else K,
                       num_dec_embedding=2 * K if base else K + 1, self_attn_heads=64,
                       cross_attn_heads=None if base else 64, transformer_in_dim=512,
                       transformer_out_dim=K + 1, transformer_hidden_dim=2048).to(dev).eval()


def run_cascade(args, dev, K, N, patches, prev, models=None):
    """One pass over the stages; returns (last-stage tokens, per-stage records).  models: the stage models of a
    loaded generator (kept between cascades: the decode caches sampling keeps per model -- conditioning tables,
    captured step graphs -- are reused from the second cascade on); None builds fresh ones per stage, i.e. every
    cascade pays what a first call pays."""
    stages = []
    for s in range(args.stages):
        base = s == 0
        model = models[s] if models is not None else build_stage_model(s, K, dev)
        total = (32 // patches[s + 1]) ** 2
        first = prev if base else torch.full((N, 1), K, dtype=torch.int64, device=dev)
        lr_in = None if base else prev
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = sampling.generate_tokens(model, first, lr_in, total, 1.0, True, 256, end_token=K,
                                        shift=K if base else 0, num_beam=args.num_beam,
                                        beam_width=args.beam_width, mode="generate",
                                        batch_beams=args.batch_beams,
                                        use_kv_cache=not args.no_kv_cache, sampler=args.sampler)
        t_host = time.perf_counter() - t0           # until the call returned: everything enqueued, nothing awaited
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prev = toks[:, 1:] - (K if base else 0)
        acc = N * total
        stages.append({"stage": s, "seq": total, "seconds": round(dt, 3), "host_enqueue_seconds": round(t_host, 3),
                       "accepted_tokens_per_s": round(acc / dt, 1),
                       "model_eval_tokens_per_s": round(acc * args.num_beam / dt, 1)})
        del model
    return prev, stages


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--stages", type=int, default=3)
    ap.add_argument("--num-beam", type=int, default=4)
    ap.add_argument("--beam-width", type=int, default=4)
    ap.add_argument("--batch-beams", action="store_true")
    ap.add_argument("--one-by-one", action="store_true",
                    help="fused sampler without --batch-beams: run the candidates of a chunk literally one after the "
                         "other instead of as rows of one batch under the reference's draw numbers")
    ap.add_argument("--no-kv-cache", action="store_true")
    ap.add_argument("--sampler", choices=["fused", "torch"], default=None,
                    help="cached loop: in-graph sampling kernel (default) or one torch.multinomial per token")
    ap.add_argument("--rebuild-models", action="store_true",
                    help="new random stage models for every cascade (no decode cache is ever reused: the cost of a "
                         "generator's first call) instead of one set kept for the run")
    ap.add_argument("--cold", action="store_true",
                    help="skip the untimed warm-up pass (code-object loads, allocator growth, "
                         "first graph instantiation then land in stage 0)")
    args = ap.parse_args()
    if args.one_by_one:
        sampling.ORDERED_ROWS = 0
    if os.environ.get("QARIG_NO_GROUPS") == "1":      # A/B: the whole batch at once on the general kernels
        sampling.GROUP_IMAGES = False
    dev = torch.device("cuda", 0)
    torch.manual_seed(69)
    K, N = 512, args.images
    patches = [32, 8, 4, 2][:args.stages + 1]       # conditional, then HR patch 8 -> 4 -> 2
    cbs = [Codebook(patch_dim=(p, p), image_dim=(32, 32), image_channel=4, num_embeddings=K).to(dev)
           for p in patches]
    dec = FC_Decoder(num_layers=2, image_channel=3, min_channel=256, max_channel=512,
                     latent_channel=4).to(dev).eval()
    out = {"config": f"cascade generate, {args.stages} stages, N={N}, num_beam={args.num_beam}, "
                     f"beam_width={args.beam_width}, window 256, fp32, batch_beams={args.batch_beams}, "
                     f"candidates={'one by one' if args.one_by_one else 'rows of one batch'}, "
                     f"kv_cache={not args.no_kv_cache}, sampler={args.sampler or sampling.DEFAULT_SAMPLER}, warm={not args.cold}"}
    prev0 = torch.randint(0, K, (N, 1), device=dev)
    models = None if args.rebuild_models else [build_stage_model(s_, K, dev) for s_ in range(args.stages)]
    out["config"] += f", models={'rebuilt per cascade' if models is None else 'kept'}"
    if not args.cold:
        run_cascade(args, dev, K, N, patches, prev0, models)
    prev, out["stages"] = run_cascade(args, dev, K, N, patches, prev0, models)
    tot_tokens = sum(N * st["seq"] for st in out["stages"])
    tot_time = sum(st["seconds"] for st in out["stages"])
    with torch.no_grad():
        img = dec(cbs[args.stages].get_quantized_image(prev))     # first call: re-ordered weight copies are cached
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            img = dec(cbs[args.stages].get_quantized_image(prev))
        torch.cuda.synchronize()
        ddt = (time.perf_counter() - t0) / reps
    out["decode_images_per_s"] = round(N / ddt, 1)
    out["accepted_tokens_per_s"] = round(tot_tokens / tot_time, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
