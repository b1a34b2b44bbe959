#!/usr/bin/env python3
"""bench.py -- the reference's headline path on MI355X, BASELINE.json config 2:
BMU codebook quantise + base decoder-only Transformer TRAIN step (fwd + CE + bwd +
Adam), 128x128 images == 32x32x4 latents, batch 64 per GPU, fp32.

One "step" = one pass of the hot loop of train_quantized_transformer.py (reference
:404-514) over one batch of synthetic latents already resident in HBM:
  BMU(LR codebook, patch 32) + BMU(HR codebook, patch 2) -> 257-token sequences ->
  random 256-token window -> 7-layer DiT-style decoder (512 / 2048 / 64 heads,
  AdaLN-Zero on window positions) -> CE -> backward -> [RCCL all-reduce] -> Adam.
value = image tokens trained per second, whole job (weak scaling: 64 latents / GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4|c5]

With --gpus N > 1 and no torchrun environment the script launches its own N ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py ...`) BEFORE anything
touches a GPU and exits with their code; under torchrun it is one rank of the job.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(the MFMA GEMM family, timed with HIP events around every launch in a short second
pass -- the headline loop carries no per-launch events) and `cpu_baseline` (the
torch-CPU oracle of the same step on the host cores, a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "quantized-autoregression-image-generator_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP8_MFMA_TFLOPS = 5000.0    # dense fp8, MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBPS = 8000.0
METRIC = "image-tokens/sec at 1/2/4/8 GPUs; BMU argmin GB/s vs HBM peak"

CFG = dict(batch=64, latent=(4, 32, 32), k_lr=512, k_hr=512, lr_patch=32, hr_patch=2, window=256,
           in_dim=512, hidden=2048, heads=64, dec_layers=7, enc_layers=0, lr=1e-4, base=True,
           name="BASELINE configs[1]: BMU (K=512; LR patch 32, HR patch 2) + base decoder-only "
                "Transformer train step, 32x32x4 latents (128x128 images), 257-token sequences, "
                "window 256")
# BASELINE configs[3] per-GPU shard (a parity / scaling test case, not the headline line):
# encoder-decoder stage on 64x64x4 latents, LR patch 4 (256 encoder tokens), HR patch 2
# (1024 tokens, window 256), global batch 64 = 8 per GPU.
CFG_C4 = dict(batch=8, latent=(4, 64, 64), k_lr=512, k_hr=512, lr_patch=4, hr_patch=2, window=256,
              in_dim=512, hidden=2048, heads=64, dec_layers=7, enc_layers=5, lr=1e-4, base=False,
              name="BASELINE configs[3] shard: BMU + encoder-decoder Transformer train step, "
                   "64x64x4 latents (256x256 images), 256 encoder tokens, 1025-token sequences, "
                   "window 256, 8 latents per GPU")
# BASELINE configs[4] per-GPU shard: 8192-entry HR codebook on single latent pixels (patch 1:
# 4096 tokens per 64x64x4 latent), full 4096-token window, encoder over the previous stage's
# 1024 tokens (patch 2, K=512); reduced-precision (bf16 / fp8 MFMA) attention + Linear layers.
CFG_C5 = dict(batch=8, latent=(4, 64, 64), k_lr=512, k_hr=8192, lr_patch=2, hr_patch=1, window=4096,
              in_dim=512, hidden=2048, heads=64, dec_layers=7, enc_layers=5, lr=1e-4, base=False,
              name="BASELINE configs[4] shard: BMU (K=8192, patch 1) + encoder-decoder Transformer "
                   "train step, 64x64x4 latents, 1024 encoder tokens, 4097-token sequences, window "
                   "4096, 8 latents per GPU (global batch 64 on 8 GPUs, as config 4)")
CONFIGS = {"c2": CFG, "c4": CFG_C4, "c5": CFG_C5}
C3_NAME = ("BASELINE configs[2]: full cascade (base + 2 encoder-decoder stages, README sizes, K = 512) autoregressive "
           "generate, 4 images, num_beam = beam_width = 4, window 256, HR patch 8 -> 4 -> 2 (16 / 64 / 256 tokens), "
           "T = 1.0, then codebook gather + conv decoder")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16", "fp8"], default=None,
                    help="f32 = the parity mode and the headline number (default for c2/c4); bf16 / "
                         "fp8 = opt-in reduced precision (Linear GEMMs + attention products on the "
                         "bf16 / fp8 MFMA, fp32 accumulate), the default for --config c5 (bf16); "
                         "reported under its own workload name, never as the c2 headline")
    ap.add_argument("--graph", action="store_true",
                    help="replay the training step from captured HIP graphs (one graph on a single GPU; "
                         "under data parallelism a chain of segments cut at the gradient buckets, the "
                         "bucket all-reduces issued between them).  The default for --config c4, whose "
                         "8-sequence shards are ~1,000 launches of 20-80 us")
    ap.add_argument("--eager", action="store_true", help="plain stream launches (the default except for c4)")
    ap.add_argument("--config", choices=sorted(CONFIGS) + ["c3"], default="c2",
                    help="c2 (default, the headline workload), c3 (cascade generation), or the per-GPU shard of "
                         "config 4 / 5")
    ap.add_argument("--gemm-x3", action="store_true",
                    help="opt-in: the fp32 Linear products on the bf16 matrix pipe from exact three-way operand splits "
                         "(csrc/gemm_x3.hip, option gemm_x3): fp32 operands, results and tolerances, reported under its "
                         "own dtype and workload tag")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="default c2 run on one GPU: do not append the short c3 / c4-shard / c5-shard measurements "
                         "(`configs` object)")
    ap.add_argument("--batch", type=int, default=None, help="override the per-GPU batch")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`bench.py --gpus N` outside torchrun: start the N ranks ourselves.  Nothing in this
    process has touched a GPU (torch is not even imported yet), the ranks are CHILD processes
    and this process only relays their exit code -- no exec of a GPU-initialised process."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def stub_main(args):
    """QARIG_BENCH_STUB=1: the launch / rendezvous / max-over-ranks / one-line plumbing of a
    multi-rank run over gloo on CPU, with no model and no GPU (tests/test_bench_launch.py).
    The line says so ("stub": true) and carries no measurement."""
    import torch
    import torch.distributed as dist
    from qarig import parallel
    world, rank, _ = parallel.init(backend="gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    flat = torch.full((1024,), float(rank + 1))
    for _ in range(args.steps):
        parallel.allreduce_flat(flat.clone())
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "stub": True, "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(t.item()) / max(1, args.steps) * 1e3, 3)}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def build_models(device, cfg, seed=3):
    import torch
    from models.Codebook import Codebook
    from models.Transformer import Transformer
    g = torch.Generator().manual_seed(2)
    C, H, W = cfg["latent"]
    lr_cb = Codebook(patch_dim=(cfg["lr_patch"],) * 2, image_dim=(H, W), image_channel=C,
                     num_embeddings=cfg["k_lr"], init_neighbour_range=4)
    hr_cb = Codebook(patch_dim=(cfg["hr_patch"],) * 2, image_dim=(H, W), image_channel=C,
                     num_embeddings=cfg["k_hr"], init_neighbour_range=4)
    with torch.no_grad():  # trained-like codebooks: tanh(N(0,1))
        lr_cb.codebook.weight.copy_(torch.tanh(torch.randn(lr_cb.codebook.weight.shape, generator=g)))
        hr_cb.codebook.weight.copy_(torch.tanh(torch.randn(hr_cb.codebook.weight.shape, generator=g)))
    torch.manual_seed(seed)
    base = cfg["base"]
    model = Transformer(use_encoder=not base, use_pos_cond=True,
                        num_enc_layers=None if base else cfg["enc_layers"],
                        num_dec_layers=cfg["dec_layers"],
                        num_enc_embedding=None if base else cfg["k_lr"],
                        num_dec_embedding=cfg["k_lr"] + cfg["k_hr"] if base else cfg["k_hr"] + 1,
                        self_attn_heads=cfg["heads"],
                        cross_attn_heads=None if base else cfg["heads"],
                        transformer_in_dim=cfg["in_dim"],
                        transformer_out_dim=cfg["k_hr"] + 1, transformer_hidden_dim=cfg["hidden"],
                        hidden_activation="silu")
    return lr_cb.to(device), hr_cb.to(device), model.to(device)


def cpu_baseline(cfg, budget_s=float(os.environ.get("QARIG_CPU_BASELINE_SECONDS", "15"))):
    """The oracle (torch-CPU restatement, oracle/ref_models.py) running the same train
    step on the host cores: batch 2 sequences of 256 tokens with a random window per
    sample (drawn as the GPU loop draws it), as many steps as fit the budget (>= 1)."""
    import torch
    from oracle import ref_models as rm
    from oracle import bmu as obmu
    # the GPU box gives one GPU's share of the host: 16 cores (task statement)
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    obmu.set_threads(threads)
    torch.manual_seed(3)
    lr_cb, hr_cb, model = build_models("cpu", cfg)
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point)
          for k, v in model.state_dict().items()}
    names = [k for k, v in sd.items() if v.requires_grad]
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    mcfg = dict(use_encoder=False, use_pos_cond=True, num_dec_layers=cfg["dec_layers"],
                self_attn_heads=cfg["heads"], hidden_activation="silu")
    N, W = 2, cfg["window"]
    g = torch.Generator().manual_seed(1)
    rng = torch.Generator().manual_seed(4)
    C, H, Wd = cfg["latent"]
    z = torch.tanh(torch.randn((N, C, H, Wd), generator=g))
    steps, t0 = 0, time.perf_counter()
    while True:
        lr_idx = torch.from_numpy(obmu.bmu(z.numpy(), lr_cb.codebook.weight.detach().numpy(),
                                           (cfg["lr_patch"],) * 2)).reshape(N, -1)
        hr_idx = torch.from_numpy(obmu.bmu(z.numpy(), hr_cb.codebook.weight.detach().numpy(),
                                           (cfg["hr_patch"],) * 2)).reshape(N, -1)
        x_full = torch.cat((lr_idx, hr_idx + cfg["k_lr"]), 1)
        t_full = torch.cat((hr_idx, torch.full((N, 1), cfg["k_hr"])), 1)
        rand = torch.randint(0, x_full.shape[1] - W + 1, (N,), generator=rng)
        pos = rand[:, None] + torch.arange(W)[None]
        x, t = x_full.gather(1, pos), t_full.gather(1, pos)
        for k in names:
            sd[k].grad = None
        loss = rm.cross_entropy(rm.transformer_forward(sd, mcfg, x, None, pos), t)
        loss.backward()
        with torch.no_grad():
            rm.adam_step([sd[k] for k in names], [sd[k].grad for k in names], m, v, steps + 1,
                         cfg["lr"])
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 64:
            break
    return {"value": round(steps * N * W / el, 2), "unit": "image-tokens/s",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} train steps (BMU + random window + fwd + CE + bwd + Adam) of batch "
                      f"{N} x {W} tokens, torch-CPU oracle, {torch.get_num_threads()} threads of "
                      f"{os.cpu_count()} host cpus, {el:.1f} s"}


def bmu_side_measure(device, K=512):
    """BMU argmin GB/s on a C4-sized launch (65,536 patch rows, K=512, D=16), timed
    with HIP events on the launch stream; algorithmic bytes = 4*D + 8 per row.  The codebook is a frozen
    nn.Parameter, as models/Codebook.py hands it over in the Transformer training loop (from its second search on
    the launch takes the codebook's prepared image)."""
    import torch
    from qarig import ops
    g = torch.Generator().manual_seed(9)
    x = torch.tanh(torch.randn((64, 4, 64, 64), generator=g)).to(device)
    w = torch.nn.Parameter(torch.tanh(torch.randn((K, 16), generator=g)).to(device), requires_grad=False)
    for _ in range(2):
        ops.bmu(x, w, (2, 2))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.bmu(x, w, (2, 2))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    rows = 64 * 32 * 32
    gb = rows * (4 * 16 + 8) / 1e9
    fl = rows * 2.0 * K * (16 + 2)
    return {"rows": rows, "K": K, "D": 16, "ms": round(ms, 4),
            "rows_per_s": round(rows / ms * 1e3, 1),
            "algorithmic_GBps": round(gb / ms * 1e3, 2),
            "frac_of_hbm_peak": round(gb / ms * 1e3 / HBM_PEAK_GBPS, 5),
            "TFLOPs": round(fl / ms / 1e9, 2),
            "frac_of_f32_peak": round(fl / ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4)}


def run_c3(args):
    """BASELINE config 3 on one GPU: the cascade of generate_images.py:101-366 -- in the reference's draw order
    (the independent candidates of a chunk run as rows of one batch, every draw numbered as the reference's
    candidate-after-candidate loop numbers it: same draws -> same tokens), the same with the candidates
    literally one after the other, and with --batch-beams (draws numbered by row) -- the decode step alone
    (graph replay) with its weight-streaming roofline, and codebook gather + conv decoder.  `steps` timed
    cascades after `warmup` untimed ones; value = accepted image tokens per second in the reference's order."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_generate as bg
    from models.Codebook import Codebook
    from models.FC_Decoder import FC_Decoder
    from qarig import kvcache
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    K, N, patches = 512, 4, [32, 8, 4, 2]
    ns = argparse.Namespace(stages=3, num_beam=4, beam_width=4, batch_beams=False, no_kv_cache=False, sampler=None)
    torch.manual_seed(69)
    prev0 = torch.randint(0, K, (N, 1), device=dev)
    res = {}
    from qarig import sampling
    ordered_rows = sampling.ORDERED_ROWS
    # the stage models of a loaded generator, kept for the run: from the second cascade on sampling re-uses the
    # decode caches it keeps per model (conditioning tables, captured step graphs); `first_call` = the same
    # cascade on freshly built models (nothing to re-use)
    models = [bg.build_stage_model(s_, K, dev) for s_ in range(3)]
    for name, batched in (("sequential", False), ("first_call", False), ("sequential_one_by_one", False),
                          ("batched_beams", True)):
        ns.batch_beams = batched
        sampling.ORDERED_ROWS = 0 if name == "sequential_one_by_one" else ordered_rows
        ms_ = None if name == "first_call" else models
        for _ in range(max(1, args.warmup)):
            bg.run_cascade(ns, dev, K, N, patches, prev0, ms_)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            prev, stages = bg.run_cascade(ns, dev, K, N, patches, prev0, ms_)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        gen_s = sum(st["seconds"] for st in stages)
        toks = sum(N * st["seq"] for st in stages)
        res[name] = {"accepted_tokens_per_s": round(toks / gen_s, 1), "cascade_ms": round(gen_s * 1e3, 2),
                     "wall_ms": round(dt * 1e3, 1),
                     "stage_tokens_per_s": [st["accepted_tokens_per_s"] for st in stages]}
    sampling.ORDERED_ROWS = ordered_rows
    # the decode step alone: encoder-decoder stage, 4 and 16 rows, graph replay; algorithmic bytes = the fp32
    # weights one step streams (every Linear of the decoder blocks + classifier; the cond projections are a
    # per-position table row, the embedding one row per sequence)
    step = {}
    with torch.no_grad():
        model = bg.build_stage_model(2, K, dev)
        wbytes = 0
        for name, p_ in model.named_parameters():
            if name.startswith("decoder_layers") and name.endswith("linear_layer.0.weight") and \
                    ".cross_attn.k_block" not in name and ".cross_attn.v_block" not in name:
                wbytes += p_.numel() * 4
            if name.startswith("classifier") and name.endswith("weight"):
                wbytes += p_.numel() * 4
        for rows in (4, 16):
            enc = model.encode(torch.randint(0, K, (rows, 64), device=dev))
            positions = [0.0] + [float(i + 1) for i in range(1, 256)]
            cache = kvcache.DecodeCache(model, enc, rows, 256, graph=True, positions=positions)
            ids = torch.randint(0, K, (rows,), device=dev)
            for t in range(8):
                cache.step(ids, None, t)
            cache.ctl[0:1].fill_(255)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 300
            for _ in range(reps):
                cache._graph.replay()
            torch.cuda.synchronize()
            step[rows] = (time.perf_counter() - t0) / reps * 1e3
            del cache
        del model
        cb = Codebook(patch_dim=(2, 2), image_dim=(32, 32), image_channel=4, num_embeddings=K).to(dev)
        dec = FC_Decoder(num_layers=2, image_channel=3, min_channel=256, max_channel=512, latent_channel=4).to(dev).eval()
        dec(cb.get_quantized_image(prev))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dec(cb.get_quantized_image(prev))
        torch.cuda.synchronize()
        img_s = N / ((time.perf_counter() - t0) / 20)
    gbps = wbytes / (step[16] * 1e-3) / 1e9
    out = {"metric": METRIC, "value": res["sequential"]["accepted_tokens_per_s"], "unit": "image-tokens/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": res["sequential"]["cascade_ms"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": C3_NAME, "images": N, "num_beam": 4, "beam_width": 4, "window": 256,
                      "sampler": "fused in-graph sampling kernel (inverse CDF from device-generator uniforms)",
                      "step": "one cascade = 1344 accepted tokens (value: the reference's draw order -- the 4 "
                              "independent candidates of a chunk as rows of one batch, each draw numbered as the "
                              "reference's candidate loop numbers it; sequential_one_by_one: the candidates "
                              "literally one after the other; batched_beams: draws numbered by row; the stage "
                              "models are kept for the run as a loaded generator keeps them -- first_call: the "
                              "same cascade on freshly built models, no decode cache to re-use)"},
           "c3": {"sequential": res["sequential"], "first_call": res["first_call"],
                  "sequential_one_by_one": res["sequential_one_by_one"],
                  "batched_beams": res["batched_beams"],
                  "decode_step_ms_rows4": round(step[4], 4), "decode_step_ms_rows16": round(step[16], 4),
                  "decoder_images_per_s": round(img_s, 1)},
           "roofline": {"bound": "hbm", "kernel": "qarig::decode_linear_kernel<*> chain of one encoder-decoder "
                                                  "decode step (80 dependent launches, 16 rows = 4 images x 4 "
                                                  "candidates, HIP graph replay)",
                        "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": None,
                        "algorithmic_bytes_per_step": wbytes,
                        "rows4_GBps": round(wbytes / (step[4] * 1e-3) / 1e9, 1),
                        "note": "weights streamed once per step / replay time; the step is bound by its ~80 "
                                "dependent launch boundaries (1.7 us each measured) and memory round trips, "
                                "not by bytes"}}
    print(json.dumps(out), flush=True)


def side_configs():
    """The default run's `configs` object: BASELINE configs 3, 4 (per-GPU shard) and 5 (per-GPU shard) measured
    for a few steps each -- and configs 2 and 4 again under the opt-in kernel option gemm_x3 --, every one by this
    script in a child process (started after the headline measurement; the parent only waits), reduced to the
    fields a reader compares."""
    out = {}
    runs = (("c3", ["--config", "c3", "--steps", "1", "--warmup", "1"]),
            ("c4_shard", ["--config", "c4", "--steps", "6", "--warmup", "3", "--no-cpu-baseline"]),
            ("c5_shard", ["--config", "c5", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]),
            # opt-in kernel option gemm_x3: the same fp32 workloads with the Linear products on the bf16 matrix pipe
            ("c2_gemm_x3", ["--config", "c2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--gemm-x3"]),
            ("c4_shard_gemm_x3", ["--config", "c4", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--gemm-x3"]))
    for name, flags in runs:
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + flags + ["--no-side-configs"],
                               capture_output=True, text=True, timeout=150, cwd=ROOT)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            j = json.loads(lines[-1])
            rf = j.get("roofline") or {}
            o = {"workload": j["config"]["workload"], "value": j["value"], "unit": j["unit"],
                 "ms_per_step": j["ms_per_step"], "steps": j["steps"], "warmup": j["warmup"], "dtype": j["dtype"],
                 "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac")},
                 "wall_s": round(time.perf_counter() - t0, 1)}
            if "c3" in j:
                o.update(j["c3"])
            else:
                o["launch"] = j["config"].get("launch")
            out[name] = o
        except Exception as e:        # a side measurement never takes the headline line down
            out[name] = {"error": repr(e)[:300]}
    return out


def main():
    args = parse_args()
    if args.config == "c3":
        assert args.gpus == 1, "config 3 is measured on one GPU (images shard with no collective)"
        return run_c3(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if os.environ.get("QARIG_BENCH_STUB") == "1":
        return stub_main(args)

    import torch
    import torch.distributed as dist
    from qarig import ops, parallel, pipeline
    from qarig.optim import FlatAdam

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    precision = args.precision or ("bf16" if args.config == "c5" else "f32")
    ops.set_precision(precision)
    x3 = bool(args.gemm_x3) or os.environ.get("QARIG_GEMM_X3", "0") not in ("", "0")
    if x3:
        assert precision == "f32", "--gemm-x3 is a form of the fp32 products"
        from qarig import _lib
        _lib.set_option("gemm_x3", int(os.environ.get("QARIG_GEMM_X3", "1")) or 1)    # (2: the 128-tile form only)
    world, rank, local = parallel.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    lr_cb, hr_cb, model = build_models(device, cfg)
    optim = FlatAdam(model.parameters(), lr=cfg["lr"], betas=(0.5, 0.999))
    parallel.broadcast_params(optim)
    if os.environ.get("QARIG_DP_OVERLAP", "1") == "1":
        optim.enable_allreduce_overlap()     # no-op at world 1

    C, H, W = cfg["latent"]
    N = cfg["batch"]
    g = torch.Generator().manual_seed(1)
    # global batch drawn once, sliced per rank; already resident in HBM when timing starts
    z_all = torch.tanh(torch.randn((N * world, C, H, W), generator=g))
    z = parallel.shard(z_all).contiguous().to(device)
    seq = (H // cfg["hr_patch"]) * (W // cfg["hr_patch"]) + 1
    nwin = pipeline.num_windows(seq, cfg["window"])
    rng = torch.Generator().manual_seed(4)

    # graph replay is the c4 default on one GPU; with more ranks the segmented capture (graph segments between
    # the bucket all-reduces) is opt-in (--graph) until a multi-rank RCCL run of it is on record
    use_graph = (args.graph or (args.config == "c4" and world == 1)) and not args.eager
    graphed = pipeline.GraphedTrainStep(model, optim, lr_cb, hr_cb, cfg["base"], cfg["window"]) \
        if use_graph else None

    def eager_step(rand):
        hr_in, lr_in, hr_tg, pos = pipeline.tokenize_window(z, lr_cb, hr_cb, cfg["base"], cfg["window"], rand)
        return pipeline.train_step(model, optim, hr_in, lr_in, hr_tg, pos, pos_bound=seq)

    def step():
        rand = parallel.shard(torch.randint(0, nwin, (N * world,), generator=rng))
        if graphed is not None:
            return graphed(z, rand)
        return eager_step(rand)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enqueued = time.perf_counter() - t0          # host side done; the GPU may still be running
    fence()
    dt = time.perf_counter() - t0
    loss_val = float(loss.item())
    ops.check_index_flag(device, "bench")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # second, short pass: HIP events around every GEMM launch (on the launch stream) for the
    # roofline object; kept out of the headline loop, which carries no per-launch events
    events = []
    if not args.no_kernel_events:
        ev_steps = min(2, args.steps)
        ops.GEMM_EVENTS = []
        t1 = time.perf_counter()
        for _ in range(ev_steps):      # eager launches (a replayed graph has no per-launch events)
            eager_step(parallel.shard(torch.randint(0, nwin, (N * world,), generator=rng)))
        fence()
        dt_ev = time.perf_counter() - t1
        events, ops.GEMM_EVENTS = ops.GEMM_EVENTS, None

    # the gradient exchange on its own (the timed loop overlaps it with backward)
    allreduce = None
    if world > 1:
        flat = optim.flat_grad
        parallel.allreduce_flat(flat)
        fence()
        t2 = time.perf_counter()
        for _ in range(3):
            parallel.allreduce_flat(flat)
        fence()
        ar_ms = (time.perf_counter() - t2) / 3 * 1e3
        nbytes = flat.numel() * 4
        allreduce = {"bytes_per_step": nbytes, "standalone_ms": round(ar_ms, 3),
                     "algbw_GBps": round(nbytes / ar_ms / 1e6, 1),
                     "busbw_GBps": round(nbytes / ar_ms / 1e6 * 2 * (world - 1) / world, 1),
                     "bucket_MiB": 64, "overlapped_with_backward": bool(optim._overlap),
                     "backend": dist.get_backend()}

    tokens = N * world * cfg["window"] * args.steps
    dtype = {"f32": "f32", "bf16": "bf16 products / f32 accumulate",
             "fp8": "fp8 (e4m3) forward products + bf16 backward products / f32 accumulate"}[precision]
    tag = {"f32": "",
           "bf16": " [bf16-MFMA Linear + attention products, f32 accumulate; NOT the fp32 parity configuration]",
           "fp8": " [fp8 (e4m3, per-tensor scale) MFMA on the forward x W^T products of the Linear layers, bf16 MFMA "
                  "on the backward products and in attention, f32 accumulate; NOT the fp32 parity configuration]"
           }[precision]
    if x3:
        dtype = "f32 operands and results; products as 6 bf16-MFMA products of exact 3-way bf16 splits (gemm_x3)"
        tag = (" [opt-in gemm_x3: the Linear products run on the bf16 matrix pipe from exact three-way operand splits, "
               "fp32 accumulate; same operands, epilogues and test tolerances as the fp32-MFMA kernels; NOT the "
               "fp32-MFMA headline]")
    out = {"metric": METRIC,
           "value": round(tokens / dt, 1), "unit": "image-tokens/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": dtype, "data": "synthetic",
           "config": {"workload": cfg["name"] + tag,
                      "batch_per_gpu": N, "global_batch": N * world, "seq_len": cfg["window"],
                      "in_dim": cfg["in_dim"], "hidden_dim": cfg["hidden"], "heads": cfg["heads"],
                      "dec_layers": cfg["dec_layers"], "enc_layers": cfg["enc_layers"],
                      "params": int(optim.total),
                      "parallelism": f"dp{world}", "loss": round(loss_val, 5),
                      "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 3),
                      "launch": ("captured HIP graph replay" + (f" ({len(graphed.graph.graphs)} segments, bucket "
                                 "all-reduces between them)" if graphed.segmented else ""))
                      if graphed is not None else "eager stream launches"}}
    if allreduce is not None:
        out["allreduce"] = allreduce
    if rank == 0:
        if events and os.environ.get("QARIG_GEMM_SHAPES") == "1":
            # per-shape table of the event pass (stderr): launches, mean us, TFLOP/s, algorithmic HBM bytes and
            # the time those take at 8 TB/s
            agg = {}
            for e in events:
                if len(e) > 4:
                    a = agg.setdefault(e[4], [0, 0.0, e[0], e[5]])
                    a[0] += 1
                    a[1] += e[1].elapsed_time(e[2])
            for k, (n, ms, fl, hbm) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                us = ms / n * 1e3
                print(f"{k:64s} n={n // ev_steps:4d}/step {us:8.1f} us {fl / us / 1e6:7.1f} TF  "
                      f"{hbm / 1e6:7.1f} MB = {hbm / 8e6:6.1f} us at 8 TB/s  total {ms / ev_steps:6.2f} ms/step",
                      file=sys.stderr)
        if events:
            times = [(e[0], e[1].elapsed_time(e[2]), e[3] if len(e) > 3 else "") for e in events]
            # dominant kernel = the 128x128-tile GEMM family on the model's (N*S)-row operands;
            # the position-table GEMMs (a few hundred rows, < 1 GFLOP, launch-latency bound by
            # construction) are listed beside it, not averaged into it
            # (reduced-precision mode: the conditioning projections of the position table stay on the fp32
            # grouped kernel; they are listed beside the family, not priced against the bf16 peak)
            f32g = [t for t in times if t[2] == "f32g"]
            times = [t for t in times if t[2] != "f32g"]
            big = [t for t in times if t[0] >= 1e9]
            small = [t for t in times if t[0] < 1e9]
            fl, ms = sum(t[0] for t in big), sum(t[1] for t in big)
            fl_all, ms_all = sum(t[0] for t in times), sum(t[1] for t in times)
            ach = fl / ms / 1e9
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "gemm_traffic.json")
            default_precision = "bf16" if args.config == "c5" else "f32"      # (as profiles/summarize.py names them)
            key = args.config if precision == default_precision else f"{args.config}_{precision}"
            if x3:
                key += "_x3"
            tj = json.load(open(pmc)).get(key) if os.path.exists(pmc) else None
            if tj:
                traffic = tj.get("fabric_bytes_per_launch")
                traffic_src = (f"profiles/gemm_traffic.json[{key}] (round {tj.get('round')}: rocprofv3 --pmc "
                               "FETCH_SIZE / WRITE_SIZE passes of this command, reads doubled per the gfx950 "
                               "calibration; L2-miss (fabric-side) bytes, Infinity-Cache hits included = an "
                               "upper bound of the HBM bytes; not re-measured by this run)")
            # fp8 mode: the forward x W^T products run on gemm_f8_kernel, every other product on the
            # bf16 kernel, so the family is priced against the bf16 peak and the e4m3 launches are
            # listed beside it against their own
            peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS,
                    "fp8": PEAK_BF16_MFMA_TFLOPS}[precision]
            kname = {"f32": "qarig::gemm_dma_pf_kernel<*> / gemm_dma_pf_grouped_kernel<*> / gemm_dma_pf2_kernel<*> "
                            "(fp32 MFMA 32x32x2, 4-stage LDS-DMA ring)",
                     "bf16": "qarig::gemm_lp_kernel<bf16,*> (bf16 MFMA 32x32x16, bf16 operands in HBM)",
                     "fp8": "qarig::gemm_lp_kernel<bf16,*> + gemm_f8_kernel (forward x W^T on fp8 e4m3 "
                            "MFMA 32x32x64, e4m3 operands in HBM); priced against the bf16 peak"}[precision]
            if x3:
                # algorithmic fp32 FLOPs against the bf16 matrix pipe's capacity for them: six bf16 products each
                peak = round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)
                kname = ("qarig::gemm_x3_kernel<*> / gemm_x3_grouped_kernel<*> (six v_mfma_f32_32x32x16_bf16 per block "
                         "on exact 3-way bf16 splits of the fp32 operands) + the fp32-MFMA kernels on the shapes it "
                         "does not take; peak = dense bf16 peak / 6; the fp32-MFMA peak is "
                         f"{PEAK_F32_MFMA_TFLOPS} TFLOP/s")
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": round(ach, 2), "peak": peak,
                               "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": traffic, "traffic_source": traffic_src,
                               "launches": len(big),
                               "avg_launch_us": round(ms / len(big) * 1e3, 2),
                               "avg_launch_gflop": round(fl / len(big) / 1e9, 3),
                               "event_pass": f"{ev_steps} extra steps after the timed region "
                                             f"({dt_ev / ev_steps * 1e3:.2f} ms/step with events)",
                               "gemm_share_of_step": round(ms_all / (dt_ev * 1e3), 3),
                               "small_launches": {"count": len(small), "what": "position-table and other < 1 GFLOP GEMMs",
                                                  "ms_per_step": round(sum(t[1] for t in small) / ev_steps, 3),
                                                  "achieved_all_launches_TFLOPs": round(fl_all / ms_all / 1e9, 2)}}
            if f32g:
                out["roofline"]["fp32_grouped_launches"] = {
                    "what": "position-table conditioning projections on qarig_gemm_f32_grouped (fp32 MFMA)",
                    "count": len(f32g), "ms_per_step": round(sum(t[1] for t in f32g) / ev_steps, 3),
                    "achieved": round(sum(t[0] for t in f32g) / sum(t[1] for t in f32g) / 1e9, 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s"}
            f8 = [t for t in times if t[2] == "f8"]
            if f8:
                f8fl, f8ms = sum(t[0] for t in f8), sum(t[1] for t in f8)
                out["roofline"]["fp8_launches"] = {
                    "kernel": "qarig::gemm_f8_kernel (v_mfma_f32_32x32x64_f8f6f4)", "count": len(f8),
                    "achieved": round(f8fl / f8ms / 1e9, 2), "peak": PEAK_FP8_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(f8fl / f8ms / 1e9 / PEAK_FP8_MFMA_TFLOPS, 4),
                    "share_of_gemm_flops": round(f8fl / fl_all, 3)}
        out["bmu"] = bmu_side_measure(device)
        if not args.no_cpu_baseline and world == 1 and args.config == "c2":
            out["cpu_baseline"] = cpu_baseline(cfg)
        if world == 1 and args.config == "c2" and not args.no_side_configs and precision == "f32" and not x3:
            out["configs"] = side_configs()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
