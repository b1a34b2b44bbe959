#!/usr/bin/env python3
"""Single-token decode step of a README-size encoder-decoder stage (config 3, stage 3) in isolation:
ms per captured-graph replay (the device chain alone), per DecodeCache.step (with the host feeds) and
C-ABI launches per step: "table" = the per-position table of the cond projections + fused norms (what
generation runs), "fused" = per-token cond path + fused norms, "separate" = every norm its own launch.
    python tools/decode_step_probe.py [--rows 4] [--steps 200]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from bench_generate import build_stage_model  # noqa: E402
from qarig import kvcache, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--base", action="store_true", help="decoder-only base stage")
    ap.add_argument("--opt", action="append", default=[], help="name=value kernel-selection option (qarig_set_option)")
    args = ap.parse_args()
    for o in args.opt:
        k, v = o.split("=")
        _lib.load().qarig_set_option(k.encode(), int(v))
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    K, B, S = 512, args.rows, 256
    model = build_stage_model(0 if args.base else 2, K, dev)
    out = {"rows": B, "window": S, "options": args.opt, "stage": "base" if args.base else "encoder-decoder", "steps": args.steps}
    with torch.no_grad():
        enc = None if args.base else model.encode(torch.randint(0, K, (B, 64), device=dev))
        ids = torch.randint(0, K, (B,), device=dev)
        pos = torch.rand(B, device=dev) * 100
        positions = [0.0] + [float(i + 1) for i in range(1, S)]
        for name, fused, table in (("table", True, True), ("fused", True, False), ("separate", False, False)):
            kvcache.FUSE_NORMS = fused
            eager = kvcache.DecodeCache(model, enc, B, S, graph=False, positions=positions if table else None)
            n0 = _lib.N_CALLS
            eager.step(ids, pos, 0)
            launches = _lib.N_CALLS - n0
            cache = kvcache.DecodeCache(model, enc, B, S, graph=True, positions=positions if table else None)
            for t in range(8):
                cache.step(ids, pos, t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(args.steps):
                cache.step(ids, pos, 8 + t % (S - 8))
            torch.cuda.synchronize()
            step_ms = (time.perf_counter() - t0) / args.steps * 1e3
            cache.ctl[0:1].fill_(S - 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(args.steps):
                cache._graph.replay()
            torch.cuda.synchronize()
            replay_ms = (time.perf_counter() - t0) / args.steps * 1e3
            out[name] = {"qarig_launches_per_step": launches, "step_ms": round(step_ms, 4),
                         "graph_replay_ms": round(replay_ms, 4)}
        kvcache.FUSE_NORMS = True
    print(json.dumps(out))


if __name__ == "__main__":
    main()
