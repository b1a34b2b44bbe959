#!/usr/bin/env python3
"""Do independent decode steps overlap?  G decode caches of `rows` rows each (README-size encoder-decoder stage,
window 256), each replaying its captured step graph on a stream of its own: ms per round of G steps against G
times the time of one.  The step is a chain of ~80 dependent launches of one wave per SIMD, so other chains have
the room; this is what generating several groups of images at once rests on (qarig/sampling.py).
    python tools/decode_concurrency_probe.py [--rows 16] [--steps 300]"""
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
from qarig import kvcache  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    K, B, S = 512, args.rows, 256
    model = build_stage_model(2, K, dev)
    positions = [0.0] + [float(i + 1) for i in range(1, S)]
    out = {"rows": B, "window": S, "steps": args.steps}
    with torch.no_grad():
        caches, streams = [], []
        for g in range(4):
            enc = model.encode(torch.randint(0, K, (B, 64), device=dev))
            c = kvcache.DecodeCache(model, enc, B, S, graph=True, positions=positions)
            ids = torch.randint(0, K, (B,), device=dev)
            for t in range(4):
                c.step(ids, None, t)
            c.ctl[0:1].fill_(S - 1)
            caches.append(c)
            streams.append(torch.cuda.Stream(device=dev))
        torch.cuda.synchronize()
        for G in (1, 2, 3, 4):
            for s in streams[:G]:
                s.wait_stream(torch.cuda.current_stream())
            t0 = time.perf_counter()
            for _ in range(args.steps):
                for g in range(G):
                    with torch.cuda.stream(streams[g]):
                        caches[g]._graph.replay()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            out[f"streams_{G}"] = {"ms_per_round": round(ms, 4), "ms_per_step_equivalent": round(ms / G, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
