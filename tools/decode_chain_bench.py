#!/usr/bin/env python3
"""Where a decode step's time goes: a chain of DEPENDENT launches captured in a HIP graph, as
DecodeCache replays one per token -- per-launch time of
  - the two Linear shapes of a decoder block (512 -> 2048, 2048 -> 512) alternating, every launch on
    its OWN weights (cold: 96 x 2 x 4 MB = 768 MB, nothing survives in the 256 MB Infinity Cache) or
    all on the same two matrices (hot);
  - the 512 -> 512 residual layers;
  - a trivial elementwise kernel (the launch boundary alone).
    python tools/decode_chain_bench.py [--rows 4] [--opt decode_stream=0]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import ops, _lib  # noqa: E402
from qarig import functional as QF  # noqa: E402


def timed_graph(fn, reps=30):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--layers", type=int, default=96)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    for o in args.opt:
        k, v = o.split("=")
        _lib.load().qarig_set_option(k.encode(), int(v))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    M, L = args.rows, args.layers
    w1 = [torch.randn(2048, 512, device=dev) * 0.03 for _ in range(L)]
    w2 = [torch.randn(512, 2048, device=dev) * 0.03 for _ in range(L)]
    w3 = [torch.randn(512, 512, device=dev) * 0.03 for _ in range(L)]
    b1 = torch.randn(2048, device=dev)
    b2 = torch.randn(512, device=dev)
    sc, sh = torch.randn(512, device=dev), torch.randn(512, device=dev)
    x0 = torch.randn(M, 512, device=dev)
    out = {"rows": M, "layers": L, "options": args.opt}

    def mlp(cold, ln):
        def fn():
            x = x0
            for i in range(L):
                j = i if cold else 0
                if ln:
                    h = ops.decode_linear(x, w1[j], b1, act=1, scale=sc, shift=sh)
                else:
                    h = ops.gemm(x, w1[j], bias=b1, act=1)
                x = ops.gemm(h, w2[j], bias=b2, act=1)
            return x
        return fn

    def res(cold):
        def fn():
            x = x0
            for i in range(L):
                x = ops.gemm(x, w3[i if cold else 0], bias=b2, residual=x0, act=1)
            return x
        return fn

    def tiny():
        x = x0
        for i in range(2 * L):
            x = QF.mul(x, x0)
        return x

    for name, fn, n in (("mlp_cold", mlp(True, False), 2 * L), ("mlp_hot", mlp(False, False), 2 * L),
                        ("res_cold", res(True), L), ("res_hot", res(False), L), ("tiny", tiny, 2 * L)):
        out[name + "_us_per_launch"] = round(timed_graph(fn) / n * 1e6, 3)
    if args.rows <= 16 and not any(o.startswith("decode_stream=0") for o in args.opt):
        out["mlp_ln_cold_us_per_launch"] = round(timed_graph(mlp(True, True)) / (2 * L) * 1e6, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
