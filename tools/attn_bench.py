#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels (run on the GPU box).  Shapes: BASELINE config 2
(64 x 256 tokens, 64 heads of 8) and config 5 (2 x 4096).  Sweeps the workgroup geometry
(QARIG_ATTN_QW / QARIG_ATTN_BW); interleaved rounds in one process, HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import _lib  # noqa: E402
from qarig import ops  # noqa: E402


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, N, S, H, d in (("c2 64x256", 64, 256, 64, 8), ("c5 2x4096", 2, 4096, 64, 8)):
        q, k, v, do = (torch.randn((N, S, H * d), device="cuda", generator=g) for _ in range(4))
        pairs = N * H * S * (S + 1) / 2
        res = {}
        for rnd in range(3):
            for qw in (1, 2, 4):
                _lib.set_option("attn_qw", qw)
                o, lse = ops.attention_fwd(q, k, v, H, True)
                res.setdefault(("fwd", qw), []).append(timed(lambda: ops.attention_fwd(q, k, v, H, True)))
            for bw in (1, 2):
                _lib.set_option("attn_bw", bw)
                res.setdefault(("bwd", bw), []).append(
                    timed(lambda: ops.attention_bwd(q, k, v, o, do, lse, H, True)))
        for key in sorted(res):
            us = sorted(res[key])[1]
            fl = pairs * (4 * d if key[0] == "fwd" else 10 * d)
            print(f"{name:10s} {key[0]} W={key[1]}: {us:8.1f} us   {fl / us / 1e6:6.1f} TFLOP/s (causal pairs)")


if __name__ == "__main__":
    main()
