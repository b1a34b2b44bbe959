#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM on the shapes of the bench step (run on the GPU
box: `python tools/gemm_bench.py`).  Interleaved rounds in one process, HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import _lib, ops  # noqa: E402

M = int(os.environ.get("ROWS", "16384"))   # ROWS=2048: the 8-sequence shard of BASELINE config 4
SHAPES = [  # (name, M, N, K, a_kcontig, b_kcontig, kwargs)
    ("fwd 512->2048 +bias+silu+preact", M, 2048, 512, True, True, dict(bias=True, act=1, pre=True)),
    ("fwd 2048->512 +bias", M, 512, 2048, True, True, dict(bias=True)),
    ("fwd 512->512 +bias", M, 512, 512, True, True, dict(bias=True)),
    ("fwd 512->2048 plain", M, 2048, 512, True, True, {}),
    ("dX  [M,2048]@[2048,512]", M, 512, 2048, True, False, {}),
    ("dH  [M,512]@[512,2048] *act'", M, 2048, 512, True, False, dict(gradz=True)),
    ("dH  plain (no act' fusion)", M, 2048, 512, True, False, {}),
    ("dW  2048x512 over M (splitk)", 2048, 512, M, False, False, dict(splitk=-1)),
    ("dW  512x2048 over M (splitk)", 512, 2048, M, False, False, dict(splitk=-1)),
    ("dW  512x512 over M (splitk)", 512, 512, M, False, False, dict(splitk=-1)),
    # what batching the 42 cond projections of a step into one launch would run as
    ("dX  512->512 [M,512]@[512,512]", M, 512, 512, True, False, {}),
    ("fwd cond-all 512->21504 +bias", M, 21504, 512, True, True, dict(bias=True)),
    ("dX  cond-all [M,21504]@[21504,512]", M, 512, 21504, True, False, {}),
    ("dW  cond-all 21504x512 over M", 21504, 512, M, False, False, dict(splitk=-1)),
]


def main():
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    results = {}
    cases = []
    for name, m, n, k, ak, bk, kw in SHAPES:
        A = torch.randn((m, k) if ak else (k, m), device=dev, generator=g)
        B = torch.randn((n, k) if bk else (k, n), device=dev, generator=g)
        bias = torch.randn(n, device=dev, generator=g) if kw.get("bias") else None
        gz = torch.randn((m, n), device=dev, generator=g) if kw.get("gradz") else None
        sk = kw.get("splitk", 1)
        if sk == -1:
            sk = ops.pick_splitk(m, n, k)
        cases.append((name, m, n, k, dict(A=A, B=B, a_kcontig=ak, b_kcontig=bk, bias=bias,
                                          want_preact=kw.get("pre", False), act=kw.get("act", 0),
                                          gradz=gz, gact=1 if gz is not None else 0, splitk=sk)))
    # SWEEP="option=v0,v1,...": A/B of a kernel-selection option (qarig_set_option: gemm_dma, gemm_pair),
    # interleaved rounds in one process
    sweep_env, sweep = "gemm_pair", []
    if os.environ.get("SWEEP"):
        sweep_env, vals = os.environ["SWEEP"].split("=")
        sweep = vals.split(",")
    if sweep:
        res = {}
        for rnd in range(5):
            for st in sweep:
                _lib.set_option(sweep_env, int(st))
                for name, m, n, k, kw in cases[:10]:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ops.gemm(**kw)
                    e0.record()
                    for _ in range(10):
                        ops.gemm(**kw)
                    e1.record()
                    torch.cuda.synchronize()
                    res.setdefault((name, st), []).append(2.0 * m * n * k / (e0.elapsed_time(e1) / 10) / 1e9)
        for name, m, n, k, kw in cases[:10]:
            print(f"{name:36s} " + "  ".join(f"{sweep_env[-8:]}={st}: {sorted(res[(name, st)])[2]:6.1f}" for st in sweep))
        return
    if os.environ.get("SPLITS"):
        # SPLITS=0,1,2,4,8: time each shape (reduce pass included) per reduction split; 0 = the
        # library's own choice (ops.auto_splitk for the forward shapes, pick_splitk for dW)
        splits = [int(v) for v in os.environ["SPLITS"].split(",")]
        for name, m, n, k, kw in cases[:10]:
            line = []
            for sk in splits:
                kw2 = dict(kw)
                kw2["splitk"] = (kw["splitk"] if kw["splitk"] > 1 else None) if sk == 0 else sk
                if sk > 1 and (k % sk or (k // sk) % 16):
                    continue
                ts = []
                for rnd in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ops.gemm(**kw2)
                    e0.record()
                    for _ in range(10):
                        ops.gemm(**kw2)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 10)
                us = sorted(ts)[2] * 1e3
                line.append(f"sk={sk}: {us:6.1f} us {2.0 * m * n * k / us / 1e6:6.1f} TF")
            print(f"{name:34s} M={m:5d} N={n:5d} K={k:5d}  " + " | ".join(line))
        return
    for rnd in range(4):
        for name, m, n, k, kw in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            ops.gemm(**kw)
            e0.record()
            for _ in range(reps):
                ops.gemm(**kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            results.setdefault(name, []).append(2.0 * m * n * k / ms / 1e9)
    for name, m, n, k, kw in cases:
        r = sorted(results[name])
        print(f"{name:36s} M={m:6d} N={n:5d} K={k:6d} splitk={kw['splitk']:2d}  "
              f"median {r[len(r) // 2]:6.1f} TF  best {r[-1]:6.1f} TF")


if __name__ == "__main__":
    main()
