#!/usr/bin/env python3
"""Fixed cost of a reduced-precision GEMM launch: time against K at fixed M, N (run on the GPU box).
Intercept = launch ramp + prologue + epilogue + tail, slope = the k-loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch
from qarig import ops
M = int(os.environ.get("ROWS", "32768"))
for N in (2048, 512):
    for mode in ("f32C", "bf16C"):
        row = []
        for K in (64, 128, 256, 512, 1024, 2048):
            A = torch.randn((M, K), device="cuda").bfloat16()
            B = torch.randn((N, K), device="cuda").bfloat16()
            C = torch.empty((M, N), device="cuda") if mode == "f32C" else None
            Cb = torch.empty((M, N), device="cuda", dtype=torch.bfloat16) if mode == "bf16C" else None
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ops.gemm_lp(A, B, 0, M, N, K, C=C, Cb=Cb)
                e0.record()
                for _ in range(10):
                    ops.gemm_lp(A, B, 0, M, N, K, C=C, Cb=Cb)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            row.append(f"K={K}: {sorted(ts)[2]:6.1f}")
        print(f"NT M={M} N={N} {mode}:  " + "  ".join(row), "us")
