#!/usr/bin/env python3
"""Weight-gradient (TN) products of the reduced-precision mode against the K split
(run on the GPU box; QARIG_LP_BIG=0/1 selects the 128 / 256 tile kernel where both apply)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch
from qarig import ops
K = int(os.environ.get("ROWS", "32768"))
for (M, N) in ((2048, 512), (512, 2048), (512, 512), (2048, 2048)):
    A = torch.randn((K, M), device="cuda").bfloat16()
    B = torch.randn((K, N), device="cuda").bfloat16()
    C = torch.zeros((M, N), device="cuda")
    row = []
    for sk in (4, 8, 16, 32, 64):
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.gemm_lp(A, B, 1, M, N, K, C=C, splitk=sk, accumulate=True)
            e0.record()
            for _ in range(10):
                ops.gemm_lp(A, B, 1, M, N, K, C=C, splitk=sk, accumulate=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        us = sorted(ts)[2]
        row.append(f"sk={sk}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF")
    print(f"TN dW {M}x{N} over K={K}:  " + " | ".join(row))
