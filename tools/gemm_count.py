#!/usr/bin/env python3
"""Instruction-count probe: a few single-shape GEMM launches for tools/pmc_one.sh
(kernel name + grid identify the shape in its output)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch
from qarig import ops
g = torch.Generator(device="cuda").manual_seed(0)
M = 16384
for (N, K, kw) in ((2048, 512, {}), (512, 2048, {}), (1024, 1024, {}), (2048, 512, dict(bias=True, act=1, want_preact=True))):
    A = torch.randn((M, K), device="cuda", generator=g)
    B = torch.randn((N, K), device="cuda", generator=g)
    if kw.get("bias"):
        kw = dict(kw, bias=torch.randn(N, device="cuda", generator=g))
    for _ in range(3):
        ops.gemm(A, B, **kw)
torch.cuda.synchronize()
