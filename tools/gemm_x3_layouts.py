#!/usr/bin/env python3
"""The three operand layouts of one product (16384 x 512 x 2048) through the gemm_x3 kernel, a few launches each
(for counter passes: tools/pmc_one.sh <tag> gemm_x3 "<counters>" tools/gemm_x3_layouts.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import _lib, ops  # noqa: E402

_lib.set_option("gemm_x3", 1)
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 16384, 512, 2048
for ak, bk in ((True, True), (True, False), (False, False), (False, True)):
    A = torch.randn((M, K) if ak else (K, M), device="cuda", generator=g)
    B = torch.randn((N, K) if bk else (K, N), device="cuda", generator=g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.gemm(A, B, a_kcontig=ak, b_kcontig=bk, splitk=1)
    e0.record()
    for _ in range(5):
        ops.gemm(A, B, a_kcontig=ak, b_kcontig=bk, splitk=1)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"A {'[M][K]' if ak else '[K][M]'}  B {'[N][K]' if bk else '[K][N]'}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF-eq")
