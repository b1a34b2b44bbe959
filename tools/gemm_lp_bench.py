#!/usr/bin/env python3
"""Micro-benchmark of the bf16-operand GEMM (csrc/gemm_lp.hip) on the shapes of the bench step
(run on the GPU box).  Kernel alone (operands already bf16 in HBM) and the fp32-tensor
contract of ops.gemm in bf16 mode (casts included), next to the fp32 kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import ops  # noqa: E402

M = 16384
SHAPES = [  # name, layout, M, N, K, splitk
    ("fwd 512->2048 NT", 0, M, 2048, 512, 1),
    ("fwd 2048->512 NT", 0, M, 512, 2048, 1),
    ("fwd 512->512 NT", 0, M, 512, 512, 1),
    ("classifier 2048->8192 NT", 0, M, 8192, 2048, 1),
    ("dW 2048x512 over M TN", 1, 2048, 512, M, 8),
    ("dW 512x2048 over M TN", 1, 512, 2048, M, 8),
    ("dW 512x512 over M TN", 1, 512, 512, M, 16),
]


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, layout, m, n, k, sk in SHAPES:
        a32 = torch.randn((m, k) if layout == 0 else (k, m), device="cuda", generator=g)
        b32 = torch.randn((n, k) if layout == 0 else (k, n), device="cuda", generator=g)
        A, B = a32.bfloat16(), b32.bfloat16()
        C = torch.empty((m, n), device="cuda")
        Cb = torch.empty((m, n), device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * m * n * k
        res = []
        res.append(("lp kernel -> fp32 C", timed(lambda: ops.gemm_lp(A, B, layout, m, n, k, C=C, splitk=sk))))
        if sk == 1:
            res.append(("lp kernel -> bf16 C only", timed(lambda: ops.gemm_lp(A, B, layout, m, n, k, Cb=Cb))))
        ops.PRECISION = "bf16"
        res.append(("ops.gemm bf16 mode (casts incl.)",
                    timed(lambda: ops.gemm(a32, b32, layout == 0, layout == 0, splitk=sk))))
        ops.PRECISION = "f32"
        res.append(("ops.gemm fp32", timed(lambda: ops.gemm(a32, b32, layout == 0, layout == 0,
                                                           splitk=ops.pick_splitk(m, n, k) if layout else 1))))
        print(f"{name:28s} M={m:6d} N={n:5d} K={k:6d}: " +
              "  |  ".join(f"{w}: {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF" for w, ms in res))


if __name__ == "__main__":
    main()
