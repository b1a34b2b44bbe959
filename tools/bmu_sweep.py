#!/usr/bin/env python3
"""BMU tokenisation throughput over the cascade's patch sizes (SURVEY 8d, config 2 sweep):
batch 64 latents (64,4,32,32), K = 512, patch 32/8/4/2/1, plus the config 4/5 shape
(64,4,64,64) with patch 1 and K = 8192.  Algorithmic bytes = 4*D per row read + 8 written
(+ codebook once); flops = 2*K*(D+2) per row.   python tools/bmu_sweep.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import ops  # noqa: E402


def main():
    g = torch.Generator().manual_seed(1)
    rows = []
    for (N, H, p, K) in ((64, 32, 32, 512), (64, 32, 8, 512), (64, 32, 4, 512), (64, 32, 2, 512),
                         (64, 32, 1, 512), (64, 64, 2, 512), (64, 64, 1, 8192)):
        x = torch.tanh(torch.randn((N, 4, H, H), generator=g)).cuda()
        D = 4 * p * p
        w = torch.tanh(torch.randn((K, D), generator=g)).cuda()
        for _ in range(5):
            ops.bmu(x, w, (p, p))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            ops.bmu(x, w, (p, p))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        R = N * (H // p) ** 2
        byt = R * (4 * D + 8) + 4 * K * D
        fl = 2.0 * R * K * (D + 2)
        rows.append({"latent": f"{N}x4x{H}x{H}", "patch": p, "D": D, "K": K, "rows": R,
                     "us": round(ms * 1e3, 1), "rows_per_s": round(R / ms * 1e3),
                     "algorithmic_GBps": round(byt / ms / 1e6, 1), "TFLOPs": round(fl / ms / 1e9, 1)})
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
