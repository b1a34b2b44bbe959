#!/usr/bin/env python3
"""BMU launch time without host overhead (run on the GPU box: `python tools/bmu_bench.py`).
Each shape is timed twice: N eager calls between two events (what a Python caller sees) and the
same N launches replayed from one HIP graph (GPU time per launch).  QARIG_BMU_COARSE=0 (option bmu_coarse) selects
the streamed-codebook kernels for an A/B."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import ops  # noqa: E402

SHAPES = [  # (name, latent shape, patch, K)
    ("bench side measure 65536 x 512 x 16", (64, 4, 64, 64), (2, 2), 512),
    ("c2 HR tokens       16384 x 512 x 16", (64, 4, 32, 32), (2, 2), 512),
    ("c4 HR shard         8192 x 512 x 16", (8, 4, 64, 64), (2, 2), 512),
    ("c5 HR shard         8192 x 8192 x 4", (2, 4, 64, 64), (1, 1), 8192),
    ("c5 LR shard         2048 x 512 x 16", (2, 4, 64, 64), (2, 2), 512),
    ("patch 4            16384 x 512 x 64", (64, 4, 64, 64), (4, 4), 512),
    ("conditional LR        64 x 512 x 4096", (64, 4, 32, 32), (32, 32), 512),
]


def main():
    if os.environ.get("K_SWEEP"):   # time against codebook size: slope = main loop, intercept = fixed cost
        del SHAPES[:]
        for k in (64, 256, 512, 768):
            SHAPES.append((f"65536 rows x {k} x 16", (64, 4, 64, 64), (2, 2), k))
        for k in (64, 256, 512, 768):
            SHAPES.append((f"16384 rows x {k} x 16", (64, 4, 32, 32), (2, 2), k))
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    reps = 20
    for name, shape, patch, K in SHAPES:
        x = torch.tanh(torch.randn(shape, generator=g)).to(dev)
        w = torch.tanh(torch.randn((K, shape[1] * patch[0] * patch[1]), generator=g)).to(dev)
        if os.environ.get("BMU_FROZEN", "1") != "0":     # a frozen nn.Parameter, as models/Codebook.py passes it:
            w = torch.nn.Parameter(w, requires_grad=False)   # large launches take its prepared image (ops.bmu)
        for _ in range(3):
            ops.bmu(x, w, patch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.bmu(x, w, patch)
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / reps * 1e3
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ops.bmu(x, w, patch)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(reps):
                out = ops.bmu(x, w, patch)
        graph.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
        rows = out.numel()
        gpu = sorted(ts)[2]
        fl = rows * 2.0 * K * w.shape[1]
        print(f"{name:38s} eager {eager:7.1f} us/call   graph {gpu:7.1f} us/launch   "
              f"{fl / gpu / 1e6:6.1f} TF   {rows * (4 * w.shape[1] + 8) / gpu / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()


def phases():
    """Where a block of the coarse-pass kernel spends its time (clock64() cycles at ~100 MHz / shader clock,
    summed over blocks by the kernel itself when the counter buffer is given)."""
    from qarig import ops
    ops.BMU_COARSE_PHASES = True
    g = torch.Generator().manual_seed(9)
    for rows_shape in ((64, 4, 64, 64), (16, 4, 64, 64), (8, 4, 64, 64)):
        x = torch.tanh(torch.randn(rows_shape, generator=g)).cuda()
        w = torch.tanh(torch.randn((512, 16), generator=g)).cuda()
        for prepared in (False, True):
            ops.bmu_coarse(x, w, (2, 2), prepared=prepared)
            torch.cuda.synchronize()
            _, c = ops.bmu_coarse(x, w, (2, 2), prepared=prepared)
            c = c.cpu().tolist()
            nb = max(1, c[4])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gph = torch.cuda.CUDAGraph()
            img = ops.bmu_image(w) if prepared else None
            with torch.cuda.graph(gph):
                for _ in range(20):
                    ops.bmu_coarse(x, w, (2, 2), prepared=prepared)
            gph.replay()
            e0.record()
            gph.replay()
            e1.record()
            torch.cuda.synchronize()
            print(f"coarse {rows_shape[0] * 1024:6d} rows prepared={int(prepared)}: {e0.elapsed_time(e1) * 50:6.2f} us/launch, "
                  f"re-scanned rows {c[0]}, per block cycles: stage+gather {c[1] / nb:7.0f}  scan {c[2] / nb:7.0f}  "
                  f"finish {c[3] / nb:7.0f}  (blocks {nb})")


if __name__ == "__main__" and os.environ.get("BMU_PHASES") == "1":
    phases()
