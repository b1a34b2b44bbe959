"""Gaussian-neighbourhood quantise (reference Codebook.py:112-130) forward + codebook gradient:
the band form (gather of a K x D table) against the literal rows x K matrix form (QARIG_SOM_DENSE=1)."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "quantized-autoregression-image-generator_amd"))
from models.Codebook import _SomQuantize  # noqa: E402


def run(K, D, R, rng, dense):
    os.environ["QARIG_SOM_DENSE"] = "1" if dense else "0"
    g = torch.Generator().manual_seed(1)
    w = ((torch.rand((K, D), generator=g) * 2 - 1) / K).cuda().requires_grad_(True)
    bmu = torch.randint(0, K, (R,), generator=g).cuda()
    dq = torch.randn((R, D), generator=g).cuda()
    two_var = 2 * -(rng / (2 * math.log(0.1)))

    def step():
        w.grad = None
        q = _SomQuantize.apply(w, bmu, two_var)
        q.backward(dq)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def piece(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    print(f"   {name}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us", flush=True)


from qarig import ops  # noqa: E402

for K, D, R in ((512, 16, 65536), (8192, 4, 32768)):
    g = torch.Generator().manual_seed(1)
    w = ((torch.rand((K, D), generator=g) * 2 - 1) / K).cuda()
    bmu = torch.randint(0, K, (R,), generator=g).cuda()
    dq = torch.randn((R, D), generator=g).cuda()
    print(f"K={K} D={D} rows={R}")
    piece("som_band (range 256)", lambda: ops.som_band(w, 111.2))
    piece("gather_rows", lambda: ops.gather_rows(bmu, w))
    piece("embedding_bwd", lambda: ops.embedding_bwd(bmu, dq, K))

for K, D, R, rng in ((512, 16, 65536, 256), (512, 16, 65536, 1.0), (8192, 4, 32768, 4096), (8192, 4, 32768, 1.0)):
    b = run(K, D, R, rng, False)
    d = run(K, D, R, rng, True)
    print(f"K={K} D={D} rows={R} range={rng}: band {b:.1f} us, rows x K matrix {d:.1f} us (fwd+bwd)", flush=True)
