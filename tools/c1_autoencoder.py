#!/usr/bin/env python3
"""BASELINE config 1 (plumbing): autoencoder reconstruction train step on 64x64x3 images,
batch 4, README channel widths.
  python tools/c1_autoencoder.py --cpu   torch-CPU oracle (oracle/ref_models.py): recon + MSE +
                                         backward + Adam; runs without a GPU
  python tools/c1_autoencoder.py         the HIP path on cuda:0, same step and weights, and the
                                         reconstruction checked against the oracle (<= 1e-5)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402

N, HW = 4, 64


def weights():
    """Default torch init of the README autoencoder (seed 3), as a state dict."""
    from models.Autoencoder import Autoencoder
    torch.manual_seed(3)
    m = Autoencoder(num_layers=2, image_channel=3, min_channel=256, max_channel=512, latent_channel=4)
    return m, {k: v.detach().clone() for k, v in m.state_dict().items()}


def cpu(steps=3):
    from oracle import ref_models as rm
    _, sd = weights()
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    names = list(sd)
    m1 = [torch.zeros_like(sd[k]) for k in names]
    m2 = [torch.zeros_like(sd[k]) for k in names]
    x = torch.rand((N, 3, HW, HW), generator=torch.Generator().manual_seed(0)) * 2 - 1
    times = []
    for step in range(1, steps + 2):
        t0 = time.perf_counter()
        recon = rm.autoencoder(sd, x)
        loss = torch.mean((recon - x) ** 2)
        grads = torch.autograd.grad(loss, [sd[k] for k in names])
        with torch.no_grad():
            rm.adam_step([sd[k] for k in names], grads, m1, m2, step, 1e-4)
        times.append(time.perf_counter() - t0)
    dt = sorted(times[1:])[len(times[1:]) // 2]
    return {"config": "C1 autoencoder train step 64x64x3 batch 4, torch-CPU oracle",
            "threads": torch.get_num_threads(), "seconds_per_step": round(dt, 4),
            "images_per_s": round(N / dt, 2), "loss": round(float(loss), 6)}


def gpu(steps=20):
    from oracle import ref_models as rm
    from qarig import functional as QF
    from qarig.optim import FlatAdam
    m, sd = weights()
    x = torch.rand((N, 3, HW, HW), generator=torch.Generator().manual_seed(0)) * 2 - 1
    want = rm.autoencoder(sd, x)
    m = m.cuda()
    xg = x.cuda()
    with torch.no_grad():
        got = m(xg).cpu()
    err = float((got - want).abs().max() / want.abs().max())
    assert err <= 1e-5, err
    opt = FlatAdam(m.parameters(), lr=1e-4, betas=(0.5, 0.999))

    def step():
        opt.zero_grad()
        loss = QF.mse_loss(m(xg), xg)
        loss.backward()
        opt.step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"config": "C1 autoencoder train step 64x64x3 batch 4, HIP path on one MI355X",
            "recon_rel_err_vs_oracle": err, "seconds_per_step": round(dt, 5),
            "images_per_s": round(N / dt, 1), "loss": round(float(loss), 6)}


if __name__ == "__main__":
    print(json.dumps(cpu() if "--cpu" in sys.argv else gpu()))
