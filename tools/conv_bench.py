#!/usr/bin/env python3
"""Throughput of the conv autoencoder kernels on the README shapes (256/512 channels,
128x128 images <-> 32x32x4 latents): encoder / decoder images/s and per-layer TFLOP/s.
Run on the GPU box: python tools/conv_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import _lib  # noqa: E402
from models.FC_Decoder import FC_Decoder  # noqa: E402
from models.FC_Encoder import FC_Encoder  # noqa: E402
from qarig import ops  # noqa: E402


def timeit(fn, reps=5, warm=6):
    for _ in range(warm):     # the first calls of a shape load code objects and fill the inference weight-copy cache
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    dec = FC_Decoder(num_layers=2, image_channel=3, min_channel=256, max_channel=512,
                     latent_channel=4).to(dev).eval()
    enc = FC_Encoder(num_layers=2, image_channel=3, min_channel=256, max_channel=512,
                     latent_channel=4).to(dev).eval()
    with torch.no_grad():
        for N in (4, 32):
            z = torch.randn(N, 4, 32, 32, device=dev)
            x = torch.randn(N, 3, 128, 128, device=dev)
            for name, m, inp, gf in (("decoder", dec, z, 27.64), ("encoder", enc, x, 53.41)):
                dt = timeit(lambda: m(inp))
                print(f"{name} N={N}: {N / dt:8.1f} img/s  {gf * N / dt / 1e3:6.1f} TF")
        N = 16
        for cin, cout, hw, s in ((512, 512, 32, 1), (256, 256, 128, 1), (256, 512, 128, 2),
                                 (3, 256, 128, 1), (256, 3, 128, 1)):
            x = torch.randn(N, cin, hw, hw, device=dev)
            w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
            b = torch.zeros(cout, device=dev)
            dt = timeit(lambda: ops.conv2d_fwd(x, w, b, s, 1, 1))
            ho = hw // s
            print(f"conv {cin}->{cout} @{hw} s{s}: {dt * 1e3:7.2f} ms "
                  f"{2.0 * N * cout * cin * 9 * ho * ho / dt / 1e12:6.1f} TF")
        x = torch.randn(N, 512, 32, 32, device=dev)
        w = torch.randn(512, 256, 4, 4, device=dev) * 0.02
        b = torch.zeros(256, device=dev)
        dt = timeit(lambda: ops.conv_transpose2d_fwd(x, w, b, 1))
        print(f"convT 512->256 @32: {dt * 1e3:7.2f} ms {2.0 * N * 256 * 512 * 16 * 32 * 32 / dt / 1e12:6.1f} TF")


def train():
    """Autoencoder training step (train_autoencoder.py loop body: recon, MSE, backward, Adam)
    at the README config, batch 16 of 128x128 images: images/s and TFLOP/s (3x forward)."""
    from models.Autoencoder import Autoencoder
    from qarig import functional as QF
    from qarig.optim import FlatAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = Autoencoder(num_layers=2, image_channel=3, min_channel=256, max_channel=512,
                    latent_channel=4).to(dev)
    opt = FlatAdam(m.parameters(), lr=1e-4, betas=(0.5, 0.999))
    N = int(os.environ.get("AE_BATCH", "16"))
    x = torch.rand(N, 3, 128, 128, device=dev) * 2 - 1

    def step():
        opt.zero_grad()
        loss = QF.mse_loss(m(x), x)
        loss.backward()
        opt.step()
    dt = timeit(step, reps=5)
    print(f"autoencoder train N={N}: {N / dt:8.1f} img/s  {3 * (27.64 + 53.41) * N / dt / 1e3:6.1f} TF "
          f"({dt * 1e3:.1f} ms/step)")


def ab():
    """Ring kernel on / off, interleaved in one process (the clock state of the box moves run to run by more
    than the difference): median of 7 alternating rounds per case."""
    import statistics
    from models.Autoencoder import Autoencoder
    from qarig import functional as QF
    from qarig.optim import FlatAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cases = []
    N = 16
    for cin, cout, hw in ((512, 512, 32), (256, 256, 128), (512, 512, 64)):
        x = torch.randn(N, cin, hw, hw, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
        b = torch.zeros(cout, device=dev)
        fl = 2.0 * N * cout * cin * 9 * hw * hw
        cases.append((f"conv {cin}->{cout} @{hw} fwd", (lambda x=x, w=w, b=b: ops.conv2d_fwd(x, w, b, 1, 1, 1)), fl, "TF"))
    for cin, cout, hw in ((256, 512, 128), (512, 512, 64)):
        x = torch.randn(N, cin, hw, hw, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
        b = torch.zeros(cout, device=dev)
        fl = 2.0 * N * cout * cin * 9 * (hw // 2) * (hw // 2)
        cases.append((f"conv {cin}->{cout} @{hw} stride 2 fwd", (lambda x=x, w=w, b=b: ops.conv2d_fwd(x, w, b, 2, 1, 1)), fl, "TF"))
    for cin, cout, hw in ((512, 256, 32), (256, 256, 64)):
        x = torch.randn(N, cin, hw, hw, device=dev)
        w = torch.randn(cin, cout, 4, 4, device=dev) * 0.02
        b = torch.zeros(cout, device=dev)
        fl = 2.0 * N * cout * cin * 16 * hw * hw
        cases.append((f"convT {cin}->{cout} @{hw} fwd", (lambda x=x, w=w, b=b: ops.conv_transpose2d_fwd(x, w, b, 1)), fl, "TF"))
    m = Autoencoder(num_layers=2, image_channel=3, min_channel=256, max_channel=512, latent_channel=4).to(dev)
    opt = FlatAdam(m.parameters(), lr=1e-4, betas=(0.5, 0.999))
    xi = torch.rand(N, 3, 128, 128, device=dev) * 2 - 1

    def step():
        opt.zero_grad()
        QF.mse_loss(m(xi), xi).backward()
        opt.step()
    cases.append(("autoencoder train step N=16", step, 3 * (27.64 + 53.41) * N * 1e9, "TF"))
    with torch.no_grad():
        enc, dec = m.fc_encoder, m.fc_decoder
        z = torch.randn(32, 4, 32, 32, device=dev)
        x32 = torch.randn(32, 3, 128, 128, device=dev)
        cases.append(("decoder N=32", (lambda: dec(z)), 27.64e9 * 32, "TF"))
        cases.append(("encoder N=32", (lambda: enc(x32)), 53.41e9 * 32, "TF"))
    for name, fn, fl, unit in cases:
        res = {"1": [], "0": []}
        grad = torch.enable_grad() if "train" in name else torch.no_grad()
        with grad:
            for rnd in range(7):
                for mode in ("1", "0"):
                    _lib.set_option("conv_ring", int(mode))
                    res[mode].append(timeit(fn, reps=3))
        a, b_ = statistics.median(res["1"]), statistics.median(res["0"])
        print(f"{name:32s} ring {a * 1e3:8.3f} ms {fl / a / 1e12:6.1f} {unit}   gather kernel {b_ * 1e3:8.3f} ms "
              f"{fl / b_ / 1e12:6.1f} {unit}   ({b_ / a:.3f}x)", flush=True)
    _lib.set_option("conv_ring", 1)


if __name__ == "__main__":
    if "--ab" in sys.argv:
        ab()
    elif "--train" in sys.argv:
        train()
    else:
        main()
