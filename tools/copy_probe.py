#!/usr/bin/env python3
"""Which host-side calls become `__amd_rocclr_copyBuffer` launches?  Run under rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/probe -- python3 tools/copy_probe.py
Each candidate runs REPS times between two marker kernels whose names are unique in the trace
(torch.cumsum / torch.flip are used nowhere else), so the kernel trace can be cut per candidate."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import functional as QF, ops  # noqa: E402
from qarig.optim import FlatAdam  # noqa: E402

REPS = 50
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
marker = torch.arange(4096, device=dev, dtype=torch.float32)
names = []


def section(name, fn):
    torch.cuda.synchronize()
    torch.cumsum(marker, 0)          # section start marker
    for _ in range(REPS):
        fn()
    torch.flip(marker, (0,))         # section end marker
    torch.cuda.synchronize()
    names.append(name)


x = torch.randn((2048, 512), device=dev, generator=g)
w = torch.nn.Parameter(torch.randn((2048, 512), device=dev, generator=g) * 0.02)
b = torch.nn.Parameter(torch.zeros(2048, device=dev))
w2 = torch.nn.Parameter(torch.randn((512, 2048), device=dev, generator=g) * 0.02)
b2 = torch.nn.Parameter(torch.zeros(512, device=dev))
opt = FlatAdam([w, b, w2, b2], lr=1e-3)
xr = x.clone().requires_grad_(True)
section("gemm plain", lambda: ops.gemm(x, w))
section("gemm splitk4 + reduce", lambda: ops.gemm(w2.detach().t().contiguous().t() if False else x, w, splitk=4))
section("gemm bias act preact", lambda: ops.gemm(x, w, bias=b, want_preact=True, act=1))
section("layernorm fwd", lambda: ops.layernorm_fwd(x))
section("linear_act fwd only", lambda: QF.linear_act(xr, w, b, act=1))


def fb():
    y = QF.linear_act(xr, w, b, act=1)
    y.backward(torch.ones_like(y))


section("linear_act fwd+bwd (ones_like included)", fb)
gy = torch.ones((2048, 2048), device=dev)


def fb2():
    y = QF.linear_act(xr, w, b, act=1)
    y.backward(gy)


section("linear_act fwd+bwd (given grad)", fb2)
gy2 = torch.ones((2048, 512), device=dev)


def fb3():
    y = QF.mlp2(xr, w, b, w2, b2, 1, 0)
    y.backward(gy2)


section("mlp2 fwd+bwd", fb3)
section("zero_grad", lambda: opt.zero_grad())
section("adam step", lambda: opt.step())
section("torch.empty", lambda: torch.empty((2048, 512), device=dev))
cpu_idx = torch.randint(0, 10, (8,))
section("small H2D .to(device)", lambda: cpu_idx.to(dev))
section("pinned non_blocking H2D", lambda: cpu_idx.pin_memory().to(dev, non_blocking=True))
print("SECTIONS", "|".join(names))
