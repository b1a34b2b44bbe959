#!/usr/bin/env python3
"""Which CPU-side op launched each device memcpy of ONE training step (run on the GPU box):
    python tools/memcpy_trace.py [--config c4]
torch.profiler (CPU + device activities) around the third step; every Memcpy / Memset device event
is matched to the innermost CPU op or Python function whose time range contains its launch."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from qarig import pipeline  # noqa: E402

sys.argv = ["bench.py", "--steps", "1", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-events", "--eager"] + sys.argv[1:]
torch.autograd.set_multithreading_enabled(False)
state = {"n": 0, "prof": None}
orig = pipeline.train_step


def traced(*a, **k):
    state["n"] += 1
    if state["n"] == 3:
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                     record_shapes=True) as prof:
            out = orig(*a, **k)
            torch.cuda.synchronize()
        state["prof"] = prof
        return out
    return orig(*a, **k)


pipeline.train_step = traced
bench.main()
prof = state["prof"]
events = prof.events()
cpu = [e for e in events if e.device_type == torch.autograd.DeviceType.CPU]
dev = [e for e in events if e.device_type != torch.autograd.DeviceType.CPU]
names = Counter(e.name for e in dev)
print("device events by name (top 12):", names.most_common(12))
mem = [e for e in dev if "emcpy" in e.name or "emset" in e.name or "copyBuffer" in e.name]
print("memcpy / memset / copyBuffer device events in the step:", len(mem))
by = Counter()
for m in mem:
    # the CPU op with the same correlation / the innermost one containing the launch time
    cands = [c for c in cpu if c.time_range.start <= m.time_range.start and any(
        k.id == m.id or k.name == m.name for k in getattr(c, "kernels", []))]
    owner = None
    for c in cpu:
        for k in getattr(c, "kernels", []) or []:
            if k.name == m.name and abs(getattr(k, "duration", 0) - m.time_range.elapsed_us()) < 1e-3:
                owner = c
    if owner is None and cands:
        owner = min(cands, key=lambda c: c.time_range.elapsed_us())
    stack = ""
    if owner is not None and owner.stack:
        fr = [f for f in owner.stack if "dist-packages" not in f and "memcpy_trace" not in f][:3]
        stack = " <- ".join("/".join(f.split("/")[-2:]) for f in fr)
    shape = str(getattr(owner, "input_shapes", ""))[:60] if owner is not None else ""
    by[(m.name[:30], owner.name if owner is not None else "?", shape, stack)] += 1
for (mn, on, shp, st), n in by.most_common(25):
    print(f"{n:5d} x {mn:30s} by {on:28s} {shp:60s} {st}")
