#!/usr/bin/env python3
"""Which ATen ops still run inside one training step, and which line of this repo asked for them
(run on the GPU box):  python tools/dispatch_trace.py [--config c4]

A TorchDispatchMode sees every aten call of the step -- the ones autograd's engine issues in
backward included (single-threaded backward) -- and records the innermost frames of this repository
on the Python stack.  Ops that allocate or only re-view (empty, view, as_strided, ...) are listed
separately: they launch nothing."""
import os
import sys
import traceback
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402

NO_LAUNCH = ("empty", "view", "as_strided", "reshape", "_unsafe_view", "detach", "alias", "t.", "transpose",
             "unbind", "select", "slice", "expand", "permute", "squeeze", "unsqueeze", "_reshape_alias",
             "empty_like", "empty_strided", "new_empty", "split", "narrow", "is_", "_local_scalar", "lift_fresh",
             "resize_", "set_", "record_stream", "stride", "size", "sym_", "result_type", "_has_compatible")


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.launch, self.free = Counter(), Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        frames = [f for f in traceback.extract_stack()[:-1]
                  if ROOT in f.filename and "dispatch_trace" not in f.filename]
        where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:][::-1]) or "(autograd engine)"
        if name.startswith(("add.", "add_.")):
            node = torch._C._current_autograd_node()
            where += f"  [while running {node.name() if node is not None else None}]"
        extra = ""
        if name.startswith(("copy_", "add_", "add.", "clone", "_to_copy", "fill_", "zero_", "cat", "stack")) and args:
            t = args[0] if isinstance(args[0], torch.Tensor) else (args[0][0] if args[0] else None)
            if isinstance(t, torch.Tensor):
                extra = f" {tuple(t.shape)}"
        (self.free if name.startswith(NO_LAUNCH) else self.launch)[(name + extra, where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    sys.argv = ["bench.py", "--steps", "1", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-events"] + sys.argv[1:]
    torch.autograd.set_multithreading_enabled(False)
    tr = Trace()
    state = {"n": 0}
    orig_zero = None
    from qarig import pipeline
    orig = pipeline.train_step

    def traced(*a, **k):
        state["n"] += 1
        if state["n"] == 3:                      # the timed step, after the two warm-up steps
            with tr:
                return orig(*a, **k)
        return orig(*a, **k)

    pipeline.train_step = traced
    bench.main()
    print(f"--- aten ops that launch device work in ONE train_step: {sum(tr.launch.values())}")
    for (name, where), n in sorted(tr.launch.items(), key=lambda kv: -kv[1])[:60]:
        print(f"{n:5d} x {name:44s} {where}")
    print(f"--- allocation / view ops (no launch): {sum(tr.free.values())}")
    for (name, where), n in sorted(tr.free.items(), key=lambda kv: -kv[1])[:12]:
        print(f"{n:5d} x {name:44s} {where}")


if __name__ == "__main__":
    main()
