import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "quantized-autoregression-image-generator_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is present, so a plain
    # `pytest tests/` on a CPU box stays green; `-m gpu` on the GPU box runs them.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """npz fixture -> nested dict ('sd/key' -> d['sd']['key']) of torch tensors."""
    raw = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in raw.files:
        v = torch.from_numpy(np.array(raw[k]))
        if "/" in k:
            a, b = k.split("/", 1)
            out.setdefault(a, {})[b] = v
        else:
            out[k] = v
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max|a-b| / max|b|  (the tolerance definition of SURVEY.md 8d)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def grad_err(a, b, floor=1e-6):
    """rel_err with an absolute floor on the denominator: some gradients are exactly
    zero mathematically (e.g. the key bias under softmax) and pure rounding noise
    numerically."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(floor))
