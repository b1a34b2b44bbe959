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


class DrawTape:
    """Sampling draws of one generation run replayed into another.  `record(fn)` runs fn with
    torch.multinomial wrapped and keeps every probability matrix it sampled from and the tokens it drew (one
    segment per call); `replay(i, fn, sampler)` runs fn with segment i's tokens injected -- the fused sampler's
    draws through qarig.sampling.FUSED_DEBUG (forced tokens, probability log), every torch.multinomial call of the
    run through the patched function -- and requires the probabilities at EVERY draw to equal the recorded ones
    within `tol`, and the run to make exactly the recorded number of draws."""

    def __init__(self, monkeypatch, tol=1e-5):
        self.mp, self.tol, self.segments = monkeypatch, tol, []
        self.real = torch.multinomial
        self.worst = 0.0
        self.fused_draws = 0

    def add_segment(self, probs, tokens):
        self.segments.append([(p, t.reshape(-1)) for p, t in zip(probs, tokens)])

    def record(self, fn):
        seg = []

        def rec(probs, num_samples, *a, **k):
            out = self.real(probs, num_samples, *a, **k)
            seg.append((probs.detach().clone(), out.detach().clone().reshape(-1)))
            return out

        self.mp.setattr(torch, "multinomial", rec)
        try:
            r = fn()
        finally:
            self.mp.setattr(torch, "multinomial", self.real)
        self.segments.append(seg)
        return r

    def replay(self, i, fn, sampler="fused"):
        from qarig import sampling
        seg = self.segments[i]
        state = {"d": 0}
        if sampler == "fused":
            sampling.FUSED_DEBUG = {"forced": torch.stack([t.cpu() for _, t in seg]), "log": True}

        def inj(probs, num_samples, *a, **k):
            d = (sampling.FUSED_DEBUG or {}).get("draws", 0) + state["d"]
            assert num_samples == 1 and d < len(seg), "more draws than were recorded"
            want, tok = seg[d]
            err = float((probs.detach().cpu() - want.cpu()).abs().max())
            self.worst = max(self.worst, err)
            assert err < self.tol, f"draw {d}: probabilities differ from the recorded ones by {err}"
            state["d"] += 1
            return tok.to(probs.device)[:, None]

        self.mp.setattr(torch, "multinomial", inj)
        try:
            r = fn()
        finally:
            dbg, sampling.FUSED_DEBUG = sampling.FUSED_DEBUG, None
            self.mp.setattr(torch, "multinomial", self.real)
        fused = int((dbg or {}).get("draws", 0))
        if fused:
            got = dbg["probs"][:fused].cpu()
            want = torch.stack([p.cpu() for p, _ in seg[:fused]])
            err = float((got - want).abs().max())
            self.worst = max(self.worst, err)
            assert err < self.tol, f"fused sampler: probabilities differ from the recorded ones by {err}"
        assert fused + state["d"] == len(seg), f"{fused} + {state['d']} draws made, {len(seg)} recorded"
        self.fused_draws = fused
        return r
