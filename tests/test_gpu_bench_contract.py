"""The driver's contract for bench.py: one JSON line with the agreed keys, the BASELINE
metric string, the roofline and cpu_baseline objects; and __graft_entry__.smoke()."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
        "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_bench_line_contract():
    env = dict(os.environ, QARIG_CPU_BASELINE_SECONDS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert KEYS <= set(out), KEYS - set(out)
    want = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert out["metric"] == want
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "f32" and out["data"] == "synthetic" and "workload" in out["config"]
    assert "model" not in out["config"]
    assert abs(out["value"] - 64 * 256 / (out["ms_per_step"] / 1e3)) < 0.01 * out["value"]
    rf = out["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf)
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = out["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] in ("port", "reference")
    assert cb["value"] > 0 and cb["cores"] >= 1


def test_graft_entry_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.smoke()
