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
        "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs"}


def test_bench_line_contract():
    env = dict(os.environ, QARIG_CPU_BASELINE_SECONDS="2")
    env.pop("QARIG_GEMM_X3", None)      # (the suite may run under the option; the line under test is the default one)
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
    # the other BASELINE configurations on the same line: cascade generation (config 3), the per-GPU shards of
    # configs 4 and 5 -- a few steps each, measured by child runs of this script
    side = out["configs"]
    assert set(side) == {"c3", "c4_shard", "c5_shard", "c2_gemm_x3", "c4_shard_gemm_x3"}
    for name, o in side.items():
        assert "error" not in o, (name, o)
        assert {"workload", "value", "unit", "ms_per_step", "dtype", "roofline"} <= set(o), name
        assert o["value"] > 0 and o["ms_per_step"] > 0
        assert {"bound", "achieved", "peak", "unit", "frac"} <= set(o["roofline"]), name
    c3 = side["c3"]
    assert {"sequential", "first_call", "sequential_one_by_one", "batched_beams", "decode_step_ms_rows4",
            "decode_step_ms_rows16",
            "decoder_images_per_s"} <= set(c3)
    assert c3["sequential"]["accepted_tokens_per_s"] == c3["value"] and c3["roofline"]["bound"] == "hbm"
    # the reference's draw order with the candidates as rows of one batch beats running them one after the other
    assert c3["value"] > c3["sequential_one_by_one"]["accepted_tokens_per_s"]
    assert c3["batched_beams"]["accepted_tokens_per_s"] > 0.8 * c3["value"]
    assert side["c4_shard"]["dtype"] == "f32" and side["c5_shard"]["dtype"].startswith("bf16")
    # the opt-in gemm_x3 form of the fp32 products: its own dtype and workload tag, never the headline's
    for name in ("c2_gemm_x3", "c4_shard_gemm_x3"):
        assert "gemm_x3" in side[name]["dtype"] and "gemm_x3" in side[name]["workload"]
    assert out["dtype"] == "f32" and "gemm_x3" not in out["config"]["workload"]
    assert side["c4_shard"]["roofline"]["bound"] == side["c5_shard"]["roofline"]["bound"] == "mfma"


def test_graft_entry_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.smoke()
