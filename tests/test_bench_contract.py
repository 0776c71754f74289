"""bench.py prints ONE JSON line with the contract's fields (run small on a GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--texts", "8192", "--steps", "3", "--warmup", "1",
                        "--cpu-sample", "512"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "GB/s" and d["dtype"] == "u8" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["untimed_steps"] == 1 + d["config"]["settle_steps"]   # every untimed step is counted
    # `value` is the strictly serial figure; the overlapped one is an extra object
    assert d["config"]["streams"] == 1 and d["two_streams_overlapped"]["ms_per_step"] > 0
    assert abs(d["value"] - 8192 * 1024 / (d["ms_per_step"] * 1e-3) / 1e9) < 0.02 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["parity_on_sample"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_bench_help_needs_no_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout
