"""bench.py --gpus N: never a silent 1-GPU run.  Without the torchrun environment the script starts
its N ranks itself; WORLD_SIZE != --gpus is an error.  The GPU test rehearses N = 2 on the one-GPU
box (ranks share cuda:0 over gloo; RCCL refuses two ranks on one device) and runs the results
exchange of SURVEY.md 8(e)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(kw)
    return env


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=300,
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0
    assert "WORLD_SIZE (2) != --gpus (4)" in (r.stderr + r.stdout)


def test_more_ranks_than_gpus_is_an_error_not_a_smaller_run():
    import torch
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 1) if have else "2"], capture_output=True, text=True,
                       timeout=300, env=_env())
    assert r.returncode != 0
    assert "GPU(s)" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_two_ranks_self_launched_with_results_exchange():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--texts", "16384", "--steps", "3",
                        "--warmup", "1", "--settle", "2", "--c3-texts", "32768"], capture_output=True, text=True, timeout=900,
                       env=_env(MRX_BENCH_SHARE_GPU="1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["texts_per_gpu"] == 16384
    # the exchange legs run by default for N > 1 (gloo here: two ranks share the GPU, RCCL refuses that)
    g = d["scan_plus_gather"]
    assert "error" not in g, g
    assert g["exchange"]["global_texts"] == 2 * 16384 and g["scan_plus_gather"]["ms_per_step"] > 0
    assert g["scan_only"]["ms_per_step"] > 0 and g["exchange"]["global_spans"] > 0
    c3 = d["config3"]
    assert "error" not in c3, c3
    assert c3["exchange"]["global_texts"] == 2 * 32768 and c3["scan_plus_gather"]["GBps_whole_job"] > 0
    s = d["strong"]
    assert s["total_texts"] == 16384 and s["texts_per_gpu"] == 8192 and s["value"] > 0
    assert "cpu_baseline" not in d   # rank 0 at N = 1 only


@pytest.mark.gpu
def test_a_hanging_exchange_leg_does_not_lose_the_headline():
    """The exchange legs are the only part of bench.py no multi-GPU box has run: past MRX_BENCH_EXTRAS_TIMEOUT the
    watchdog prints the headline line without them -- exactly once, with the leg and the call that was in flight --
    and every rank leaves with a NON-ZERO status: a hang is never reported to the driver as a clean run."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--texts", "16384", "--steps", "3",
                        "--warmup", "1", "--settle", "2", "--c3-texts", "32768"], capture_output=True, text=True, timeout=900,
                       env=_env(MRX_BENCH_SHARE_GPU="1", MRX_BENCH_EXTRAS_TIMEOUT="0.0001"))
    assert r.returncode != 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["frac"] > 0
    err = d["scan_plus_gather"]["error"]
    assert "watchdog" in err and "leg 'scan_plus_gather'" in err and "in flight:" in err
