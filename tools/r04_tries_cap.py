#!/usr/bin/env python3
"""Where should the pending-tries table stop?  Generated plans whose table lies between 16 KB and 44 KB (run with
MRX_TRIES_CAP_ENTRIES=1400): count / findall on k_mwalk against marks + stepper (mrx_debug... MRX_NO_TRIES is read once,
so the second route comes from a second process: MRX_NO_TRIES=1).  2^19 x 1 KiB texts of bench.py's mix.
usage: MRX_TRIES_CAP_ENTRIES=1400 [MRX_NO_TRIES=1] python tools/r04_tries_cap.py"""
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd import workloads as W  # noqa: E402
from pattern_gen import patterns  # noqa: E402


def timeit(fn, reps=4):
    for _ in range(6):   # (the tries-vs-marks tuner has settled by then)
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


lib = M.load_library()
d = W.make_c2_batch(1 << 19, 1024)
n, L = d.shape
batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
seen = set()
done = 0
for seed in (20260503, 20260504, 20260505, 20260506):
    for ps in patterns(seed, 300):
        if ps in seen or done >= 24:
            continue
        seen.add(ps)
        try:
            rx = M.compile_regex(ps.encode())
        except Exception:
            continue
        dsc = rx.describe()
        m = re.search(r"tries_walk=yes configs=(\d+)", dsc)
        if not m or "required-byte route" in dsc:
            continue
        ncls = int(re.search(r"device\.kind=\d+ nstates=\d+ ncls=(\d+)", dsc).group(1))
        done += 1
        row = {"pattern": ps, "configs": int(m.group(1)), "ncls": ncls}
        try:
            row["count_GBps"] = round(n * L / timeit(lambda: rx.count(batch)) / 1e9, 1)
            row["count_kernel"] = lib.mrx_last_kernel_name().decode()
        except M.UnsupportedPattern as e:
            row["refused"] = str(e)[:60]
        print(json.dumps(row), flush=True)
