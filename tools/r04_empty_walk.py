#!/usr/bin/env python3
"""Empty-match plans whose walks never overshoot (PF_MW_EMPTY): k_mwalk's one pass against the stepper's EMPTY form.
2^18 x 1 KiB texts of bench.py's mix; GB/s of input."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd import workloads as W  # noqa: E402


def timeit(fn, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


lib = M.load_library()
d = W.make_c2_batch(1 << 18, 1024)
n, L = d.shape
batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
for pat in (b"a{0,2}", b"\\s?", b"z*", b"(abc)*", b"http?", b"x?y?", b"(foo)?(bar)?"):
    rx = M.compile_regex(pat)
    if "empty_walk=1" not in rx.describe():
        print(json.dumps({"pattern": pat.decode(), "skipped": "no empty_walk form"}))
        continue
    _, _, total = rx._dev_findall(batch, span_cap=n * (L + 1))
    prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    spans = torch.empty((total, 2), dtype=torch.int32, device="cuda")
    row = {"pattern": pat.decode(), "matches": total}
    for name, mode in (("mwalk", 0), ("stepper", 2)):
        lib.mrx_debug_multiwalk(mode)
        row["findall_GBps_" + name] = round(n * L / timeit(lambda: rx.findall_async(batch, (prefix, spans))) / 1e9, 1)
        row["findall_kernel_" + name] = lib.mrx_last_kernel_name().decode()
        row["count_GBps_" + name] = round(n * L / timeit(lambda: rx.count(batch)) / 1e9, 1)
    lib.mrx_debug_multiwalk(0)
    print(json.dumps(row), flush=True)
    del prefix, spans
    torch.cuda.empty_cache()
