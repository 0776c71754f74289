#!/usr/bin/env python3
"""Plain-route plans outside the multi-walk proofs whose walks stay within seven bytes of their match (PF_MW_TRIES):
one pass on k_mwalk against round 3's route (marks + stepper; MRX_NO_TRIES=1 in a second run).  2^20 x 1 KiB texts:
bench.py's mix and config 4's phone texts; GB/s of input.  usage: python tools/r04_tries_walk.py"""
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
    for _ in range(6):   # (the tries-vs-marks tuner has settled by then: four measured calls, one to read the events)
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


lib = M.load_library()
for bname, d in (("config2_mix", W.make_c2_batch(1 << 20, 1024)), ("phone", W.make_phone_batch(1 << 20, 1024))):
    n, L = d.shape
    batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
    for pat in (b"foo|[a-z]{3}\\d|[ab]", b"bar|[+]\\w([.-]){2,}|c", b"xy[a-z]\\d{2}|[0-9a-f]|hello", b"(?:ab|abc)d|[0-9]{2}x"):
        rx = M.compile_regex(pat)
        if "tries_walk=yes" not in rx.describe():
            print(json.dumps({"pattern": pat.decode(), "skipped": "no tries form"}))
            continue
        _, _, total = rx._dev_findall(batch)
        prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        spans = torch.empty((max(total, 64), 2), dtype=torch.int32, device="cuda")
        row = {"batch": bname, "pattern": pat.decode(), "matches": total,
               "findall_GBps": round(n * L / timeit(lambda: rx.findall_async(batch, (prefix, spans))) / 1e9, 1),
               "findall_kernel": None, "count_GBps": None}
        row["findall_kernel"] = lib.mrx_last_kernel_name().decode()
        row["count_GBps"] = round(n * L / timeit(lambda: rx.count(batch)) / 1e9, 1)
        row["count_kernel"] = lib.mrx_last_kernel_name().decode()
        print(json.dumps(row), flush=True)
        del prefix, spans
    del d, batch
    torch.cuda.empty_cache()
