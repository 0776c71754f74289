#!/usr/bin/env python3
"""Stepper plans with a multi-walk table: k_mwalk (several walks side by side, one pass) against the windowed
stepper's restart-per-position loop (mrx_debug_multiwalk(2)) -- count, findall and search on 2^20 x 1 KiB texts:
printable noise, config 2's mix (letter runs: worst case for restarts) and config 4's phone texts."""
import json
import os, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mojo_regex_amd as M
from mojo_regex_amd.workloads import make_c2_batch, make_phone_batch
n, L = 1 << 20, 1024
g = torch.Generator(device="cuda"); g.manual_seed(7)
batches = {"noise": (torch.randint(0, 95, (n, L), generator=g, device="cuda") + 32).to(torch.uint8),
           "config2_mix": make_c2_batch(n, L, device="cuda"), "phone": make_phone_batch(n, L, device="cuda")}
lib = M.load_library()
pats = [b"\\w+\\d{2}", b"\\d+(\\.\\d+)?", b"(foo|foobar)x", b"[a-z]{2}-9*", b"(?:xy){4}@{2}", b"[a-z]+@[a-z]+\\.com", b"\\d{3}-\\d{4}",
        b"[0-9]+\\.[0-9]+", b"\\d{3}-\\d{3}-\\d{4}", b"[A-Z]{10,20}[0-9]{15,25}", b"foo|[a-z]{3}\\d|[ab]", b"xy[a-z]+\\d{2,}|[0-9a-f]|hello"]
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r
for name, d in batches.items():
    batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
    for pat in pats:
        rx = M.compile_regex(pat)
        if os.environ.get("MRX_PROBE_ONLY") == "backset" and "backset=yes" not in rx.describe():
            continue
        if os.environ.get("MRX_PROBE_ONLY") == "backset" and "multiwalk=yes" in rx.describe():
            continue
        if "multiwalk=yes" not in rx.describe() and "multiwalk_req=yes" not in rx.describe() and "backset=yes" not in rx.describe():
            print(json.dumps({"batch": name, "pattern": pat.decode(), "multiwalk": False})); continue
        row = {"batch": name, "pattern": pat.decode()}
        for mode, tag in ((0, "mwalk"), (2, "stepper")):
            lib.mrx_debug_multiwalk(mode)
            dt, c = timed(lambda: rx.count(batch))
            row[tag + "_count_GBps"] = round(n * L / dt / 1e9, 1); row[tag + "_count_kernel"] = lib.mrx_last_kernel_name().decode()
            dt, r = timed(lambda: rx._dev_findall(batch))
            row[tag + "_findall_GBps"] = round(n * L / dt / 1e9, 1); row[tag + "_matches"] = int(r[2])
            dt, _ = timed(lambda: rx.match_next(batch))
            row[tag + "_search_GBps"] = round(n * L / dt / 1e9, 1)
        lib.mrx_debug_multiwalk(0)
        row["same_counts"] = row["mwalk_matches"] == row["stepper_matches"]
        print(json.dumps(row), flush=True)
