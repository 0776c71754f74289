#!/usr/bin/env python3
"""search (match_next) of plans with a pending-tries table against other plan families: whole-batch rate on bench.py's
mix (2^20 x 1 KiB).  usage: python tools/r04_search_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojo_regex_amd as M
from mojo_regex_amd import workloads as W
lib = M.load_library()
d = W.make_c2_batch(1 << 20, 1024)
n, L = d.shape
batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
def timeit(fn, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for pat in (b"foo|[a-z]{3}\\d|[ab]", b"bar|[+]\\w([.-]){2,}|c", b"xy[a-z]\\d{2}|[0-9a-f]|hello", b"(abc)*", b"[A-Z]{3}\\d|zz9"):
    rx = M.compile_regex(pat)
    t = timeit(lambda: rx.match_next(batch))
    print(json.dumps({"pattern": pat.decode(), "search_GBps": round(n * L / t / 1e9, 1), "kernel": lib.mrx_last_kernel_name().decode(),
                      "tries": "tries_walk=yes" in rx.describe()}))
