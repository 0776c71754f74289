#!/usr/bin/env python3
"""Throughput of the streaming kernel as a function of text length (fixed pitch = length), 2 GiB of
input each: count, search and findall of [a-z]+\\d+ over the config-2 token mix."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mojo_regex_amd as M
from mojo_regex_amd import workloads as W

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

base = W.make_c2_batch(1 << 21, 1024)          # 2 GiB
rx = M.compile_regex(b"[a-z]+\\d+")
for L in (32, 64, 128, 256, 512, 1024, 4096):
    n = base.numel() // L
    d = base.reshape(-1)
    batch = M.DeviceBatch.strided(d, L, length=L)
    prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    spans = torch.empty((base.numel() // 8, 2), dtype=torch.int32, device="cuda")
    nb = n * L
    row = {"text_bytes": L, "texts": n,
           "count_GBps": round(nb / timeit(lambda: rx.count(batch)) / 1e9, 1),
           "search_GBps": round(nb / timeit(lambda: rx.match_next(batch)) / 1e9, 1),
           "findall_GBps": round(nb / timeit(lambda: rx.findall_async(batch, (prefix, spans))) / 1e9, 1)}
    print(json.dumps(row), flush=True)
