#!/usr/bin/env python3
"""CSR batches of short ragged texts (log-line like): count / findall of [a-z]+\\d+ vs length mix."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mojo_regex_amd as M
from mojo_regex_amd import workloads as W

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

rx = M.compile_regex(b"[a-z]+\\d+")
for lo, hi, n in ((16, 200, 1 << 23), (64, 512, 1 << 22), (200, 1024, 1 << 21)):
    base = W.make_c2_batch(n, hi)
    data, off = W.to_ragged(base, lo, seed=3)
    del base
    batch = M.DeviceBatch(data, off)
    nb = int(data.numel())
    prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    spans = torch.empty((nb // 6 + n, 2), dtype=torch.int32, device="cuda")
    row = {"lens": [lo, hi], "texts": n, "bytes": nb,
           "count_GBps": round(nb / timeit(lambda: rx.count(batch)) / 1e9, 1),
           "search_GBps": round(nb / timeit(lambda: rx.match_next(batch)) / 1e9, 1),
           "findall_GBps": round(nb / timeit(lambda: rx._dev_findall(batch, out=(prefix, spans))) / 1e9, 1)}
    print(json.dumps(row), flush=True)
    del data, off, batch, prefix, spans
    torch.cuda.empty_cache()
