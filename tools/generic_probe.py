#!/usr/bin/env python3
"""Generic lane-per-text kernels on config 4 (forced): timing for a PMC / trace run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mojo_regex_amd as M
from mojo_regex_amd import workloads as W
n, L = 1 << 20, 1024
d = W.make_phone_batch(n, L)
batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
rx = M.compile_regex(b"(\\d{3})(\\d{3})(\\d{4})")
lib = M.load_library()
lib.mrx_debug_force_generic(1)
for _ in range(2):
    rx.count(batch); rx.match_next(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): rx.count(batch)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(3): rx.match_next(batch)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(lib.mrx_last_kernel_name(), "generic count %.2f ms (%.0f GB/s)  search %.2f ms (%.0f GB/s)" % ((t1 - t0) / 3 * 1e3, n * L / ((t1 - t0) / 3) / 1e9, (t2 - t1) / 3 * 1e3, n * L / ((t2 - t1) / 3) / 1e9))
