#!/usr/bin/env python3
"""Does the scan kernel's time depend on WHERE the batch (or the scratch) sits?  bench.py's step is bimodal between
processes (0.214-0.218 or 0.238-0.240 ms scan).  Here, in ONE process: the same batch contents in several allocations
(earlier ones kept alive, so each sits on other pages), the scan timed on each; then the same for fresh output buffers."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd.workloads import make_c2_batch  # noqa: E402

lib = M.load_library()
n, L = 1 << 20, 1024
rx = M.compile_regex(b"[a-z]+\\d+")
src = make_c2_batch(n, L)
keep = []


def scan_ms(batch, out, reps=20):
    for _ in range(5):
        rx.findall_async(batch, out)
    torch.cuda.synchronize()
    lib.mrx_timing_enable(1)
    lib.mrx_timing_reset()
    for _ in range(reps):
        rx.findall_async(batch, out)
        torch.cuda.synchronize()
    k = ctypes.c_int64(0)
    lib.mrx_timing_scan_ms.restype = ctypes.c_double
    ms = lib.mrx_timing_scan_ms(ctypes.byref(k))
    lib.mrx_timing_enable(0)
    return round(ms, 4)


out = (torch.empty(n + 1, dtype=torch.int64, device="cuda"), torch.empty((n * 32, 2), dtype=torch.int32, device="cuda"))
for trial in range(8):
    pad = torch.empty(((trial * 37 + 11) << 20,), dtype=torch.uint8, device="cuda")   # shift what the allocator hands out next
    d = src.clone()
    keep.append((pad, d))
    b = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
    # the same pages under two other readers: the count kernel (no record stream) and a plain device copy
    import time
    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / reps * 1e3, 4)
    dst = torch.empty_like(src) if trial == 0 else dst
    print(json.dumps({"trial": trial, "batch_ptr": hex(d.data_ptr()), "scan_ms": scan_ms(b, out),
                      "count_ms": timed(lambda: rx.count(b)), "copy_ms": timed(lambda: dst.copy_(d))}), flush=True)

# the same batch (the last one), the record stream skewed inside its scratch allocation
b = M.DeviceBatch.strided(keep[-1][1].reshape(-1), L, length=L)
for skew in (0, 256, 1024, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 8 << 20, (8 << 20) + 4096, 64 << 20, (64 << 20) + 65536):
    lib.mrx_debug_rec_skew(skew)
    print(json.dumps({"rec_skew": skew, "scan_ms": scan_ms(b, out)}), flush=True)
lib.mrx_debug_rec_skew(0)
