#!/usr/bin/env python3
"""Non-streamable but steppable pattern on printable noise: windowed stepper (level 1) vs the
literal restatement (level 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mojo_regex_amd as M
n, L = 1 << 20, 1024
g = torch.Generator(device="cuda"); g.manual_seed(7)
d = (torch.randint(0, 95, (n, L), generator=g, device="cuda") + 32).to(torch.uint8)
batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
lib = M.load_library()
for pat in (b"\\w+\\d{2}", b"\\d+(\\.\\d+)?", b"(foo|foobar)x", b"(\\d{3})(\\d{3})(\\d{4})"):
    rx = M.compile_regex(pat)
    row = [pat.decode()]
    for level in (1, 2):
        lib.mrx_debug_force_generic(level)
        rx.count(batch); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): c = rx.count(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        row.append("%s %.2f ms %.0f GB/s total=%d" % (lib.mrx_last_kernel_name().decode(), dt * 1e3, n * L / dt / 1e9, int(c.sum().item())))
    lib.mrx_debug_force_generic(0)
    print(" | ".join(row), flush=True)
