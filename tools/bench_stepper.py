#!/usr/bin/env python3
"""Table plans of the windowed stepper (patterns that fail the streaming proof) on 2^20 x 1 KiB batches:
printable noise and config 2's texts.  MRX_NO_UNION_PASS=1 gives the walk without the union first pass."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd.workloads import make_c2_batch  # noqa: E402


def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    lib = M.load_library()
    n, L = 1 << 20, 1024
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    batches = {"noise": (torch.randint(32, 127, (n, L), generator=g, device="cuda")).to(torch.uint8),
               "c2": make_c2_batch(n, L)}
    for pat in (b"\\d+(\\.\\d+)?", b"(foo|foobar)x", b"\\w+\\d{2}", b"[a-z]+@[a-z]+\\.(com|org)"):
        rx = M.compile_regex(pat)
        for bname, d in batches.items():
            batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
            nb = float(n) * L
            row = {"pattern": pat.decode(), "batch": bname, "union_pass": not os.environ.get("MRX_NO_UNION_PASS")}
            for op, fn in (("search", lambda: rx.match_next(batch)), ("count", lambda: rx.count(batch))):
                dt = timed(fn)
                row[op + "_kernel"] = lib.mrx_last_kernel_name().decode()
                row[op + "_ms"] = round(dt * 1e3, 3)
                row[op + "_GBps"] = round(nb / dt / 1e9, 1)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
