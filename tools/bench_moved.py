#!/usr/bin/env python3
"""Round 2: operations that left the generic lane-per-text kernels, timed on their new kernel and on the
literal restatement (mrx_debug_force_generic(2)) over the same device-resident batch.

Batch: the first 2^18 texts of bench.py's workload without its adversarial tenth (the restatement is
quadratic there and would take seconds per call), 1 KiB each, fixed pitch.
usage: python tools/bench_moved.py   -> one JSON line per (pattern, op)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd import workloads  # noqa: E402

N, L, REPS = 1 << 18, 1024, 5

CASES = [
    (b"[a-z]+\\d+", "is_match"), (b"\\d+", "is_match"), (b"hello", "is_match"),
    (b"(\\d{3})(\\d{3})(\\d{4})", "is_match"),
    (b"^[a-z]+", "findall"), (b"^[a-z]+", "count"), (b"^hello", "findall"), (b"^\\d+x", "count"),
    (b"hello.*world", "search"), (b"[a-z]+@example\\.com", "search"), (b"\\d+@example\\.com", "search"),
    (b"hello.*", "findall"), (b".*@example\\.com", "count"),
    (b"(abc)*", "count"), (b"(abc)*", "findall"), (b"(?:ab|cd)*x?", "count"), (b"^abc$", "search"), (b"^hello$", "findall"),
]


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS


def main():
    lib = M.load_library()
    dev = workloads.make_c2_batch(N + N // 4, L, seed=7)       # uint8 [n, L] on the device
    adversarial = (dev[:, L - 1] == 33) & (dev[:, 0] >= 97)     # L - 1 lowercase bytes + '!'
    dev = dev[~adversarial][:N].contiguous()
    n = dev.shape[0]
    batch = M.DeviceBatch.strided(dev.reshape(-1), L, length=L)
    for pat, op in CASES:
        rx = M.compile_regex(pat)
        row = {"pattern": pat.decode(), "op": op, "texts": n, "engine": rx.get_engine_type()}
        fn = {"is_match": lambda: rx.is_match(batch), "findall": lambda: rx._dev_findall(batch),
              "count": lambda: rx.count(batch), "search": lambda: rx.match_next(batch)}[op]
        try:
            for mode, key in ((0, "new"), (2, "restatement")):
                lib.mrx_debug_force_generic(mode)
                dt = timed(fn)
                row[key] = {"kernel": lib.mrx_last_kernel_name().decode(), "ms": round(dt * 1e3, 3),
                            "GBps": round(n * L / dt / 1e9, 1)}
            row["speedup"] = round(row["restatement"]["ms"] / row["new"]["ms"], 1)
        except M.UnsupportedPattern as e:
            row["refused"] = str(e)[:120]
        finally:
            lib.mrx_debug_force_generic(0)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
