#!/usr/bin/env python3
"""Ragged CSR batch (config 2's texts cut to U[64, 1024] bytes, packed back to back): findall / count /
search of `[a-z]+\\d+`, whole calls.  One JSON line; MRX_PIECE_C etc. are read by the library."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd import workloads as W  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    lib = M.load_library()
    if os.environ.get("MRX_DYN_MODE"):
        lib.mrx_debug_dynamic_texts(int(os.environ["MRX_DYN_MODE"]))   # 1 always k_stream_dyn, 2 never
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    d = W.make_c2_batch(1 << 20, 1024)
    data, offsets = W.to_ragged(d, lo)
    del d
    batch = M.DeviceBatch(data, offsets)
    n, nbytes = batch.n, int(data.numel())
    pat = (sys.argv[2] if len(sys.argv) > 2 else "[a-z]+\\d+").encode()
    rx = M.compile_regex(pat)
    prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    spans = torch.empty((n * 32, 2), dtype=torch.int32, device="cuda")
    row = {"batch": "c2 texts cut to U[%d, 1024], CSR" % lo, "pattern": pat.decode(), "texts": n, "bytes": nbytes,
           "env": {k: v for k, v in os.environ.items() if k.startswith("MRX_")}}
    t = timeit(lambda: rx._dev_findall(batch, out=(prefix, spans)))
    row.update({"findall_kernel": lib.mrx_last_kernel_name().decode(), "findall_ms": round(t * 1e3, 3),
                "findall_GBps": round(nbytes / t / 1e9, 1), "matches": int(prefix[n].item())})
    t = timeit(lambda: rx.count(batch))
    row.update({"count_kernel": lib.mrx_last_kernel_name().decode(), "count_ms": round(t * 1e3, 3),
                "count_GBps": round(nbytes / t / 1e9, 1)})
    t = timeit(lambda: rx.match_next(batch))
    row.update({"search_kernel": lib.mrx_last_kernel_name().decode(), "search_ms": round(t * 1e3, 3),
                "search_GBps": round(nbytes / t / 1e9, 1)})
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
