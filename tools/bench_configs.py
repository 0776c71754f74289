#!/usr/bin/env python3
"""Throughput of every BASELINE.json config on one GPU (documentation numbers, not the
bench.py contract line).  GB/s = input bytes of the batch / wall time of the whole
call (device-resident inputs, outputs stay on device)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd import workloads as W  # noqa: E402


def timeit(fn, reps=5):
    for _ in range(4):   # the per-stream scratch arena settles within two calls of a new shape; a handle learns from the
        fn()             # call before whether its batches are full of matches (event rows) once that call is through
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    out = []
    lib = M.load_library()
    cases = [
        ("c2 [a-z]+\\d+", b"[a-z]+\\d+", lambda: W.make_c2_batch(1 << 20, 1024), 32),
        ("c3 \\d+", b"\\d+", lambda: W.make_digits_batch(1 << 23, 256), 8),
        ("c4 (\\d{3})(\\d{3})(\\d{4})", b"(\\d{3})(\\d{3})(\\d{4})", lambda: W.make_phone_batch(1 << 20, 1024), 56),
        ("c5 (x|y|foo|bar)+", b"(x|y|foo|bar)+", lambda: W.make_alt_batch(1 << 22, 4096), 720),
        ("c1 hello", b"hello", lambda: W.make_c2_batch(1 << 20, 1024, seed=5), 4),
        # the PikeVM program of configs 4 / 5 (MRX_COMPILE_LAZYDFA_SEMANTICS): determinised
        # table walk vs. bitset-NFA walk, generic lane-per-text kernels
        ("c4 LazyDFA table", b"(\\d{3})(\\d{3})(\\d{4})", lambda: W.make_phone_batch(1 << 20, 1024), 56,
         dict(lazydfa_semantics=True)),
        ("c4 bitset NFA", b"(\\d{3})(\\d{3})(\\d{4})", lambda: W.make_phone_batch(1 << 20, 1024), 56,
         dict(lazydfa_semantics=True, bitset_nfa=True)),
        ("c5 LazyDFA table ('+' honoured)", b"(x|y|foo|bar)+", lambda: W.make_alt_batch(1 << 20, 4096), 720,
         dict(lazydfa_semantics=True)),
        ("c5 bitset NFA ('+' honoured)", b"(x|y|foo|bar)+", lambda: W.make_alt_batch(1 << 20, 4096), 720,
         dict(lazydfa_semantics=True, bitset_nfa=True)),
    ]
    only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "only" else None
    for case in cases:
        name, pat, gen, per_text = case[:4]
        if only and not name.startswith(only):
            continue
        opts = case[4] if len(case) > 4 else {}
        d = gen()
        n, L = d.shape
        rx = M.compile_regex(pat, **opts)
        batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
        prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        spans = torch.empty((n * per_text, 2), dtype=torch.int32, device="cuda")
        nbytes = n * L
        t_find = timeit(lambda: rx.findall_async(batch, (prefix, spans)))
        kernel = lib.mrx_last_kernel_name().decode()
        total = int(prefix[n].item())
        assert total <= spans.shape[0]
        t_search = timeit(lambda: rx.match_next(batch))
        t_first = timeit(lambda: rx.match_first(batch))
        t_count = timeit(lambda: rx.count(batch))
        row = {"config": name, "texts": n, "bytes_per_text": L, "matches": total,
               "findall_kernel": kernel,
               "findall_GBps": round(nbytes / t_find / 1e9, 1), "findall_ms": round(t_find * 1e3, 3),
               "search_GBps": round(nbytes / t_search / 1e9, 1),
               "count_GBps": round(nbytes / t_count / 1e9, 1),
               "match_first_ms": round(t_first * 1e3, 3)}
        out.append(row)
        print(json.dumps(row), flush=True)
        del d, batch, prefix, spans
        torch.cuda.empty_cache()


def sub_bench():
    """regex.sub on config 4 (`\\1-\\2-\\3`) and config 2, device resident."""
    for name, pat, repl, gen in (
            ("c4 sub", b"(\\d{3})(\\d{3})(\\d{4})", b"\\1-\\2-\\3", lambda: W.make_phone_batch(1 << 20, 1024)),
            ("c4 sub literal", b"(\\d{3})(\\d{3})(\\d{4})", b"<phone>", lambda: W.make_phone_batch(1 << 20, 1024)),
            ("c2 sub", b"[a-z]+\\d+", b"#", lambda: W.make_c2_batch(1 << 20, 1024))):
        d = gen()
        n, L = d.shape
        rx = M.compile_regex(pat)
        batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
        cap = n * L * 2
        t = timeit(lambda: rx.sub_dev(repl, batch, 0, out_cap=cap), reps=3)
        off, out = rx.sub_dev(repl, batch, 0, out_cap=cap)
        print(json.dumps({"config": name, "texts": n, "in_bytes": n * L, "out_bytes": int(out.numel()),
                          "sub_ms": round(t * 1e3, 3), "sub_GBps": round(n * L / t / 1e9, 1),
                          "kernel": M.load_library().mrx_last_kernel_name().decode()}), flush=True)
        del d, batch, off, out
        torch.cuda.empty_cache()


def ragged():
    """Config 2 as a ragged CSR batch (the C ABI's primary layout): rows cut to U[64, 1024] bytes and
    packed back to back, so almost every text starts unaligned."""
    lib = M.load_library()
    d = W.make_c2_batch(1 << 20, 1024)
    data, offsets = W.to_ragged(d, 64)
    del d
    batch = M.DeviceBatch(data, offsets)
    n, nbytes = batch.n, int(data.numel())
    for pat in (b"[a-z]+\\d+", b"(\\d{3})(\\d{3})(\\d{4})"):
        rx = M.compile_regex(pat)
        prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        spans = torch.empty((n * 32, 2), dtype=torch.int32, device="cuda")
        row = {"config": "c2 texts, CSR ragged", "pattern": pat.decode(), "texts": n, "bytes": nbytes}
        for name, force in (("stream", 0), ("generic", 1)):
            lib.mrx_debug_force_generic(force)
            t_find = timeit(lambda: rx._dev_findall(batch, out=(prefix, spans)))
            row["findall_kernel_" + name] = lib.mrx_last_kernel_name().decode()
            t_search = timeit(lambda: rx.match_next(batch))
            t_first = timeit(lambda: rx.match_first(batch))
            row.update({"findall_GBps_" + name: round(nbytes / t_find / 1e9, 1),
                        "search_GBps_" + name: round(nbytes / t_search / 1e9, 1),
                        "match_first_ms_" + name: round(t_first * 1e3, 3)})
        lib.mrx_debug_force_generic(0)
        # the same with offsets[n] and the longest text known to the caller (mrx_findall_known_dev: no read-back
        # before the scan)
        batch._end_offset, batch._max_len = nbytes, 1024
        row["findall_GBps_known_totals"] = round(nbytes / timeit(lambda: rx._dev_findall(batch, out=(prefix, spans))) / 1e9, 1)
        row["findall_async_GBps_known_totals"] = round(nbytes / timeit(lambda: rx.findall_async(batch, (prefix, spans))) / 1e9, 1)
        batch._end_offset = batch._max_len = None
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ragged":
        ragged()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sub":
        sub_bench()
        sys.exit(0)
    main()
