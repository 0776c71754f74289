#!/usr/bin/env python3
"""The reference's benchmark list (tests/bench_engine_cases.py) on one GPU.

Each case's text becomes a batch: n rotations of it (row i = the text rotated by 37 i bytes), n chosen
so that the batch is about 256 MiB (at most 2^20 texts), fixed pitch, device resident.  The case's
operation is enqueued REPS times and timed as a whole.  Printed per case: which kernel ran, GB/s of
input, ns per text.  (Parity of every case is tests/test_gpu_bench_suite.py's business; nothing here
touches the oracle.)
usage: python tools/bench_suite.py [name-prefix]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # bench_engine_cases: the reference's benchmark list as data
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
import bench_engine_cases as B  # noqa: E402
from mojo_regex_amd.api import UnsupportedPattern  # noqa: E402

TARGET_BYTES = 256 << 20
REPS = 5
WARM = 6   # untimed calls (round 4: the required-byte route tuner has settled by then: two routes untimed, two timed, one to read the events)


def make_batch(text: bytes, csr: bool):
    L = len(text)
    pitch = L if csr else (L + 15) // 16 * 16   # sub / is_match take CSR batches: rows back to back
    n = max(64, min(1 << 20, TARGET_BYTES // pitch))
    t2 = torch.frombuffer(bytearray(text + text), dtype=torch.uint8).cuda()
    data = torch.zeros((n, pitch), dtype=torch.uint8, device="cuda")
    col = torch.arange(L, device="cuda")
    step = max(1, (64 << 20) // max(L, 1))
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        rot = (torch.arange(lo, hi, device="cuda") * 37) % L
        data[lo:hi, :L] = t2[rot[:, None] + col[None, :]]
    if csr:
        # (the builder of these offsets knows offsets[n] and the longest text: the known-totals entry points)
        return M.DeviceBatch.csr_known(data.reshape(-1), torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device="cuda"),
                                       n * L, L), n, L
    return M.DeviceBatch.strided(data.reshape(-1), pitch, length=L), n, L


def main():
    lib = M.load_library()
    lib.mrx_debug_long_text_kernels(int(os.environ.get("MRX_LONG_TEXT_MODE", "0")))   # 1 always, 2 never (A/B runs)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    ops = os.environ.get("MRX_SUITE_OPS", "")   # e.g. "findall": only the cases of these operations
    for case in B.CASES:
        if not case.name.startswith(only) or (ops and case.op not in ops.split(",")):
            continue
        row = {"case": case.name, "op": case.op, "pattern": case.pattern.decode()[:60], "text_bytes": len(case.text)}
        rx = M.compile_regex(case.pattern)
        row["engine"] = rx.get_engine_type()
        batch, n, L = make_batch(case.text, case.op in ("sub", "is_match"))
        row["texts"] = n
        try:
            if case.op == "findall":
                prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
                _, _, total = rx._dev_findall(batch)
                spans = torch.empty((max(total, 64), 2), dtype=torch.int32, device="cuda")
                fn = lambda: rx.findall_async(batch, (prefix, spans))  # noqa: E731
                row["matches"] = total
            elif case.op == "search":
                fn = lambda: rx.match_next(batch)  # noqa: E731
            elif case.op == "match_first":
                fn = lambda: rx.match_first(batch)  # noqa: E731
            elif case.op == "is_match":
                fn = lambda: rx.is_match(batch)  # noqa: E731
            else:
                cap = n * (2 * L + 64)
                fn = lambda: rx.sub_dev(case.repl, batch, case.count, out_cap=cap)  # noqa: E731
            for _ in range(WARM):   # the per-stream scratch arena settles within two calls of a new shape, the route tuner within five
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(REPS):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / REPS
            row.update({"kernel": lib.mrx_last_kernel_name().decode(), "ms": round(dt * 1e3, 3),
                        "GBps": round(n * L / dt / 1e9, 1), "ns_per_text": round(dt / n * 1e9, 2)})
        except UnsupportedPattern as e:
            row["refused"] = str(e)[:100]
        print(json.dumps(row), flush=True)
        del batch
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
