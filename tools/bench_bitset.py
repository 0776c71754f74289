#!/usr/bin/env python3
"""Bitset-NFA kernel (k_wstep<., 0, 1>) against the literal restatement's walk_bitset and the LazyDFA table
kernels on BASELINE config 4's batch (2^20 x 1 KiB phone texts) and on printable noise.
usage: python tools/bench_bitset.py  -> one JSON line per (pattern, batch, kernel, op)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd.workloads import make_phone_batch, make_alt_batch  # noqa: E402


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    lib = M.load_library()
    n, L = 1 << 20, 1024
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    batches = {"config4": make_phone_batch(n, L),
               "noise": (torch.randint(32, 127, (n, L), generator=g, device="cuda")).to(torch.uint8)}
    pats = [(b"(\\d{3})(\\d{3})(\\d{4})", "config4"), (b"(\\d{3})(\\d{3})(\\d{4})", "noise"),
            (b"(foo|bar)+baz?", "noise"), (b"[a-z]+\\d+", "noise")]
    for pat, bname in pats:
        d = batches[bname]
        batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
        nb = float(n) * L
        for label, kw, force in (("bitset_kernel", dict(lazydfa_semantics=True, bitset_nfa=True), 0),
                                 ("bitset_literal_restatement", dict(lazydfa_semantics=True, bitset_nfa=True), 2),
                                 ("lazydfa_table", dict(lazydfa_semantics=True), 0)):
            rx = M.compile_regex(pat, **kw)
            lib.mrx_debug_force_generic(force)
            try:
                for op, fn in (("search", lambda: rx.match_next(batch)), ("count", lambda: rx.count(batch))):
                    reps = 2 if force else 5
                    dt = timed(fn, reps)
                    print(json.dumps({"pattern": pat.decode(), "batch": bname, "form": label, "op": op,
                                      "kernel": lib.mrx_last_kernel_name().decode(), "ms": round(dt * 1e3, 3),
                                      "GBps": round(nb / dt / 1e9, 1)}), flush=True)
            finally:
                lib.mrx_debug_force_generic(0)


if __name__ == "__main__":
    main()
