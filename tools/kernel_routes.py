#!/usr/bin/env python3
"""Which kernel serves each operation of each pattern of tests/test_gpu_parity.py's PATTERNS list
(small CSR batch).  usage: python tools/kernel_routes.py  -> one line per pattern"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402


def main():
    import re
    src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
    body = src[src.index("PATTERNS = ["):]
    body = body[:body.index("]\n") + 1]
    pats = eval(body.split("=", 1)[1])
    lib = M.load_library()
    rng = np.random.default_rng(1)
    al = np.frombuffer(b"abcxyz0189 -.@helowrd", dtype=np.uint8)
    texts = [bytes(rng.choice(al, size=int(rng.integers(0, 200))).tolist()) for _ in range(512)]
    batch = M.DeviceBatch.from_texts(texts)
    generic = 0
    for p in pats:
        rx = M.compile_regex(p)
        row = []
        for op, fn in (("match_first", lambda: rx.match_first(batch)), ("search", lambda: rx.match_next(batch)),
                       ("findall", lambda: rx._dev_findall(batch)), ("count", lambda: rx.count(batch)),
                       ("is_match", lambda: rx.is_match(batch)), ("sub", lambda: rx.sub_dev(b"#", batch))):
            try:
                fn()
                k = lib.mrx_last_kernel_name().decode()
            except M.UnsupportedPattern:
                k = "refused"
            if k in ("k_match", "k_findall_count", "k_sub_size"):
                generic += 1
                k = k.upper()
            row.append("%s=%s" % (op, k))
        print("%-34s %s" % (p.decode(), "  ".join(row)), flush=True)
    print("generic routes:", generic)


if __name__ == "__main__":
    main()
