#!/usr/bin/env python3
"""Round 4: the chain sub on arbitrary bytes (NUL, high bytes) against the lane-per-text interpreter, text by text."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mojo_regex_amd as M  # noqa: E402

lib = M.load_library()
rng = np.random.default_rng(7)
texts = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 300))).astype(np.uint8)) for _ in range(400)]
texts += [bytes(rng.choice(list(b"ab \x00\xff\x801_"), size=int(rng.integers(0, 900))).astype(np.uint8)) for _ in range(200)]
bad = 0
for pat, repl in ((rb"([^ ]+) (\w+)", rb"\2 \1"), (rb"(\w+) (\w+)", rb"<\2|\1>"), (rb"([^ ]{2,})(\d+)", rb"\2\1"), (rb"(\w+)@([a-z]+)", rb"\2")):
    rx = M.compile_regex(pat)
    form = "chain_groups=yes" in rx.describe()
    got = rx.sub(repl, texts, 0)
    k = lib.mrx_last_kernel_name()
    lib.mrx_debug_force_generic(1)
    want = rx.sub(repl, texts, 0)
    lib.mrx_debug_force_generic(0)
    d = sum(a != b for a, b in zip(got, want))
    bad += d
    print(pat, "form" if form else "interpreter", k.decode(), "texts", len(texts), "different", d, flush=True)
print("bad", bad)
