#!/usr/bin/env python3
"""Does a HIP graph of one findall call (scan + prefix sums + decode) shorten the step?  Captures the
asynchronous entry point with torch.cuda.CUDAGraph and replays it; prints ms per step both ways."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd import workloads as W  # noqa: E402

n, L = 1 << 20, 1024
d = W.make_c2_batch(n, L)
rx = M.compile_regex(b"[a-z]+\\d+")
batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
spans = torch.empty((n * 32, 2), dtype=torch.int32, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        rx.findall_async(batch, (prefix, spans))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        rx.findall_async(batch, (prefix, spans))
    torch.cuda.synchronize()
    print("plain  ms/step %.4f" % ((time.perf_counter() - t0) / 50 * 1e3))
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            rx.findall_async(batch, (prefix, spans))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        print("graph  ms/step %.4f  total=%d" % ((time.perf_counter() - t0) / 50 * 1e3, int(prefix[n].item())))
    except Exception as e:
        print("capture failed:", repr(e)[:300])
