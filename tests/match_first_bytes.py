#!/usr/bin/env python3
"""match_first's algorithmic bytes are data dependent -- sum over the texts of
min(len, bytes consumed before the dead transition + 1) -- so they come from the oracle's C port.
Prints that fraction for config 2's batch (first 65536 texts) and the match_first rate against it.
Lives under tests/ because it uses the oracle.   usage (GPU box): python tests/match_first_bytes.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd import workloads as W  # noqa: E402
from mrx_ref.cfast import CDfa  # noqa: E402

pat = b"[a-z]+\\d+"
n, L, m = 1 << 20, 1024, 1 << 16
d = W.make_c2_batch(n, L)
host = d[:m].cpu().numpy()
offs = np.arange(0, (m + 1) * L, L, dtype=np.int64)
frac = CDfa(pat).match_first_bytes(host.reshape(-1), offs) / float(m * L)
rx = M.compile_regex(pat)
batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
rx.match_first(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    rx.match_first(batch)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 5
print(json.dumps({"match_first_algorithmic_fraction": round(frac, 4), "match_first_ms": round(t * 1e3, 3),
                  "match_first_algorithmic_GBps": round(frac * n * L / t / 1e9, 1)}))
