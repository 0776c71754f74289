"""Config 5 ((x|y|foo|bar)+, 4 KiB texts, ~700 matches per text) findall on a batch of `n` texts: whole-call time; run
under rocprofv3 (--kernel-trace --stats / --pmc FETCH_SIZE / WRITE_SIZE) by tools/r04_c5_pmc.sh."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojo_regex_amd as M
from mojo_regex_amd.workloads import make_alt_batch
lib = M.load_library()
n, L = int(os.environ.get("C5_TEXTS", 1 << 20)), 4096
d = make_alt_batch(n, L, device="cuda")
rx = M.compile_regex(b"(x|y|foo|bar)+")
batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
spans = torch.empty((n * 720, 2), dtype=torch.int32, device="cuda")
for _ in range(2):
    rx.findall_async(batch, (prefix, spans))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    rx.findall_async(batch, (prefix, spans))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(json.dumps({"texts": n, "spans": int(prefix[-1].item()), "findall_ms": round(dt * 1e3, 3), "GBps": round(n * L / dt / 1e9, 1),
                  "ms_per_16GiB": round(dt * 1e3 * (1 << 22) / n, 2), "kernel": lib.mrx_last_kernel_name().decode()}), flush=True)
