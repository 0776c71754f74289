"""Ragged CSR batch (config 2's texts cut to U[64, 1024]): count / findall on k_stream_dyn, a few calls each -- run under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE to see what the kernel really moves (tools/r04_ragged_pmc.sh)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojo_regex_amd as M
from mojo_regex_amd import workloads as W
lib = M.load_library()
d = W.make_c2_batch(1 << 20, 1024)
data, offsets = W.to_ragged(d, 64)
del d
batch = M.DeviceBatch(data, offsets)
n, nbytes = batch.n, int(data.numel())
rx = M.compile_regex(b"[a-z]+\\d+")
prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
spans = torch.empty((n * 32, 2), dtype=torch.int32, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / reps
row = {"bytes": nbytes, "texts": n}
row["count_ms"] = round(t(lambda: rx.count(batch)) * 1e3, 4); row["count_kernel"] = lib.mrx_last_kernel_name().decode()
row["findall_ms"] = round(t(lambda: rx.findall_async(batch, (prefix, spans))) * 1e3, 4); row["findall_kernel"] = lib.mrx_last_kernel_name().decode()
row["search_ms"] = round(t(lambda: rx.match_next(batch)) * 1e3, 4)
print(json.dumps(row), flush=True)
