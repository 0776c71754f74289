import sys, time, json
sys.path.insert(0, '/root/repo')
import torch
import mojo_regex_amd as M
from mojo_regex_amd.workloads import make_c2_batch
n, L = 1 << 20, 1024
d = make_c2_batch(n, L)
b = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
lib = M.load_library()
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / reps
for pat in (b"[a-z]+[0-9]+$", b"(foo|[0-9]+)$", b"x*y$|z[0-9]"):
    rx = M.compile_regex(pat)
    if "lazy_end_cache=yes" not in rx.describe():
        print(pat, "not lazy_end"); continue
    row = {"pattern": pat.decode()}
    row["search_GBps"] = round(n * L / t(lambda: rx.match_next(b)) / 1e9, 1); row["search_kernel"] = lib.mrx_last_kernel_name().decode()
    row["count_GBps"] = round(n * L / t(lambda: rx.count(b)) / 1e9, 1); row["count_kernel"] = lib.mrx_last_kernel_name().decode()
    print(json.dumps(row), flush=True)
