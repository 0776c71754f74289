import sys, time
sys.path.insert(0, '/root/repo')
import torch
import mojo_regex_amd as M
from mojo_regex_amd.workloads import make_c2_batch
n, L = 1 << 20, 1024
d = make_c2_batch(n, L)
batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
rx = M.compile_regex(b"[a-z]+\\d+")
out = (torch.empty(n + 1, dtype=torch.int64, device="cuda"), torch.empty((n * 32, 2), dtype=torch.int32, device="cuda"))
for _ in range(5): rx.findall_async(batch, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): rx.findall_async(batch, out)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue per step %.1f us, total per step %.1f us" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
counts = torch.empty(n, dtype=torch.int32, device="cuda")
import ctypes as C
lib = M.load_library()
def cnt():
    lib.mrx_count_strided_dev(rx._h, C.c_void_p(d.data_ptr()), L, None, L, n, C.c_void_p(counts.data_ptr()), rx._stream_ptr())
for _ in range(5): cnt()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): cnt()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("count: enqueue per call %.1f us, total per call %.1f us" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
x = torch.empty(1 << 20, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): x.add_(1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("torch add_: enqueue per call %.1f us" % ((t1 - t0) / 200 * 1e6))
