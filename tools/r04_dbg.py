import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mojo_regex_amd as M
lib = M.load_library()
rx = M.compile_regex(b"[a-z]+\\d+")
rng = np.random.default_rng(1)
al = np.frombuffer(b"abcxyz0189 -", dtype=np.uint8)
for n, pitch in ((64 * 16, 1024), (1, 16), (63, 48), (300, 1024)):
    arr = rng.choice(al, size=(n, pitch)).astype(np.uint8)
    d = torch.from_numpy(arr).cuda().reshape(-1)
    b = M.DeviceBatch.strided(d, pitch, length=pitch)
    print("case", n, pitch, flush=True)
    lib.mrx_debug_stream_bits(0)
    p3, s3, t3 = rx._dev_findall(b)
    print(" three", lib.mrx_last_kernel_name(), t3, flush=True)
    lib.mrx_debug_stream_bits(1)
    p1, s1, t1 = rx._dev_findall(b)
    print(" bits", lib.mrx_last_kernel_name(), t1, flush=True)
    print(" equal", t1 == t3, bool(torch.equal(p1, p3)), bool(torch.equal(s1[:t1], s3[:t3])), flush=True)
