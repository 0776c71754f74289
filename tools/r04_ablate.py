"""k_stream_bits on the headline batch with pieces switched off (MRX_SB_DEBUG) / other grid sizes (MRX_SB_BPC):
kernel time from the library's own HIP events.  Results are wrong for any debug value but 0."""
import sys, os, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojo_regex_amd as M
from mojo_regex_amd.workloads import make_c2_batch
lib = M.load_library()
lib.mrx_debug_stream_bits(1)
n, L = 1 << 20, 1024
batch_t = make_c2_batch(n, L, seed=20260102, device="cuda")
batch = M.DeviceBatch.strided(batch_t.reshape(-1), L, length=L)
rx = M.compile_regex(b"[a-z]+\\d+")
out = (torch.empty(n + 1, dtype=torch.int64, device="cuda"), torch.empty((n * 32, 2), dtype=torch.int32, device="cuda"))
def run(label, env):
    for k in ("MRX_SB_DEBUG", "MRX_SB_BPC"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(30):
        rx.findall_async(batch, out)
    torch.cuda.synchronize()
    lib.mrx_timing_enable(1); lib.mrx_timing_reset()
    for _ in range(20):
        rx.findall_async(batch, out); torch.cuda.synchronize()
    c = ctypes.c_int64(0)
    ms = lib.mrx_timing_scan_ms(ctypes.byref(c))
    lib.mrx_timing_enable(0)
    print(json.dumps({"label": label, "kernel": lib.mrx_last_kernel_name().decode(), "ms": round(ms, 4)}), flush=True)
import sys as _s
runs = [("full (phase by slot)", {}),
        ("no phases", {"MRX_SB_PHASE": "0"}),
        ("phase by block % 3", {"MRX_SB_PHASE": "2"}),
        ("phase by block / (grid/3)", {"MRX_SB_PHASE": "3"}),
        ("phase by slot, 2 periods", {"MRX_SB_PHASE_SLEEPS": "2"}),
        ("phase by slot, 4 periods", {"MRX_SB_PHASE_SLEEPS": "4"}),
        ("phase by slot, 6 periods", {"MRX_SB_PHASE_SLEEPS": "6"}),
        ("no lookback", {"MRX_SB_DEBUG": "2"}),
        ("no lookback, no phases", {"MRX_SB_DEBUG": "2", "MRX_SB_PHASE": "0"}),
        ("no expansion", {"MRX_SB_DEBUG": "1"}),
        ("no span stores", {"MRX_SB_DEBUG": "4"}),
        ("no expansion, no lookback", {"MRX_SB_DEBUG": "3"}),
        ("scan + publish only", {"MRX_SB_DEBUG": "15"}),
        ("scan + publish only, no phases", {"MRX_SB_DEBUG": "15", "MRX_SB_PHASE": "0"}),
        ("full, 2 blocks/CU", {"MRX_SB_BPC": "2"}),
        ("full, 6 blocks/CU", {"MRX_SB_BPC": "6"})]
for label, env in runs:
    for k in ("MRX_SB_PHASE", "MRX_SB_PHASE_SLEEPS"):
        os.environ.pop(k, None)
    run(label, env)
lib.mrx_debug_stream_bits(0)
run("three launches (scan kernel only)", {})
