"""Per-task timeline of k_stream_bits on the headline batch (mrx_debug_stream_bits_trace)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mojo_regex_amd as M
from mojo_regex_amd.workloads import make_c2_batch
lib = M.load_library()
lib.mrx_debug_stream_bits(1)
n, L = 1 << 20, 1024
batch_t = make_c2_batch(n, L, seed=20260102, device="cuda")
batch = M.DeviceBatch.strided(batch_t.reshape(-1), L, length=L)
rx = M.compile_regex(b"[a-z]+\\d+")
out = (torch.empty(n + 1, dtype=torch.int64, device="cuda"), torch.empty((n * 32, 2), dtype=torch.int32, device="cuda"))
nw = n // 64
for env in ({}, {"MRX_SB_DEBUG": "2"}, {"MRX_SB_PHASE": "0"}):
    for k in ("MRX_SB_DEBUG", "MRX_SB_PHASE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(20):
        rx.findall_async(batch, out)
    torch.cuda.synchronize()
    tr = torch.zeros(4 * nw, dtype=torch.int64, device="cuda")
    lib.mrx_debug_stream_bits_trace(tr.data_ptr())
    rx.findall_async(batch, out)
    torch.cuda.synchronize()
    lib.mrx_debug_stream_bits_trace(None)
    t = tr.cpu().numpy().reshape(nw, 4).astype(np.float64)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01   # us
    scan, wait, fin = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    def pct(x): return [round(float(np.percentile(x, q)), 1) for q in (1, 25, 50, 75, 99, 100)]
    print(json.dumps({"env": env, "kernel_us": round(float(t[:, 3].max()), 1), "scan_us_pct": pct(scan), "pub_to_base_us_pct": pct(wait),
                      "base_to_done_us_pct": pct(fin)}))
    # per round (tasks in ticket order: 3072 at a time): when did its scans end / its tasks end
    for r in range(0, nw, 3072):
        sl = slice(r, min(nw, r + 3072))
        print("  tasks %5d..: start %s scan_end %s done %s" % (r, pct(t[sl, 0]), pct(t[sl, 1]), pct(t[sl, 3])))
