"""findall step time (whole call, serial, one stream) of k_stream_bits against scan -> sums -> decode for several text
lengths of the config-2 mix, `[a-z]+\\d+` and `\\d+`."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojo_regex_amd as M
from mojo_regex_amd.workloads import make_c2_batch
lib = M.load_library()
for pat in (b"[a-z]+\\d+", b"\\d+"):
    rx = M.compile_regex(pat)
    for L in (1024, 512, 256, 128):
        n = (1 << 30) // L
        batch_t = make_c2_batch(n, L, seed=20260102, device="cuda")
        batch = M.DeviceBatch.strided(batch_t.reshape(-1), L, length=L)
        out = (torch.empty(n + 1, dtype=torch.int64, device="cuda"), torch.empty((n * 32 * L // 1024 + 1024, 2), dtype=torch.int32, device="cuda"))
        res = {}
        for mode in (1, 0):
            lib.mrx_debug_stream_bits(mode)
            for _ in range(30):
                rx.findall_async(batch, out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                rx.findall_async(batch, out)
            torch.cuda.synchronize()
            res[lib.mrx_last_kernel_name().decode()] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        print(json.dumps({"pattern": pat.decode(), "text_bytes": L, "texts": n, "spans": int(out[0][-1].item()), "ms_per_step": res}), flush=True)
        del batch, batch_t, out
        torch.cuda.empty_cache()
