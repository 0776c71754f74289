"""'$' programs on the LazyDFA search: count rate per text kind of the config-2 mix (which texts are slow?)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mojo_regex_amd as M
from mojo_regex_amd.workloads import make_c2_batch
lib = M.load_library()
n, L = 1 << 19, 1024
d = make_c2_batch(n, L)
lower = (d >= 97) & (d <= 122)
digit = (d >= 48) & (d <= 57)
sp = d == 32
adv = (d[:, -1] == 33) & lower[:, :-1].all(dim=1)
full = (lower | digit).all(dim=1)
tok = (lower | digit | sp).all(dim=1) & sp.any(dim=1)
kinds = {"adversarial": adv, "full": full, "tokens": tok, "noise": ~(adv | full | tok), "mix": torch.ones(n, dtype=torch.bool, device="cuda")}
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / reps
for pat in [p.encode() for p in (sys.argv[1:] or ["[a-z]+[0-9]+$", "(foo|[0-9]+)$"])]:
    rx = M.compile_regex(pat)
    row = {"pattern": pat.decode()}
    for name, sel in kinds.items():
        rows = d[sel][: 1 << 16].contiguous()
        m = rows.shape[0]
        if m == 0: continue
        b = M.DeviceBatch.strided(rows.reshape(-1), L, length=L)
        row[name + "_GBps"] = round(m * L / t(lambda: rx.count(b)) / 1e9, 1)
    row["kernel"] = lib.mrx_last_kernel_name().decode()
    print(json.dumps(row), flush=True)
