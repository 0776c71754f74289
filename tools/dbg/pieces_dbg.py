import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch
import mojo_regex_amd as M
from mrx_ref.cfast import CDfa
pat = b"[a-z]+\\d+"
rx = M.compile_regex(pat)
g = torch.Generator(device="cuda"); g.manual_seed(11)
lens = [5 << 20, (3 << 20) + 7, 70000, 0, 1]
al = torch.tensor(list(b"abcxyz0123456789 -"), dtype=torch.uint8, device="cuda")
data = al[torch.randint(0, al.numel(), (sum(lens),), generator=g, device="cuda")]
data[100:200000] = ord("q"); data[200000:200050] = ord("7")
offsets = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int64, device="cuda")
batch = M.DeviceBatch(data, offsets)
pre, sp, tot = rx._dev_findall(batch)
print(M.load_library().mrx_last_kernel_name())
cd = CDfa(pat)
counts, osp, ototal = cd.findall_batch(data.cpu().numpy(), offsets.cpu().numpy())
got = sp[:tot].cpu().numpy()
bad = np.nonzero((got != osp).any(axis=1))[0]
print("total", tot, ototal, "bad rows", len(bad))
pr = pre.cpu().numpy()
for b in bad[:10]:
    t = int(np.searchsorted(pr, b, side="right") - 1)
    print("row", b, "text", t, "got", got[b], "want", osp[b], "prev", got[b-1], osp[b-1])
