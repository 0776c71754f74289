#!/usr/bin/env python3
"""RCCL calls bench.py / dist.py make, on ONE GPU (world_size 1): process-group init, barrier with
device_ids, float64 MAX / SUM all-reduce, the uneven all_gather behind dist.all_gather_v.  The real
N > 1 runs are the driver's; this only proves the call signatures and dtypes against RCCL."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29561")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
from mojo_regex_amd import dist as D
torch.cuda.set_device(0)
D.init("nccl")
dist.barrier(device_ids=[0])
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
u = torch.tensor([2.0, 3.0], dtype=torch.float64, device="cuda:0")
dist.all_reduce(u, op=dist.ReduceOp.SUM)
x = torch.arange(10, dtype=torch.int32, device="cuda:0").reshape(5, 2)
g = D.all_gather_v(x, [5])
sizes = D._all_gather_sizes(1, [5, 7], "cuda:0")
pre = torch.tensor([0, 2, 5], dtype=torch.int64, device="cuda:0")
gp, gs = D.gather_spans(1, pre, x, 5)
torch.cuda.synchronize()
print("nccl selftest ok", t.item(), u.tolist(), g.shape, sizes, gp.tolist(), tuple(gs.shape))
dist.destroy_process_group()
