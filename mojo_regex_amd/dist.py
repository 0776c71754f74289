"""Multi-GPU plumbing for the batch matcher: one process per GPU, texts sharded
by contiguous index ranges, compiled tables replicated, NO data-path collective
(every text is matched independently -- SURVEY.md 8(e)).  torch.distributed is
only used for the rendezvous barrier and to combine per-rank timings/totals:
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of texts owned by `rank`; global order = rank order."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init(backend: str):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend=backend)
    return dist


def barrier(world: int, device=None):
    if world > 1:
        import torch.distributed as dist
        if device is not None and str(device).startswith("cuda"):
            dist.barrier(device_ids=[int(str(device).split(":")[1])] if ":" in str(device) else None)
        else:
            dist.barrier()


def combine(world: int, elapsed_s: float, units: Dict[str, float], device="cpu") -> Dict[str, float]:
    """Whole-job figures: elapsed = MAX over ranks, every unit count = SUM over ranks."""
    out = dict(units)
    out["elapsed_s"] = elapsed_s
    if world <= 1:
        return out
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    keys = sorted(units)
    u = torch.tensor([float(units[k]) for k in keys], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    out = {k: float(v) for k, v in zip(keys, u.tolist())}
    out["elapsed_s"] = float(t.item())
    return out
