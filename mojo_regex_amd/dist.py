"""Multi-GPU plumbing for the batch matcher: one process per GPU, texts sharded
by contiguous index ranges, compiled tables replicated, NO collective inside the
scan (every text is matched independently -- SURVEY.md 8(e)).  Results stay
sharded by default.  The one optional exchange step is "results only":
gather_fixed() (match_first / search / captures: fixed bytes per text, plain
all-gather) and gather_spans() (findall: per-rank totals first, then an
all-gatherv into prefix-sum offsets).  torch.distributed backend "nccl" (= RCCL
over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple


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


# ---- results exchange (optional) -----------------------------------
def _all_gather_sizes(world: int, values: List[int], device) -> List[List[int]]:
    """values of every rank, [world][len(values)] (one small fixed-size all-gather)."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() != "nccl":
        device = "cpu"   # gloo has no all_gather for device tensors
    mine = torch.tensor(values, dtype=torch.int64, device=device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [o.tolist() for o in out]


def all_gather_v(local, rows_per_rank: List[int]):
    """All-gatherv along dim 0: rank r contributes rows_per_rank[r] rows; every rank
    receives the concatenation in rank order.  RCCL has no native allgatherv: with the
    nccl backend torch issues one grouped ncclBroadcast per rank straight into the
    views of the output buffer (direct exchange over the xGMI links); gloo gets the
    same thing as a sequence of broadcasts."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    assert len(rows_per_rank) == world and local.shape[0] == rows_per_rank[rank]
    if dist.get_backend() != "nccl" and local.is_cuda:
        # gloo rehearsal with device-resident results: exchange through host memory
        return all_gather_v(local.cpu(), rows_per_rank).to(local.device)
    out = torch.empty((sum(rows_per_rank),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    views, lo = [], 0
    for r in range(world):
        views.append(out[lo:lo + rows_per_rank[r]])
        lo += rows_per_rank[r]
    if dist.get_backend() == "nccl":
        dist.all_gather(views, local.contiguous())
    else:
        views[rank].copy_(local)
        for r in range(world):
            if rows_per_rank[r]:
                dist.broadcast(views[r], src=r)
    return out


def gather_fixed(world: int, local):
    """match_first / search / captures results ([n_local, ...], fixed size per text) of
    all ranks in global text order."""
    if world <= 1:
        return local
    sizes = _all_gather_sizes(world, [int(local.shape[0])], local.device)
    return all_gather_v(local, [s[0] for s in sizes])


def gather_spans(world: int, prefix, spans, total: int):
    """findall results of all ranks as ONE CSR: (global_prefix[N+1], global_spans[T, 2]).
    prefix: this rank's exclusive prefix [n_local+1]; spans: [>= total, 2]."""
    import torch
    if world <= 1:
        return prefix, spans[:total]
    n_local = int(prefix.shape[0]) - 1
    sizes = _all_gather_sizes(world, [n_local, int(total)], prefix.device)
    ns, totals = [s[0] for s in sizes], [s[1] for s in sizes]
    import torch.distributed as dist
    rank = dist.get_rank()
    base = sum(totals[:rank])
    shifted = prefix[1:] + base          # inclusive ends in the global span numbering
    ends = all_gather_v(shifted, ns)
    g_prefix = torch.cat([torch.zeros(1, dtype=prefix.dtype, device=prefix.device), ends])
    g_spans = all_gather_v(spans[:total], totals)
    return g_prefix, g_spans
