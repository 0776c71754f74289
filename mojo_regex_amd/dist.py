"""Multi-GPU plumbing for the batch matcher: one process per GPU, texts sharded
by contiguous index ranges, compiled tables replicated, NO collective inside the
scan (every text is matched independently -- SURVEY.md 8(e)).  Results stay
sharded by default.  The one optional exchange step is "results only":
gather_fixed() (match_first / search / captures: fixed bytes per text, plain
all-gather) and gather_spans() (findall: per-rank totals first, then an
all-gatherv into prefix-sum offsets).  torch.distributed backend "nccl" (= RCCL
over xGMI) on GPUs, "gloo" in the CPU tests.

On GPUs the exchange itself runs behind the C ABI (include/mrx_comm.h, csrc/mrx_comm.hip:
RCCL called directly, the same entry points a Mojo host binds): `Comm` below wraps it; the
torch.distributed process group only carries the 128-byte communicator id and the barriers.
The torch-only implementations further down remain for gloo (CPU tests, shared-GPU rehearsal).
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


# ---- results exchange behind the C ABI (include/mrx_comm.h) --------------------------
class Comm:
    """mrx_comm: an RCCL communicator owned by libmrx_hip.so.  Comm.create(world, rank) is collective:
    rank 0 makes the id (mrx_comm_unique_id) and the process group broadcasts its 128 bytes."""

    def __init__(self, handle, world: int, rank: int):
        self._h, self.world, self.rank = handle, world, rank

    @classmethod
    def create(cls, world: int, rank: int):
        import ctypes as C
        import torch
        from .api import load_library, _check
        lib = load_library()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _check(lib.mrx_comm_unique_id(ident))
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor(list(ident), dtype=torch.uint8)
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0)
            ident = (C.c_uint8 * 128)(*t.cpu().tolist())
        h = C.c_void_p()
        _check(lib.mrx_comm_init(ident, world, rank, C.byref(h)))
        return cls(h, world, rank)

    def close(self):
        if self._h:
            from .api import load_library
            load_library().mrx_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve_spans(self, n_global: int, cap_spans_per_rank: int):
        """Allocates the staging of the padded gather_spans now (mrx_comm_reserve), so that no call allocates or
        synchronises the device later."""
        from .api import load_library, _check
        lib = load_library()
        _check(lib.mrx_comm_reserve(self._h, lib.mrx_comm_spans_staging_bytes(self._h, n_global, cap_spans_per_rank)))

    def gather_fixed(self, local):
        """Equal shards: [world * n_local, ...] in rank order (one ncclAllGather, no host synchronisation)."""
        import torch
        from .api import load_library, _check
        local = local.contiguous()
        out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        _check(load_library().mrx_allgather_fixed(self._h, local.data_ptr(), out.data_ptr(),
                                                  local.numel() * local.element_size(),
                                                  torch.cuda.current_stream().cuda_stream))
        return out

    def gather_rows(self, local, rows_cap: int):
        """Ragged shards of fixed-width rows (exact form: one 8-byte-per-rank read-back)."""
        import ctypes as C
        import torch
        from .api import load_library, _check
        local = local.contiguous()
        row_bytes = local.element_size()
        for d in local.shape[1:]:
            row_bytes *= int(d)
        out = torch.empty((rows_cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        total = C.c_int64(0)
        _check(load_library().mrx_allgatherv_rows(self._h, local.data_ptr(), local.shape[0], row_bytes, out.data_ptr(),
                                                  rows_cap, C.byref(total), torch.cuda.current_stream().cuda_stream))
        return out[: total.value]

    def gather_spans(self, prefix, spans, n_global: int, cap_spans_per_rank: int = 0, out=None):
        """findall CSR of all ranks on every rank.  cap_spans_per_rank = 0: exact form (sizes read back once);
        > 0: padded form, nothing is read back -- returns (gprefix[n_global + 1], gspans[world * cap], status)
        with the span total in gprefix[n_global] and int32 status on the device (0 = ok)."""
        import ctypes as C
        import torch
        from .api import load_library, _check
        lib = load_library()
        n_local = int(prefix.shape[0]) - 1
        dev = prefix.device
        stream = torch.cuda.current_stream().cuda_stream
        if cap_spans_per_rank > 0:
            if out is None:
                out = (torch.empty(n_global + 1, dtype=torch.int64, device=dev),
                       torch.empty((self.world * cap_spans_per_rank, 2), dtype=torch.int32, device=dev),
                       torch.zeros(1, dtype=torch.int32, device=dev))
            gp, gs, st = out
            assert spans.shape[0] >= cap_spans_per_rank, "the local span buffer must hold cap_spans_per_rank slots"
            _check(lib.mrx_allgatherv_spans(self._h, prefix.data_ptr(), n_local, spans.data_ptr(), cap_spans_per_rank,
                                            n_global, gp.data_ptr(), gp.shape[0], gs.data_ptr(), gs.shape[0], None, None,
                                            st.data_ptr(), stream))
            return gp, gs, st
        # exact: capacities are known only after the sizes are back, so gather into generous buffers once
        if out is None:
            tot_local = int(prefix[n_local].item())
            t = torch.tensor([tot_local], dtype=torch.int64, device=dev)
            allt = self.gather_fixed(t)
            out = (torch.empty(n_global + 1, dtype=torch.int64, device=dev),
                   torch.empty((max(1, int(allt.sum().item())), 2), dtype=torch.int32, device=dev))
        gp, gs = out[0], out[1]
        N, T = C.c_int64(0), C.c_int64(0)
        _check(lib.mrx_allgatherv_spans(self._h, prefix.data_ptr(), n_local, spans.data_ptr(), 0, n_global,
                                        gp.data_ptr(), gp.shape[0], gs.data_ptr(), gs.shape[0], C.byref(N), C.byref(T),
                                        None, stream))
        return gp[: N.value + 1], gs[: T.value]


# ---- results exchange over torch.distributed (gloo rehearsal) -----------
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
