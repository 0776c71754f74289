"""CPU: the N>1 plumbing (rendezvous, sharding, max-over-ranks timing, summed
units) with world_size 2 over gloo.  Matching itself needs a GPU and is covered
by the gpu-marked tests; nothing here calls into the HIP library."""
import os
import socket
import subprocess
import sys
import textwrap

from mojo_regex_amd.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_in_rank_order():
    for n in (0, 1, 7, 64, 1 << 20, (1 << 20) + 3):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    from mojo_regex_amd import dist
    rank, local, world = dist.env_world()
    d = dist.init("gloo")
    lo, hi = dist.shard_range(1000, rank, world)
    dist.barrier(world)
    out = dist.combine(world, elapsed_s=0.5 + rank, units={"bytes": (hi - lo) * 1024.0, "matches": 10.0 * (rank + 1)})
    dist.barrier(world)
    if rank == 0:
        print("RESULT " + json.dumps(out))
""")


def test_two_rank_gloo_combine(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT ")][0]
    import json
    r = json.loads(line[7:])
    assert r["elapsed_s"] == 1.5          # MAX over ranks
    assert r["bytes"] == 1000 * 1024.0    # SUM over ranks
    assert r["matches"] == 30.0


GATHER_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch
    from mojo_regex_amd import dist
    rank, local, world = dist.env_world()
    dist.init("gloo")
    N = 11                                   # uneven shards: 6 + 5
    lo, hi = dist.shard_range(N, rank, world)
    # text i has i %% 4 spans; span k of text i is (100*i + k, 100*i + k + 1)
    counts = torch.tensor([i %% 4 for i in range(lo, hi)], dtype=torch.int64)
    prefix = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)])
    rows = [(100 * i + k, 100 * i + k + 1) for i in range(lo, hi) for k in range(i %% 4)]
    spans = torch.tensor(rows + [(-7, -7)] * 3, dtype=torch.int32).reshape(-1, 2)   # slack past total
    g_prefix, g_spans = dist.gather_spans(world, prefix, spans, len(rows))
    first = torch.tensor([[i, i + 1] for i in range(lo, hi)], dtype=torch.int32)
    g_first = dist.gather_fixed(world, first)
    print("RESULT " + json.dumps({"rank": rank, "prefix": g_prefix.tolist(), "spans": g_spans.tolist(),
                                  "first": g_first.tolist()}))
    dist.barrier(world)
""")


def test_two_rank_gloo_results_gather(tmp_path):
    """8(e) exchange step: every rank ends up with the whole batch's CSR in global order."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "gworker.py"
    script.write_text(GATHER_WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    want_counts = [i % 4 for i in range(11)]
    want_prefix = [0]
    for c in want_counts:
        want_prefix.append(want_prefix[-1] + c)
    want_spans = [[100 * i + k, 100 * i + k + 1] for i in range(11) for k in range(i % 4)]
    for o in outs:
        r = json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:])
        assert r["prefix"] == want_prefix
        assert r["spans"] == want_spans
        assert r["first"] == [[i, i + 1] for i in range(11)]
