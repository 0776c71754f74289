"""Results exchange behind the C ABI (include/mrx_comm.h, csrc/mrx_comm.hip): RCCL called directly by the library.

One GPU per box, so the collectives run at world size 1 here (from C: tests/c/comm_example.c, and from Python:
mojo_regex_amd.dist.Comm); the arithmetic of the multi-rank path -- shifting a rank's prefix by the spans before it,
compacting the padded staging into the global CSR -- is checked with several simulated ranks through the testing hooks.
The first 8-GPU run is the driver's (bench.py --gather)."""
import os
import subprocess

import numpy as np
import pytest

import mojo_regex_amd as M
from mojo_regex_amd import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "comm_example")
    libdir = os.path.join(ROOT, "mojo_regex_amd")
    M.load_library()
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "c", "comm_example.c"), "-o", exe,
                    "-L", libdir, "-lmrx_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + libdir + ":/opt/rocm/lib"], check=True)
    return exe


def test_c_comm_client_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_comm_client_output(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = r.stdout.splitlines()
    assert "comm rank=0 size=1" in out and "local total=3" in out and "exact N=3 T=3" in out
    for form in ("exact", "padded"):
        assert form + " text0 [0,8)" in out and form + " text0 [9,17)" in out and form + " text2 [2,6)" in out
    assert "padded status=0 T=3" in out
    assert "search starts 0 -1 2" in out and out[-1] == "done"


@pytest.mark.gpu
def test_comm_world_size_one_from_python():
    import torch
    comm = D.Comm.create(1, 0)
    rng = np.random.default_rng(5)
    texts = [bytes(rng.choice(np.frombuffer(b"ab12 ", dtype=np.uint8), size=int(rng.integers(0, 90))).tolist())
             for _ in range(300)]
    rx = M.compile_regex(b"[a-z]+\\d+")
    batch = M.DeviceBatch.from_texts(texts)
    prefix, spans, total = rx._dev_findall(batch)
    gp, gs = comm.gather_spans(prefix, spans, n_global=len(texts))
    assert torch.equal(gp, prefix) and torch.equal(gs, spans[:total])
    cap = int(total) + 7
    big = torch.zeros((cap, 2), dtype=torch.int32, device="cuda")
    big[:total] = spans[:total]
    gp2, gs2, st = comm.gather_spans(prefix, big, n_global=len(texts), cap_spans_per_rank=cap)
    torch.cuda.synchronize()
    assert int(st.item()) == 0 and torch.equal(gp2, prefix) and torch.equal(gs2[:total], spans[:total])
    # capacity too small: the status word says so and nothing is written
    gp3, gs3, st3 = comm.gather_spans(prefix, big, n_global=len(texts), cap_spans_per_rank=max(1, int(total) - 1))
    torch.cuda.synchronize()
    assert int(st3.item()) == 4   # MRX_E_CAPACITY
    s, e = rx.match_next(batch)
    both = torch.stack([s, e], dim=1).contiguous()
    assert torch.equal(comm.gather_fixed(both), both)
    assert torch.equal(comm.gather_rows(both, rows_cap=len(texts) + 5), both)
    comm.close()


@pytest.mark.gpu
def test_padded_exchange_arithmetic_with_simulated_ranks():
    """k_comm_shift + k_comm_compact on staging filled as ncclAllGather would fill it, for 1..5 ranks with the
    uneven shards a contiguous split leaves (empty ranks and ranks without spans included)."""
    import ctypes as C
    import torch
    lib = M.load_library()
    rng = np.random.default_rng(11)
    for G, N in ((1, 9), (2, 7), (3, 10), (5, 3), (4, 64), (8, 1001)):
        counts = rng.integers(0, 6, size=N)
        if G == 3:
            counts[: N // 3] = 0            # a rank whose texts have no spans at all
        shards = [D.shard_range(N, r, G) for r in range(G)]
        locals_ = []
        for lo, hi in shards:
            c = counts[lo:hi]
            pre = np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
            sp = rng.integers(0, 1000, size=(int(pre[-1]), 2)).astype(np.int32)
            locals_.append((pre, sp))
        meta = np.array([[hi - lo, int(l[0][-1])] for (lo, hi), l in zip(shards, locals_)], dtype=np.int64)
        P = (N + G - 1) // G + 1
        cap = int(meta[:, 1].max()) + 3
        d_meta = torch.from_numpy(meta.reshape(-1)).cuda()
        st_p = torch.zeros((G, P), dtype=torch.int64, device="cuda")
        st_s = torch.full((G, cap, 2), -7, dtype=torch.int32, device="cuda")
        for r, (pre, sp) in enumerate(locals_):
            d_pre = torch.from_numpy(pre).cuda()
            assert lib.mrx_testing_comm_shift(d_pre.data_ptr(), len(pre) - 1, d_meta.data_ptr(), 2, r,
                                              st_p[r].data_ptr(), P, None) == 0
            if len(sp):
                st_s[r, : len(sp)] = torch.from_numpy(sp).cuda()
        T = int(meta[:, 1].sum())
        gp = torch.full((N + 1,), -1, dtype=torch.int64, device="cuda")
        gs = torch.full((max(T, 1), 2), -1, dtype=torch.int32, device="cuda")
        st = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        assert lib.mrx_testing_comm_compact(d_meta.data_ptr(), 2, G, st_p.data_ptr(), P, st_s.data_ptr(), cap,
                                            gp.data_ptr(), N + 1, gs.data_ptr(), max(T, 1), st.data_ptr(), None) == 0
        torch.cuda.synchronize()
        want_p = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        want_s = np.concatenate([l[1] for l in locals_]) if T else np.zeros((0, 2), np.int32)
        assert int(st.item()) == 0, (G, N)
        assert (gp.cpu().numpy() == want_p).all(), (G, N)
        assert (gs.cpu().numpy()[:T] == want_s).all(), (G, N)
        # an output buffer one span short: status says so
        if T > 1:
            st.fill_(-1)
            lib.mrx_testing_comm_compact(d_meta.data_ptr(), 2, G, st_p.data_ptr(), P, st_s.data_ptr(), cap,
                                         gp.data_ptr(), N + 1, gs.data_ptr(), T - 1, st.data_ptr(), None)
            torch.cuda.synchronize()
            assert int(st.item()) == 4


@pytest.mark.gpu
def test_every_rank_takes_the_same_decision_from_the_gathered_size_words():
    """The words every rank gathers before anything else moves carry each rank's texts (-1: its own arguments were
    invalid), spans and OUTPUT CAPACITIES, and the compaction decides from all of them: an invalid rank, or a rank whose
    buffers are too small, makes every rank report the same status and write nothing (round 3 decided from a rank's own
    capacities: one rank returning alone leaves the others blocked in the next collective)."""
    import torch
    lib = M.load_library()
    G, N = 4, 37
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 5, size=N)
    shards = [D.shard_range(N, r, G) for r in range(G)]
    pres = [np.concatenate([[0], np.cumsum(counts[lo:hi])]).astype(np.int64) for lo, hi in shards]
    T = int(counts.sum())
    P = (N + G - 1) // G + 1
    cap = max(int(p[-1]) for p in pres) + 2
    def run(meta4):
        d_meta = torch.from_numpy(np.asarray(meta4, dtype=np.int64).reshape(-1)).cuda()
        st_p = torch.zeros((G, P), dtype=torch.int64, device="cuda")
        st_s = torch.zeros((G, cap, 2), dtype=torch.int32, device="cuda")
        for r, pre in enumerate(pres):
            if meta4[r][0] >= 0:
                assert lib.mrx_testing_comm_shift(torch.from_numpy(pre).cuda().data_ptr(), len(pre) - 1, d_meta.data_ptr(), 4, r,
                                                  st_p[r].data_ptr(), P, None) == 0
        gp = torch.full((N + 1,), -1, dtype=torch.int64, device="cuda")
        gs = torch.full((max(T, 1), 2), -1, dtype=torch.int32, device="cuda")
        st = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        assert lib.mrx_testing_comm_compact(d_meta.data_ptr(), 4, G, st_p.data_ptr(), P, st_s.data_ptr(), cap, gp.data_ptr(), N + 1,
                                            gs.data_ptr(), max(T, 1), st.data_ptr(), None) == 0
        torch.cuda.synchronize()
        return int(st.item()), gp.cpu().numpy(), gs.cpu().numpy()
    good = [[hi - lo, int(p[-1]), N + 1, T] for (lo, hi), p in zip(shards, pres)]
    st, gp, _ = run(good)
    assert st == 0 and (gp == np.concatenate([[0], np.cumsum(counts)])).all()
    for what, meta in (("rank 2's spans buffer one short", [m if r != 2 else [m[0], m[1], m[2], T - 1] for r, m in enumerate(good)]),
                       ("rank 3's offsets buffer one short", [m if r != 3 else [m[0], m[1], N, m[3]] for r, m in enumerate(good)])):
        st, gp, gs = run(meta)
        assert st == 4 and (gp == -1).all() and (gs == -1).all(), what     # MRX_E_CAPACITY, nothing written
    bad = [m if r != 1 else [-1, 0, m[2], m[3]] for r, m in enumerate(good)]
    st, gp, gs = run(bad)
    assert st == 5 and (gp == -1).all() and (gs == -1).all()               # MRX_E_ARGUMENT, nothing written


@pytest.mark.gpu
def test_padded_exchange_at_config3_size_with_eight_simulated_ranks():
    """BASELINE.json config 3 in its defining form -- 64M x 256 B over 8 GPUs, ~4.5 spans per text -- through the
    padded exchange's compaction with eight simulated ranks at the REAL sizes: 8 x 8M texts, 8 x ~36M spans, 2.3 GB of
    span staging and 0.5 GB of offset staging per rank, a 2.3 GB global span array: the 64-bit indexing and the
    memory plan of bench.py's `config3` leg, which no 8-GPU node has run yet."""
    import torch
    lib = M.load_library()
    G, n_r = 8, 1 << 23
    N = G * n_r
    g = torch.Generator(device="cuda")
    g.manual_seed(8)
    counts = torch.randint(0, 10, (N,), generator=g, device="cuda", dtype=torch.int64)     # mean 4.5 spans per text
    per_rank = counts.view(G, n_r).sum(dim=1)
    T = int(per_rank.sum().item())
    cap = int(per_rank.max().item()) + 1024
    assert T > 280_000_000 and cap * 8 * G > 2_300_000_000
    P = (N + G - 1) // G + 1
    meta = torch.stack([torch.full((G,), n_r, dtype=torch.int64, device="cuda"), per_rank,
                        torch.full((G,), N + 1, dtype=torch.int64, device="cuda"),
                        torch.full((G,), T, dtype=torch.int64, device="cuda")], dim=1).contiguous()
    st_p = torch.zeros((G, P), dtype=torch.int64, device="cuda")
    st_s = torch.empty((G, cap, 2), dtype=torch.int32, device="cuda")
    base = 0
    for r in range(G):
        pre = torch.zeros(n_r + 1, dtype=torch.int64, device="cuda")
        torch.cumsum(counts[r * n_r:(r + 1) * n_r], 0, out=pre[1:])
        assert lib.mrx_testing_comm_shift(pre.data_ptr(), n_r, meta.data_ptr(), 4, r, st_p[r].data_ptr(), P, None) == 0
        k = int(per_rank[r].item())
        idx = torch.arange(base, base + k, device="cuda", dtype=torch.int64)    # span j of the job carries j: (j mod 2^31, j >> 31)
        st_s[r, :k, 0] = (idx & 0x7FFFFFFF).to(torch.int32)
        st_s[r, :k, 1] = (idx >> 31).to(torch.int32)
        st_s[r, k:] = -7
        base += k
        del pre, idx
    gp = torch.empty(N + 1, dtype=torch.int64, device="cuda")
    gs = torch.empty((T, 2), dtype=torch.int32, device="cuda")
    st = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    assert lib.mrx_testing_comm_compact(meta.data_ptr(), 4, G, st_p.data_ptr(), P, st_s.data_ptr(), cap, gp.data_ptr(), N + 1,
                                        gs.data_ptr(), T, st.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(st.item()) == 0
    del st_s, st_p
    want = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(counts, 0, out=want[1:])
    assert torch.equal(gp, want)
    j = torch.arange(0, T, device="cuda", dtype=torch.int64)
    assert bool((gs[:, 0].to(torch.int64) == (j & 0x7FFFFFFF)).all()) and bool((gs[:, 1].to(torch.int64) == (j >> 31)).all())


@pytest.mark.gpu
def test_library_communicator_beside_a_torch_nccl_process_group():
    """bench.py's N > 1 legs hold TWO RCCL communicators in one process: torch's process group and the library's
    (mojo_regex_amd.dist.Comm).  Rehearsed at world size 1 in a child process: the library's communicator is created
    and destroyed while the process group is alive, and their collectives are interleaved on one stream."""
    import sys
    code = (
        "import os, sys, socket; sys.path.insert(0, %r)\n"
        "import torch, torch.distributed as dist\n"
        "import mojo_regex_amd as M\n"
        "from mojo_regex_amd import dist as D\n"
        "s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "x = torch.ones(1024, device='cuda'); dist.all_reduce(x)\n"
        "rx = M.compile_regex(b'[a-z]+\\\\d+')\n"
        "batch = M.DeviceBatch.from_texts([b'ab12 cd3', b'', b'zz9'] * 50)\n"
        "prefix, spans, total = rx._dev_findall(batch)\n"
        "for rounds in range(3):\n"
        "    comm = D.Comm.create(1, 0)\n"
        "    comm.reserve_spans(batch.n, total + 8)\n"
        "    for k in range(4):\n"
        "        dist.all_reduce(x)\n"
        "        big = torch.zeros((total + 8, 2), dtype=torch.int32, device='cuda'); big[:total] = spans[:total]\n"
        "        gp, gs, st = comm.gather_spans(prefix, big, n_global=batch.n, cap_spans_per_rank=total + 8)\n"
        "        dist.all_reduce(x)\n"
        "        gp2, gs2 = comm.gather_spans(prefix, spans, n_global=batch.n)\n"
        "        torch.cuda.synchronize()\n"
        "        assert int(st.item()) == 0 and torch.equal(gp, prefix) and torch.equal(gs[:total], spans[:total])\n"
        "        assert torch.equal(gp2, prefix) and torch.equal(gs2, spans[:total])\n"
        "    comm.close()\n"
        "    dist.all_reduce(x)\n"
        "torch.cuda.synchronize()\n"
        "assert float(x[0]) == 1.0\n"
        "dist.destroy_process_group()\n"
        "print('both communicators ok')\n" % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "both communicators ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
