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
            assert lib.mrx_testing_comm_shift(d_pre.data_ptr(), len(pre) - 1, d_meta.data_ptr(), r,
                                              st_p[r].data_ptr(), P, None) == 0
            if len(sp):
                st_s[r, : len(sp)] = torch.from_numpy(sp).cuda()
        T = int(meta[:, 1].sum())
        gp = torch.full((N + 1,), -1, dtype=torch.int64, device="cuda")
        gs = torch.full((max(T, 1), 2), -1, dtype=torch.int32, device="cuda")
        st = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        assert lib.mrx_testing_comm_compact(d_meta.data_ptr(), G, st_p.data_ptr(), P, st_s.data_ptr(), cap,
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
            lib.mrx_testing_comm_compact(d_meta.data_ptr(), G, st_p.data_ptr(), P, st_s.data_ptr(), cap,
                                         gp.data_ptr(), N + 1, gs.data_ptr(), T - 1, st.data_ptr(), None)
            torch.cuda.synchronize()
            assert int(st.item()) == 4
