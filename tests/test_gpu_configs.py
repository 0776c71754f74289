"""GPU parity at the sizes BASELINE.json quotes for configs 2, 3, 4 and 5 (per-GPU share): EXACT over the whole batch
-- every CSR offset and every span of every text -- against the closed form (config 3) or the oracle's C port run over
the whole batch on the host's cores (configs 2, 4, 5; round 4), next to the size-independent properties."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import mojo_regex_amd as M  # noqa: E402
from mojo_regex_amd.workloads import make_digits_batch, make_phone_batch, make_alt_batch  # noqa: E402
from mrx_ref.cfast import CDfa  # noqa: E402  (oracle: checker only)


STREAM_FINDALL = (b"k_stream_findall_fused", b"k_stream_findall", b"k_stream_bits", b"k_stream_findall_rows")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU: the HIP path has no fallback")


def _oracle_exact_check(pat, d, prefix, spans, slab_texts=1 << 18):
    """Every offset and every span of the WHOLE batch against the oracle's C port (DFAEngine.match_all,
    dfa.mojo:2028-2130 as oracle/c/mrx_oracle.c restates it), texts split over the host's cores, in slabs of
    `slab_texts` texts to bound the temporaries: a single wrong offset anywhere in the batch fails.  The oracle
    writes text i's spans where the DEVICE's offsets put them (and reports its own count per text), so one pass
    compares counts, offsets and spans."""
    n, L = d.shape
    cd = CDfa(pat, native=True)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 128))
    assert int(prefix[0].item()) == 0
    checked = 0
    for a in range(0, n, slab_texts):
        b = min(n, a + slab_texts)
        rows = d[a:b].cpu().numpy().reshape(-1)
        offsets = np.arange(0, (b - a + 1) * L, L, dtype=np.int64)
        pre = prefix[a:b + 1].cpu().numpy()
        counts, osp, total = cd.findall_at_mt(rows, offsets, pre, threads)
        dev_counts = (pre[1:] - pre[:-1])
        bad = np.nonzero(dev_counts != counts)[0]
        assert bad.size == 0, (pat, "count of text", a + int(bad[0]), int(dev_counts[bad[0]]), int(counts[bad[0]]))
        have = spans[int(pre[0]):int(pre[-1])].cpu().numpy()
        if not np.array_equal(have, osp):
            k = int(np.nonzero((have != osp).any(axis=1))[0][0])
            t = int(np.searchsorted(pre, pre[0] + k, side="right")) - 1
            raise AssertionError((pat, "span", k, "of text", a + t, have[k].tolist(), osp[k].tolist()))
        checked += total
    assert checked == int(prefix[n].item())
    return checked


def test_config2_exact_over_the_whole_batch():
    """BASELINE.json config 2 (`[a-z]+\\d+`, 2^20 x 1 KiB, SURVEY.md 8(d) mix -- the headline batch): every CSR
    offset and every span of all 2^20 texts equals the oracle's (round 4; rounds 1-3 compared 256 texts)."""
    _need_gpu()
    from mojo_regex_amd.workloads import make_c2_batch
    n, L = 1 << 20, 1024
    pat = b"[a-z]+\\d+"
    d = make_c2_batch(n, L, seed=20260102, device="cuda")
    rx = M.compile_regex(pat)
    prefix, spans, total = rx._dev_findall(M.DeviceBatch.strided(d.reshape(-1), L, length=L), span_cap=n * 32)
    assert M.load_library().mrx_last_kernel_name() in STREAM_FINDALL
    assert _oracle_exact_check(pat, d, prefix, spans) == total > n


def test_config3_digit_runs_exact_at_per_gpu_size():
    """\\d+ findall, 64M x 256 B sharded 8 ways -> 8M x 256 B per GPU; every span of the
    whole batch is checked against the closed form (maximal digit runs)."""
    _need_gpu()
    n, L = 1 << 23, 256
    d = make_digits_batch(n, L, device="cuda")
    rx = M.compile_regex(b"\\d+")
    assert "device.streamable=yes" in rx.describe()
    prefix, spans, total = rx._dev_findall(M.DeviceBatch.strided(d.reshape(-1), L, length=L),
                                           span_cap=n * 8)
    pre = prefix
    checked = 0
    blk = 1 << 20  # rows per block: keeps nonzero() well below 2^31 elements
    for a in range(0, n, blk):
        b = min(n, a + blk)
        isd = (d[a:b] >= 48) & (d[a:b] <= 57)
        prev = torch.zeros_like(isd)
        prev[:, 1:] = isd[:, :-1]
        nxt = torch.zeros_like(isd)
        nxt[:, :-1] = isd[:, 1:]
        starts = (isd & ~prev).nonzero()          # row-major: text order, then position
        ends = (isd & ~nxt).nonzero()
        lo, hi = int(pre[a].item()), int(pre[b].item())
        assert hi - lo == starts.shape[0]
        sp = spans[lo:hi]
        assert bool((sp[:, 0] == starts[:, 1].to(torch.int32)).all())
        assert bool((sp[:, 1] == (ends[:, 1] + 1).to(torch.int32)).all())
        counts = torch.bincount(starts[:, 0], minlength=b - a)
        assert bool(((pre[a + 1:b + 1] - pre[a:b]) == counts).all())
        checked += hi - lo
    assert checked == total and total > 3 * n


def test_config4_phone_groups_at_full_size():
    """(\\d{3})(\\d{3})(\\d{4}), 1M x 1 KiB: findall + search + captures."""
    _need_gpu()
    n, L = 1 << 20, 1024
    pat = b"(\\d{3})(\\d{3})(\\d{4})"
    d = make_phone_batch(n, L, device="cuda")
    rx = M.compile_regex(pat)
    batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
    prefix, spans, total = rx._dev_findall(batch, span_cap=n * 56)
    sp = spans[:total].to(torch.int64)
    counts = prefix[1:] - prefix[:-1]
    owner = torch.repeat_interleave(torch.arange(n, device="cuda"), counts)
    assert bool(((sp[:, 1] - sp[:, 0]) == 10).all())
    flat = d.reshape(-1).to(torch.int64)
    for k in range(10):  # every byte of every match is a digit
        b = flat[owner * L + sp[:, 0] + k]
        assert bool(((b >= 48) & (b <= 57)).all())
    same = owner[1:] == owner[:-1]
    assert bool((sp[1:, 0][same] >= sp[:-1, 1][same]).all())
    assert _oracle_exact_check(pat, d, prefix, spans) == total   # every offset and span of all 2^20 texts
    # search == first findall span; captures at fixed offsets, a18 order (groups, then whole)
    s, e = rx.match_next(batch)
    first = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    has = counts > 0
    first[has] = spans[prefix[:-1][has], 0]
    assert bool((s == first).all())
    # captures of the whole batch on the device (streaming search + fixed offsets)
    cd_ = rx.captures_dev(batch)
    ok = s >= 0
    assert bool((cd_[:, 3, 0] == s).all()) and bool((cd_[:, 3, 1] == e).all())
    assert bool((cd_[ok][:, 0, 0] == s[ok]).all()) and bool((cd_[ok][:, 0, 1] == s[ok] + 3).all())
    assert bool((cd_[ok][:, 1, 1] == s[ok] + 6).all()) and bool((cd_[ok][:, 2, 1] == s[ok] + 10).all())
    assert bool((cd_[~ok] == -1).all())
    sub = d[:4096].cpu().numpy()
    caps = rx.captures([r.tobytes() for r in sub])
    s_h = s[:4096].cpu().numpy()
    for i in range(4096):
        if s_h[i] < 0:
            assert (caps[i] == -1).all()
        else:
            a = int(s_h[i])
            assert caps[i].tolist() == [[a, a + 3], [a + 3, a + 6], [a + 6, a + 10], [a, a + 10]]


def test_config5_alternation_at_full_size():
    """(x|y|foo|bar)+ findall (source-faithful: the '+' is dropped, PARITY-UNPINNED),
    BASELINE.json config 5 at full size: 4M x 4 KiB (16 GiB) in ONE call; the result is
    then checked in blocks of 512K texts to bound the temporaries."""
    _need_gpu()
    n, L = 1 << 22, 4096
    pat = b"(x|y|foo|bar)+"
    d = make_alt_batch(n, L, device="cuda")
    rx = M.compile_regex(pat)
    batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
    lib = M.load_library()
    # both forms of the streaming findall at full size: event rows (what a handle takes once it has seen how full of
    # matches these batches are; checked against the oracle below) and records (round 3's form; must be the same CSR)
    try:
        lib.mrx_debug_dense_rows(2)
        prefix_r, spans_r, total_r = rx._dev_findall(batch, span_cap=n * 720)
        assert lib.mrx_last_kernel_name() in STREAM_FINDALL and lib.mrx_last_kernel_name() != b"k_stream_findall_rows"
        lib.mrx_debug_dense_rows(1)
        prefix, spans, total = rx._dev_findall(batch, span_cap=n * 720)
        assert lib.mrx_last_kernel_name() == b"k_stream_findall_rows"
    finally:
        lib.mrx_debug_dense_rows(0)
    assert total_r == total and torch.equal(prefix, prefix_r) and torch.equal(spans[:total], spans_r[:total])
    del prefix_r, spans_r
    torch.cuda.empty_cache()
    assert total == int(prefix[n].item()) > n * 600
    blk = 1 << 19
    for a in range(0, n, blk):
        pb = prefix[a:a + blk + 1]
        lo, hi = int(pb[0].item()), int(pb[-1].item())
        sp = spans[lo:hi].to(torch.int64)
        counts = pb[1:] - pb[:-1]
        owner = torch.repeat_interleave(torch.arange(blk, device="cuda"), counts)
        ln = sp[:, 1] - sp[:, 0]
        assert bool(((ln == 1) | (ln == 3)).all())
        flat = d[a:a + blk].reshape(-1)
        b0 = flat[owner * L + sp[:, 0]]
        one = ln == 1
        assert bool(((b0[one] == ord("x")) | (b0[one] == ord("y"))).all())
        three = ~one
        b1 = flat[(owner * L + sp[:, 0] + 1)[three]]
        b2 = flat[(owner * L + sp[:, 0] + 2)[three]]
        foo = (b0[three] == ord("f")) & (b1 == ord("o")) & (b2 == ord("o"))
        bar = (b0[three] == ord("b")) & (b1 == ord("a")) & (b2 == ord("r"))
        assert bool((foo | bar).all())
        same = owner[1:] == owner[:-1]
        assert bool((sp[1:, 0][same] >= sp[:-1, 1][same]).all())
        # every x / y byte of the block is the start of a one-byte match
        nxy = int(((d[a:a + blk] == ord("x")) | (d[a:a + blk] == ord("y"))).sum().item())
        assert int(one.sum().item()) == nxy
        del sp, owner, ln, b0, b1, b2, foo, bar, same, one, three
    assert _oracle_exact_check(pat, d, prefix, spans, slab_texts=1 << 18) == total   # all 2^22 texts, 2.9 G spans


def test_config5_lazydfa_semantics_switch():
    """the LazyDFA reading of config 5 ('+' honoured, leftmost-longest)
    stays available behind MRX_COMPILE_LAZYDFA_SEMANTICS and is checked against the oracle's
    NFAMatcher/LazyDFA path; it is never the default."""
    _need_gpu()
    from mrx_ref.hybrid import CompiledRegex as OracleRegex
    n, L = 4096, 512
    pat = b"(x|y|foo|bar)+"
    d = make_alt_batch(n, L, device="cuda")
    rx = M.compile_regex(pat, lazydfa_semantics=True)
    assert "option.lazydfa_semantics=1" in rx.describe() and rx.get_engine_type() == "NFA"
    prefix, spans, total = rx._dev_findall(M.DeviceBatch.strided(d.reshape(-1), L, length=L))
    default_total = M.compile_regex(pat)._dev_findall(M.DeviceBatch.strided(d.reshape(-1), L, length=L))[2]
    assert total < default_total  # runs are merged when the '+' is honoured
    o = OracleRegex(pat, force_nfa=True)
    host, pre, sph = d.cpu().numpy(), prefix.cpu().numpy(), spans.cpu().numpy()
    for i in range(0, n, 16):
        have = [tuple(int(x) for x in r) for r in sph[pre[i]:pre[i + 1]]]
        assert have == o.match_all(host[i].tobytes()), i


def test_long_texts_and_degenerate_batches():
    """Texts far beyond 64 KiB (32-bit positions, unpacked decode tile), a batch of one, and an
    empty batch; C oracle on the long texts."""
    _need_gpu()
    pat = b"[a-z]+\\d+"
    rx = M.compile_regex(pat)
    # empty batch
    e = M.DeviceBatch(torch.zeros(0, dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"))
    pre, sp, tot = rx._dev_findall(e)
    assert tot == 0 and pre.tolist() == [0]
    # three long texts, ragged, CSR (frame form) -- 5 MB, 3 MB + 7 bytes, 70 KB
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    lens = [5 << 20, (3 << 20) + 7, 70000, 0, 1]
    al = torch.tensor(list(b"abcxyz0123456789 -"), dtype=torch.uint8, device="cuda")
    data = al[torch.randint(0, al.numel(), (sum(lens),), generator=g, device="cuda")]
    data[100:200000] = ord("q")               # a 200 KB letter run, then digits: one very long match
    data[200000:200050] = ord("7")
    offsets = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int64, device="cuda")
    batch = M.DeviceBatch(data, offsets)
    pre, sp, tot = rx._dev_findall(batch)
    assert M.load_library().mrx_last_kernel_name() == b"k_stream_findall_pieces"   # few long texts: cut into pieces
    cd = CDfa(pat)
    counts, osp, ototal = cd.findall_batch(data.cpu().numpy(), offsets.cpu().numpy())
    assert tot == ototal
    assert np.array_equal((pre[1:] - pre[:-1]).cpu().numpy(), counts)
    assert np.array_equal(sp[:tot].cpu().numpy(), osp)
    assert int(sp[:tot, 1].max().item()) > 65535
    s, e2 = rx.match_next(batch)
    os_, oe_ = cd.span_batch("search", data.cpu().numpy(), offsets.cpu().numpy())
    assert np.array_equal(s.cpu().numpy(), os_) and np.array_equal(e2.cpu().numpy(), oe_)
    f, fe = rx.match_first(batch)
    of_, ofe_ = cd.span_batch("match_first", data.cpu().numpy(), offsets.cpu().numpy())
    assert np.array_equal(f.cpu().numpy(), of_) and np.array_equal(fe.cpu().numpy(), ofe_)
    # the same first text alone, at a fixed pitch (strided form, one text)
    one = M.DeviceBatch.strided(data[: lens[0]].contiguous(), lens[0], length=lens[0])
    p1, s1, t1 = rx._dev_findall(one)
    assert t1 == int(counts[0]) and np.array_equal(s1[:t1].cpu().numpy(), osp[: counts[0]])


@pytest.mark.parametrize("L", [65500, 65504, 65535, 65536])
def test_record_forms_at_the_16_bit_position_boundary(L):
    """Fixed pitch just below / above the limits of the two-group records (65 500 bytes) and of the
    16-bit decode tile (65 535): spans near the end of the text must come out exact (C oracle)."""
    _need_gpu()
    pat = b"[a-z]+\\d+"
    rx = M.compile_regex(pat)
    g = torch.Generator(device="cuda")
    g.manual_seed(L)
    al = torch.tensor(list(b"abcxyz0123456789 -"), dtype=torch.uint8, device="cuda")
    n = 5
    data = al[torch.randint(0, al.numel(), (n * L,), generator=g, device="cuda")]
    tail = torch.tensor(list(b" zz99"), dtype=torch.uint8, device="cuda")
    for i in range(n):
        data[(i + 1) * L - 5:(i + 1) * L] = tail          # a match that ends exactly at the end of the text
    batch = M.DeviceBatch.strided(data, L, length=L)
    lib = M.load_library()
    offsets = np.arange(0, (n + 1) * L, L, dtype=np.int64)
    counts, osp, ototal = CDfa(pat).findall_batch(data.cpu().numpy(), offsets)
    for mode, kernel in ((0, b"k_stream_findall_pieces"), (2, b"k_stream_findall")):   # cut into pieces / one lane per text
        lib.mrx_debug_long_text_kernels(mode)
        try:
            pre, sp, tot = rx._dev_findall(batch)
        finally:
            lib.mrx_debug_long_text_kernels(0)
        assert lib.mrx_last_kernel_name() == kernel
        assert tot == ototal and np.array_equal((pre[1:] - pre[:-1]).cpu().numpy(), counts)
        assert np.array_equal(sp[:tot].cpu().numpy(), osp)
        assert int(sp[:tot, 1].max().item()) == L
    with_lens = M.DeviceBatch.strided(data, L, length=L, lens=torch.full((n,), L - 3, dtype=torch.int32, device="cuda"))
    lib.mrx_debug_long_text_kernels(2)
    try:
        pre2, sp2, tot2 = rx._dev_findall(with_lens)
    finally:
        lib.mrx_debug_long_text_kernels(0)
    offs2 = np.stack([np.arange(n) * L, np.arange(n) * L + L - 3], axis=1)
    host = data.cpu().numpy()
    packed = np.concatenate([host[a:b] for a, b in offs2])
    c2, o2, t2 = CDfa(pat).findall_batch(packed, np.arange(0, (n + 1) * (L - 3), L - 3, dtype=np.int64))
    assert tot2 == t2 and np.array_equal(sp2[:tot2].cpu().numpy(), o2)


def test_config4_program_on_the_bitset_nfa_kernel_at_full_size():
    """BASELINE config 4, "(\\d{3})(\\d{3})(\\d{4}) capture-group NFA fallback, 1M strings, 1 GPU (bitset-NFA
    kernel)": the pattern's PikeVM program (pikevm.mojo:124-333; 11 positions) as a bitset NFA (k_bscan, and
    k_wstep<., 0, 1> for the walks) -- no determinised table -- over 2^20 x 1 KiB texts.  search, count and findall must
    equal the LazyDFA table kernels' answers on every text (the same function of the text computed two
    ways) and the oracle's NFAMatcher (PikeVM / LazyDFA restatement) on a sample."""
    _need_gpu()
    import sys
    from mrx_ref.hybrid import CompiledRegex as OracleRegex
    pat = b"(\\d{3})(\\d{3})(\\d{4})"
    n, L = 1 << 20, 1024
    d = make_phone_batch(n, L)
    batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
    lib = M.load_library()
    nfa = M.compile_regex(pat, lazydfa_semantics=True, bitset_nfa=True)
    desc = nfa.describe()
    assert "device.bitset=yes positions=11 words=1" in desc and "bitset=1" in desc, desc
    tab = M.compile_regex(pat, lazydfa_semantics=True)
    assert "device.bitset" not in tab.describe()
    # round 3: every match of this program is 10 bytes long, so the union pass takes the match ends by itself
    # (k_bscan modes 2-4); mrx_debug_multiwalk(2) brings the walk per start back -- both against the table kernels
    assert "fixed_len=10" in desc, desc
    for walks in (False, True):
        lib.mrx_debug_multiwalk(2 if walks else 0)
        try:
            s1, e1 = nfa.match_next(batch)
            assert lib.mrx_last_kernel_name() == (b"k_bstep_search" if walks else b"k_bscan_fixed_search")
            c1 = nfa.count(batch)
            assert lib.mrx_last_kernel_name() == (b"k_bstep_count" if walks else b"k_bscan_fixed")
            p1, sp1, t1 = nfa._dev_findall(batch)
            assert lib.mrx_last_kernel_name() == (b"k_bstep_count" if walks else b"k_bscan_fixed")
        finally:
            lib.mrx_debug_multiwalk(0)
        s2, e2 = tab.match_next(batch)
        assert lib.mrx_last_kernel_name() not in (b"k_bstep_search", b"k_bscan_fixed_search")
        assert torch.equal(s1, s2) and torch.equal(e1, e2)
        c2 = tab.count(batch)
        assert torch.equal(c1, c2)
        p2, sp2, t2 = tab._dev_findall(batch)
        assert t1 == t2 and torch.equal(p1, p2) and torch.equal(sp1[:t1], sp2[:t2])
    # oracle sample (NFAMatcher: PikeVM program + LazyDFA, the route the option selects)
    o = OracleRegex(pat, force_nfa=True)
    idx = np.linspace(0, n - 1, 48).astype(np.int64)
    rows = d[torch.from_numpy(idx).cuda()].cpu().numpy()
    pre = p1.cpu().numpy()
    for j, i in enumerate(idx.tolist()):
        t = rows[j].tobytes()
        want = o.match_all(t)
        have = [tuple(int(x) for x in r) for r in sp1[int(pre[i]):int(pre[i + 1])].cpu().numpy()]
        assert have == want, i
        w = o.match_next(t, 0)
        assert (int(s1[i]), int(e1[i])) == (w if w else (-1, -1)), i


@pytest.mark.gpu
def test_default_routes_in_a_fresh_process():
    """The switches of include/mrx_testing.h at their start-up values: a fresh interpreter (no test has touched
    them) must take the kernels DESIGN.md names -- round 2 shipped for a while with `sub` on round 1's kernel
    because two namespace-scope initialisers of the library had been given one body by the compiler."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "import mojo_regex_amd as M\n"
        "from mojo_regex_amd import workloads as W\n"
        "lib = M.load_library()\n"
        "d = W.make_c2_batch(4096, 1024)\n"
        "b = M.DeviceBatch.strided(d.reshape(-1), 1024, length=1024)\n"
        "rx = M.compile_regex(b'[a-z]+\\\\d+')\n"
        "rx.sub_dev(b'#', b, 0, out_cap=4096 * 2048); print('sub', lib.mrx_last_kernel_name().decode())\n"
        "rx._dev_findall(b); print('findall', lib.mrx_last_kernel_name().decode())\n"
        "rx.count(b); print('count', lib.mrx_last_kernel_name().decode())\n"
        "rx.match_next(b); print('search', lib.mrx_last_kernel_name().decode())\n"
        "rx.is_match(b); print('is_match', lib.mrx_last_kernel_name().decode())\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if not k.startswith("MRX_")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    got = dict(line.split() for line in r.stdout.strip().splitlines() if len(line.split()) == 2)
    assert got == {"sub": "k_subs_wave", "findall": "k_stream_findall", "count": "k_stream_count", "search": "k_stream_search",
                   "is_match": "k_is_match_byte"}, got


@pytest.mark.gpu
@pytest.mark.parametrize("pat", [b"[a-z]+\\d+", b"[0-9]{2,4}", b"ab", b"(\\d{3})(\\d{3})(\\d{4})"])
def test_batches_beyond_4_gib(pat):
    """Byte offsets past 2^32 in both layouts: a 64 MiB block (65536 texts x 1 KiB of config 2's mix; for the phone
    pattern config 4's) tiled 80 times = 5 GiB.  count / findall / search of the whole batch are the block's answers,
    tile after tile (fixed pitch and CSR with int64 offsets)."""
    _need_gpu()
    import torch
    from mojo_regex_amd.workloads import make_c2_batch
    T, nb, L = 80, 65536, 1024
    blk = (make_phone_batch(nb, L, device="cuda") if pat.startswith(b"(") else make_c2_batch(nb, L, seed=7, device="cuda")).reshape(nb, L)
    data = blk.repeat(T, 1).reshape(-1)
    n = nb * T
    assert data.numel() > (1 << 32)
    rx = M.compile_regex(pat)
    small = M.DeviceBatch.strided(blk.reshape(-1), L, length=L)
    c0 = rx.count(small)
    p0, s0, t0 = rx._dev_findall(small)
    ss0, se0 = rx.match_next(small)
    assert t0 > 1000
    for big in (M.DeviceBatch.strided(data, L, length=L),
                M.DeviceBatch(data, torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device="cuda"))):
        c = rx.count(big)
        assert torch.equal(c.reshape(T, nb), c0.expand(T, nb))
        p, s, tot = rx._dev_findall(big)
        assert tot == t0 * T
        assert torch.equal(s[:t0], s0[:t0]) and torch.equal(s[tot - t0:tot], s0[:t0])
        assert torch.equal(p[-(nb + 1):] - p[-(nb + 1)], p0)
        assert torch.equal(p[nb * 41:nb * 42 + 1] - p[nb * 41], p0)   # the tile that straddles 2^32 bytes and its neighbours
        ss, se = rx.match_next(big)
        assert torch.equal(ss.reshape(T, nb), ss0.expand(T, nb)) and torch.equal(se.reshape(T, nb), se0.expand(T, nb))
        del c, p, s, ss, se
    # regex.sub: the output passes 2^32 bytes as well (a replacement longer than most matches)
    repl = b"<\\3\\2\\1>" if pat.startswith(b"(") else b"<=====>"
    o0, d0 = rx.sub_dev(repl, small)
    o, d = rx.sub_dev(repl, big, out_cap=int(d0.numel()) * T + 64)
    assert d.numel() == d0.numel() * T   # (past 2^32 bytes for all but the first pattern, whose matches are long)
    assert torch.equal(o[-(nb + 1):] - o[-(nb + 1)], o0)
    assert torch.equal(d[:d0.numel()], d0) and torch.equal(d[-d0.numel():], d0)
    k = min(T - 1, (1 << 32) // d0.numel())   # the tile in which the output passes 2^32 bytes, if it does
    mid = int(o[nb * k])
    assert mid == k * d0.numel() and torch.equal(d[mid:mid + d0.numel()], d0)
