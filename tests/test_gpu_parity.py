"""GPU parity: the HIP path (through the C ABI) against the oracle.

Bit-exact is the bar: every start/end offset, every count, every replaced byte.
  * the reference's own known-answer vectors (tests/golden) through the GPU;
  * seeded random batches (ragged, empty, high bytes) for a pattern set that
    covers every kernel plan, compared with the oracle text by text;
  * the streaming kernel against the generic kernel and the oracle, with pitches
    and lengths that are not multiples of the chunk size;
  * at BASELINE.json's full size (1M x 1KiB) through size-independent properties.
"""
import collections
import json
import os
import re
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import mojo_regex_amd as M  # noqa: E402
from mrx_ref import hybrid as O  # noqa: E402  (oracle: checker only)
from mrx_ref import UnsupportedByOracle, RegexSyntaxError as OracleSyntaxError  # noqa: E402
from vector_eval import load_vectors, evaluate, Unsupported  # noqa: E402
from mojo_regex_amd.workloads import make_c2_batch  # noqa: E402


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU: the HIP path has no fallback")


class ProductBackend:
    """tests/vector_eval.py backend that answers from the GPU, one text per call."""
    name = "hip"

    def _rx(self, p):
        return M.compile_regex(p)

    def _one(self, pair):
        s, e = int(pair[0][0]), int(pair[1][0])
        return None if s < 0 else (s, e)

    def match_first(self, p, t):
        return self._one(self._rx(p).match_first([t]))

    def search(self, p, t):
        return self._one(self._rx(p).match_next([t]))

    def findall(self, p, t):
        return self._rx(p).findall_lists([t])[0]

    def sub(self, p, r, t, count=0):
        return self._rx(p).sub(r, [t], count)[0]

    def split(self, p, t, maxsplit=0):
        return M.split(p, [t], maxsplit)[0]

    # regex.nfa's module functions (tests/test_nfa.mojo): NFAEngine as the Engine, MRX_COMPILE_NFA_ENGINE
    def nfa_match_first(self, p, t):
        r = self._one(M.compile_regex(p, nfa_engine=True).match_first([t]))
        return r if (r is not None and r[0] == 0) else None   # nfa.mojo:1764-1769

    def nfa_findall(self, p, t):
        return M.compile_regex(p, nfa_engine=True).findall_lists([t])[0]

    def obj_match_first(self, p, t, start=0):
        return self._one(self._rx(p).match_first_at([t], start))   # engine-level match_first(text, start)

    def obj_match_next(self, p, t, start=0):
        return self._one(self._rx(p).match_next_at([t], start))

    def obj_test(self, p, t):
        return bool(self._rx(p).test([t])[0])

    def obj_is_match(self, p, t, start=0):
        return bool(self._rx(p).is_match_at([t], start)[0])

    def engine_type(self, p):
        return self._rx(p).get_engine_type()

    def stats(self, p):
        return self._rx(p).get_stats()

    # comptime API (comptime_regex.mojo:59-87, 176-233): the DFAEngine of compile_dfa_pattern when that
    # compiles (MRX_COMPILE_DFA_ENGINE), the runtime API when it raises
    def _ct_rx(self, p):
        try:
            return M.compile_regex(p, dfa_engine=True)
        except M.UnsupportedPattern:
            return self._rx(p)

    def ct_search(self, p, t):
        return self._one(self._ct_rx(p).match_next([t]))

    def ct_match_first(self, p, t):
        return self._one(self._ct_rx(p).match_first([t]))

    def ct_findall(self, p, t):
        return self._ct_rx(p).findall_lists([t])[0]

    def _dfa_rx(self, build):
        """DFAEngine built directly (tests/test_dfa.mojo) = MRX_COMPILE_DFA_ENGINE on the pattern whose
        shape compiler makes the same call."""
        if build["kind"] == "literal":
            # DFAEngine.compile_pattern(literal, anchors) is what compile_dfa_pattern builds for ^literal$
            # (dfa.mojo:2385-2410)
            if build["literal"] and not build["literal"].isalnum():
                raise Unsupported("direct DFAEngine construction is not an ABI entry point")
            p = (("^" if build.get("start_anchor") else "") + build["literal"] +
                 ("$" if build.get("end_anchor") else "")).encode()
            rx = M.compile_regex(p, dfa_engine=True)
            d = rx.describe()
            if "dfa.shape=literal\n" not in d or ("dfa.literal=%s\n" % build["literal"].encode().hex()) not in d:
                raise Unsupported("literal does not take the DFA literal route")
            return rx
        if build["kind"] == "char_class":
            # compile_character_class(chars, min, max) is what the single-class compiler calls for [chars]{min,max}
            # (dfa.mojo:375-496 via :2420-2440)
            q = {(1, -1): "+", (0, -1): "*", (0, 1): "?", (1, 1): ""}.get((build["min"], build["max"]))
            if q is None or not build["char_class"].isalnum():
                raise Unsupported("direct DFAEngine construction is not an ABI entry point")
            return M.compile_regex(("[" + build["char_class"] + "]" + q).encode(), dfa_engine=True)
        return M.compile_regex(build["pattern"].encode(), dfa_engine=True)

    def dfa_match_first(self, build, t, start=0):
        # DFAEngine.match_first(text, start) == the engine-level operation of the ABI
        return self._one(self._dfa_rx(build).match_first_at([t], start))

    def dfa_match_next(self, build, t, start=0):
        return self._one(self._dfa_rx(build).match_next_at([t], start))

    def dfa_match_all(self, build, t):
        return self._dfa_rx(build).findall_lists([t])[0]


def _vector_key(v):
    return "%s:%d %s" % (v["file"], v["line"], v["op"])


def test_reference_vectors_through_the_gpu():
    """Every transcribed reference vector (tests/golden/reference_vectors.json, the 104 of
    tests/test_nfa.mojo included: NFAEngine through MRX_COMPILE_NFA_ENGINE) through the HIP path.
    The vectors the product does not answer are pinned one by one in
    tests/golden/gpu_vector_skips.json (vector -> reason); anything else must pass."""
    _need_gpu()
    be = ProductBackend()
    failures, skipped, passed = [], {}, 0
    vecs = load_vectors()
    for v in vecs:
        try:
            f = evaluate(be, v)
        except (M.UnsupportedPattern, Unsupported) as e:
            skipped[_vector_key(v)] = str(e)[:100]
            continue
        if f:
            failures.extend(f)
        else:
            passed += 1
    dump = os.environ.get("MRX_WRITE_SKIPS")
    if dump:
        with open(dump, "w") as fh:
            json.dump({"note": "reference vectors the HIP path does not answer, with the reason it gives",
                       "skips": skipped}, fh, indent=1, sort_keys=True)
    assert not failures, "\n".join(failures[:25])
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gpu_vector_skips.json")))["skips"]
    assert set(skipped) == set(want), (sorted(set(skipped) - set(want)), sorted(set(want) - set(skipped)))
    assert passed == len(vecs) - len(want)


def test_hand_traced_backtracker_quirks_through_the_gpu():
    """tests/golden/backtrack_quirk_vectors.json: NFAEngine quirks traced by hand through nfa.mojo (each vector
    cites the deciding lines) -- the flat program must give the traced answer, not merely agree with the oracle."""
    _need_gpu()
    be = ProductBackend()
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "backtrack_quirk_vectors.json")))
    for v in doc["vectors"]:
        got = getattr(be, v["op"])(v["pattern"].encode("latin-1"), v["text"].encode("latin-1"))
        if v["op"] == "nfa_findall":
            assert [list(x) for x in got] == v["want"], (v, got)
        else:
            assert (list(got) if got is not None else None) == v["want"], (v, got)


PATTERNS = [
    b"hello", b"a", b"aa", b"[a-z]+\\d+", b"[a-z]+[0-9]+", b"\\d+", b"[0-9]+", b"[a-z]+", b"[0-9]*",
    b"[a-z]{3}", b"[0-9]{2,4}", b"[a-z]{2,}", b"[^0-9]+", b"[^abc]+", b"(\\d{3})(\\d{3})(\\d{4})",
    b"(x|y|foo|bar)+", b"(x|y|foo|bar)+z", b"a|b|c", b"(a|b)", b"(a|b)x", b"a+b", b"a+b*", b"(abc)+",
    b"(abc)*", b"(ab)?", b"[A-Z][a-z]+[0-9]+", b"[a-z]*[0-9]+", b"[0-9]+\\.?[0-9]*", b"\\w+@\\w+\\.com",
    b"[a-zA-Z0-9._%+-]+@[a-zA-Z0-9.-]+\\.[a-z]{2,}", b"\\d{3}-\\d{4}", b".*", b".+", b".", b"^a", b"a$",
    b"^abc$", b"", b"^", b"(hello|help|helicopter)", b"(cat|dog)+", b"3[02]|40|[68]9",
    b"(?:00|33|44)\\d{3}", b"hello world this is long", b"x[0-9]{2,4}y", b"\\s*\\d+", b"[a-c]+[x-z]?",
    b"((a|b)|(c|d))", b"a**", b"[a|b]", b"\\d+\\s\\w+",
]

ALPHABETS = {
    "words": b"abcxyzfor 0189-.@helpcatdog\n",
    "digits": b"0123456789 ab",
    "printable": bytes(range(32, 127)),
    "bytes": bytes(range(256)),
}


import contextlib

# findall of a streamable plan: one launch (scan + CSR offsets + spans fused), or scan -> sums -> decode
STREAM_FINDALL = (b"k_stream_findall_fused", b"k_stream_findall", b"k_stream_bits", b"k_stream_findall_rows")


@contextlib.contextmanager
def fused_findall(mode=2):
    """The streaming findall as ONE launch (scan + look-back + record expansion) instead of the default
    scan -> prefix sums -> decode; mode 2 = for short texts too."""
    lib = M.load_library()
    lib.mrx_debug_fused_findall(mode)
    try:
        yield
    finally:
        lib.mrx_debug_fused_findall(0)


@contextlib.contextmanager
def stream_bits(on):
    """findall of short fixed-pitch texts as ONE launch with the event bits in registers (k_stream_bits, opt-in)
    or, 0 = the default, as scan -> prefix sums -> decode."""
    lib = M.load_library()
    lib.mrx_debug_stream_bits(1 if on else 0)
    try:
        yield
    finally:
        lib.mrx_debug_stream_bits(0)


@contextlib.contextmanager
def generic_kernels():
    """Route calls to the generic lane-per-text kernels (the second implementation the
    streaming kernel is compared with)."""
    lib = M.load_library()
    lib.mrx_debug_force_generic(2)   # 2 = the literal restatement of the reference's loops only
    try:
        yield
    finally:
        lib.mrx_debug_force_generic(0)


@contextlib.contextmanager
def no_streaming_kernels():
    """Level 1: everything but the streaming kernel (the flattened k_step_* kernels stay on)."""
    lib = M.load_library()
    lib.mrx_debug_force_generic(1)
    try:
        yield
    finally:
        lib.mrx_debug_force_generic(0)


@contextlib.contextmanager
def multiwalk(mode):
    """2 = stepper plans never take their multi-walk form (k_mwalk: several walks side by side in one pass), 3 = k_mwalk
    without its packed-start form, 0 = where the plan has one (the default)."""
    lib = M.load_library()
    lib.mrx_debug_multiwalk(mode)
    try:
        yield
    finally:
        lib.mrx_debug_multiwalk(0)


@contextlib.contextmanager
def long_text_kernels(mode):
    """1 = always the long-text treatments (pieces where the plan has synchronising bytes, else a wavefront per text),
    2 = never, 3 = as 1 but stepper plans on the wavefront-per-text kernel (default: by average text length)."""
    lib = M.load_library()
    lib.mrx_debug_long_text_kernels(mode)
    try:
        yield
    finally:
        lib.mrx_debug_long_text_kernels(0)


def _random_texts(rng, n, max_len, alphabet):
    al = np.frombuffer(alphabet, dtype=np.uint8)
    lens = rng.integers(0, max_len + 1, size=n)
    lens[: min(n, 3)] = [0, 1, max_len][: min(n, 3)]
    return [bytes(rng.choice(al, size=int(k)).tolist()) for k in lens]


@pytest.mark.parametrize("pat", PATTERNS)
def test_random_batches_match_oracle(pat):
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat))
    texts = []
    for name, al in ALPHABETS.items():
        texts += _random_texts(rng, 60, 96, al)
    texts += [b"hello123 world456 test789", b"xyfoo", b"6502530000 and 4155551234", b"abcabcabc",
              b"user@example.com, other@test.org", b"x123y x1y x12345y"]
    try:
        rx = M.compile_regex(pat)
    except M.RegexSyntaxError as err:
        # the reference's parser raises on this pattern: same message from both sides
        with pytest.raises(OracleSyntaxError) as oe:
            O.compile_regex(pat)
        assert str(oe.value) == str(err)
        return
    try:
        got_all = rx.findall_lists(texts)
        s, e = rx.match_next(texts)
    except M.UnsupportedPattern as err:
        if "per-text transition cache is tracked for at most 64" in str(err):
            return   # '$' on the LazyDFA search beyond the product's 64 states (the oracle has no such limit)
        # must be out of scope for the oracle too (same routing)
        with pytest.raises(UnsupportedByOracle):
            O.findall(pat, b"abc")
        return
    for i, t in enumerate(texts):
        assert got_all[i] == O.findall(pat, t), (pat, t)
        w = O.search(pat, t)
        assert (int(s[i]), int(e[i])) == (w if w else (-1, -1)), (pat, t)
    try:
        fs, fe = rx.match_first(texts)
        fl = rx.is_match(texts)
    except M.UnsupportedPattern:
        with pytest.raises(UnsupportedByOracle):
            O.match_first(pat, b"abc")
        return
    c = O.compile_regex(pat)
    for i, t in enumerate(texts):
        w = O.match_first(pat, t)
        assert (int(fs[i]), int(fe[i])) == (w if w else (-1, -1)), (pat, t)
        assert bool(fl[i]) == c.is_match(t, 0), (pat, t)


SUB_CASES = [
    (b"(\\d{3})(\\d{3})(\\d{4})", b"\\1-\\2-\\3"), (b"(\\d{3})(\\d{4})", b"\\1-\\2"),
    (b"(\\d{2})(\\d{2})", b"\\2\\1"), (b"(\\d{4})-(\\d{2})-(\\d{2})", b"\\2/\\3/\\1"),
    (b"[0-9]+", b"NUM"), (b"hello", b"\\0hi"), (b"a*", b"-"), (b"\\d+", b"<\\1>"), (b"[a-z]+\\d+", b""),
    (b".*", b"X"), (b"(x|y|foo|bar)+", b"_"),
]


@pytest.mark.parametrize("pat,repl", SUB_CASES)
@pytest.mark.parametrize("count", [0, 1, 2])
def test_sub_matches_oracle(pat, repl, count):
    _need_gpu()
    rng = np.random.default_rng(7)
    texts = _random_texts(rng, 80, 64, b"0123456789 -ab") + [
        b"6502530000", b"6502530000 and 4155551234", b"5551234 and 9876543", b"1234",
        b"Date: 2026-04-12 is today", b"hello world", b"", b"abc123def456", b"xyfoo bar"]
    try:
        got = M.compile_regex(pat).sub(repl, texts, count)
    except M.UnsupportedPattern as err:
        if "per-text transition cache is tracked for at most 64" in str(err):
            return
        with pytest.raises(UnsupportedByOracle):
            O.sub(pat, repl, b"abc 123", count)
        return
    for t, g in zip(texts, got):
        assert g == O.sub(pat, repl, t, count), (pat, repl, t, count)


@pytest.mark.parametrize("pat", [b"x(\\d)?", b"(\\d)?", b"(\\d)*", b"^(\\d)+", b"(\\d)+", b"ab(\\d)?(\\d)?", b"(\\d)(\\d)?x?",
                                 b"(\\d{3})(\\d{3})(\\d{4})", b"ab(\\d)*", b"(\\d{2})-(\\d)?", b"(\\d)*a", b"(\\d)(\\d)x", b"(\\d{2})+"])
def test_fixed_width_group_windows_end_with_the_text(pat):
    """CompiledRegex._try_precompute_fixed_sub takes the group widths from the pattern text, quantifiers ignored
    (matcher.mojo:1002-1035), and _apply_template_fixed copies text[match start + offset ...] unchecked
    (matcher.mojo:1592-1621): near the end of a text the window reaches behind it.  The oracle's slice ends with the
    text, and so do the product's two forms -- the spans route notices (k_subs_reach) and hands such a batch to the
    lane-per-text kernel.  Batches with and without such a match, CSR neighbours whose first byte would show."""
    _need_gpu()
    rx = M.compile_regex(pat)
    orx = O.compile_regex(pat)
    assert orx.fixed_total_width >= 0
    rng = np.random.default_rng(zlib.crc32(pat))
    reaching = _random_texts(rng, 120, 40, b"abx0123456789 ") + [b"x", b"abx", b"ab", b"ab1", b"12x", b"1", b"", b"a1x", b"zz9",
                                                                  b"6502530000", b"650253000", b"call 6502530000", b"a", b"7", b"77", b"7a", b"12", b"1x", b"123", b"12x"]
    inside = [t + b" ." for t in reaching if t]   # every match has the whole window in front of the text's end
    for texts in (reaching, inside):
        for repl in (b"<\\1>", b"[\\2\\1]", b"#"):
            for count in (0, 1):
                got = rx.sub(repl, texts, count)
                for t, g in zip(texts, got):
                    try:
                        assert g == orx.sub(repl, t, count), (pat, repl, count, t)
                    except O.ReferenceDoesNotTerminate:
                        pass


def test_captures_fixed_width_groups():
    _need_gpu()
    pat = b"(\\d{3})(\\d{3})(\\d{4})"
    texts = [b"Call 6502530000 or 4155551234 today.", b"no digits", b"123456789", b"x12345678901234"]
    rx = M.compile_regex(pat)
    assert rx.num_groups == 3
    got = rx.captures(texts)
    c = O.compile_regex(pat)
    for i, t in enumerate(texts):
        w = c.captures_fixed(t)
        if w is None:
            assert (got[i] == -1).all()
        else:
            # a18 order: groups 1..g, then the whole match
            assert [tuple(int(x) for x in r) for r in got[i]] == [(s, e) for (_, s, e) in w]
    # general groups come from the backtracking matcher's flat program (round 2); the forms it does not
    # cover are refused
    got = M.compile_regex(b"(\\w+) (\\w+)").captures([b"hello world"])
    assert got[0].tolist() == [[0, 5], [6, 11], [0, 11]]
    got = M.compile_regex(b"(a|b)(c)").captures([b"ac", b"xbc"])   # alternation: _match_or on the flat program
    assert got[0].tolist() == [[0, 1], [1, 2], [0, 2]] and got[1].tolist() == [[1, 2], [2, 3], [1, 3]]
    with pytest.raises(M.UnsupportedPattern):
        M.compile_regex(b"(" * 17 + b"a" + b")" * 17 + b"(c)").captures([b"ac"])


STREAM_PATTERNS = [b"[a-z]+\\d+", b"\\d+", b"[a-z]+", b"\\w+\\s+", b"[^0-9]+", b"[a-z][0-9]", b"[a-z]{1,}",
                   b"[a-z]+[0-9]+x", b"ab|bc",
                   # class-table automata (more than 4 live states), literals, LazyDFA plans
                   b"hello", b"(\\d{3})(\\d{3})(\\d{4})", b"(x|y|foo|bar)+", b"(x|y|foo|bar)+z", b"(abc)+",
                   b"(a|b)x", b"a+b", b"[a-c]+[0-9]+[x-z]+[0-9]+", b"\\d{4}", b"[a-z]+[0-9]+[a-z]+[0-9]+[a-z]+",
                   b"(cat|dog)", b"(ab)+c", b"(foo|bar|baz)+"]


@pytest.mark.parametrize("pat", STREAM_PATTERNS)
@pytest.mark.parametrize("n,pitch,var", [(1, 16, False), (63, 48, True), (64, 64, False),
                                         (65, 80, True), (777, 208, True), (1000, 256, False),
                                         (130, 1024, True),
                                         # pitch not a multiple of 16: frame form of the kernel
                                         (200, 50, True), (70, 1001, False), (129, 7, True)])
def test_streaming_kernel_equals_generic_and_oracle(pat, n, pitch, var):
    _need_gpu()
    rng = np.random.default_rng(n * 131 + pitch)
    al = np.frombuffer(b"abcdxyz0123456789 -bcab" + bytes(c for c in pat if chr(c).isalnum()) * 2,
                       dtype=np.uint8)
    arr = rng.choice(al, size=(n, pitch)).astype(np.uint8)
    # long runs that span chunk boundaries
    for i in range(0, n, 5):
        k = int(rng.integers(0, pitch))
        arr[i, :k] = ord("q")
    lens = rng.integers(0, pitch + 1, size=n).astype(np.int32) if var else None
    rx = M.compile_regex(pat)
    assert "device.streamable=yes" in rx.describe()
    d = torch.from_numpy(arr).cuda().reshape(-1)
    dl = torch.from_numpy(lens).cuda() if var else None
    batch = M.DeviceBatch.strided(d, pitch, length=pitch, lens=dl)
    prefix, spans, total = rx._dev_findall(batch)
    assert M.load_library().mrx_last_kernel_name() in STREAM_FINDALL
    prefix, spans = prefix.cpu().numpy(), spans.cpu().numpy()
    texts = [arr[i, : (lens[i] if var else pitch)].tobytes() for i in range(n)]
    with generic_kernels():
        generic = rx.findall_lists(texts)
        assert M.load_library().mrx_last_kernel_name() in (b"k_step_count", b"k_findall_count")   # (stepper in its LZ form; start-accepting plans: generic)
    assert rx.findall_lists(texts) == generic   # CSR batch -> streaming kernel, ragged frame
    assert M.load_library().mrx_last_kernel_name() in STREAM_FINDALL
    assert int(prefix[-1]) == total == sum(len(x) for x in generic)
    # search on the same strided batch runs the streaming kernel in first-match mode
    ss, se = rx.match_next(batch)
    assert M.load_library().mrx_last_kernel_name() == b"k_stream_search"
    ss, se = ss.cpu().numpy(), se.cpu().numpy()
    fs, fe = rx.match_first(batch)
    fs, fe = fs.cpu().numpy(), fe.cpu().numpy()
    for i, t in enumerate(texts):
        have = [tuple(int(x) for x in r) for r in spans[prefix[i]:prefix[i + 1]]]
        assert have == generic[i], (pat, i)
        assert (int(ss[i]), int(se[i])) == (have[0] if have else (-1, -1)), (pat, i)
        if i % 7 == 0:
            assert have == O.findall(pat, t), (pat, i)
            w = O.match_first(pat, t)
            assert (int(fs[i]), int(fe[i])) == (w if w else (-1, -1)), (pat, i)


def test_streaming_slot_overflow_is_rewalked():
    _need_gpu()
    # 128 matches in one text: more than the 64-span slot
    t = (b"a1 " * 128)
    arr = np.frombuffer(t, dtype=np.uint8).reshape(1, -1).repeat(70, axis=0).copy()
    arr[3, :] = ord(" ")
    d = torch.from_numpy(arr).cuda().reshape(-1)
    rx = M.compile_regex(b"[a-z]+\\d+")
    prefix, spans, total = rx._dev_findall(M.DeviceBatch.strided(d, arr.shape[1], length=arr.shape[1]))
    prefix, spans = prefix.cpu().numpy(), spans.cpu().numpy()
    want = O.findall(b"[a-z]+\\d+", t)
    assert len(want) == 128 and total == 128 * 69
    assert [tuple(int(x) for x in r) for r in spans[prefix[0]:prefix[1]]] == want
    assert prefix[4] - prefix[3] == 0


@pytest.mark.parametrize("reps", [20, 50, 96, 340])
def test_streaming_dense_matches_all_decode_paths(reps):
    """Matches per wavefront below one decode tile, across several tiles, and above the
    direct-store threshold."""
    _need_gpu()
    rng = np.random.default_rng(reps)
    pitch = 1024
    arr = np.full((130, pitch), ord(" "), dtype=np.uint8)
    for i in range(130):
        k = int(rng.integers(0, reps + 1))
        body = b"".join(rng.choice([b"a1 ", b"xy22 ", b"q7"]) for _ in range(k))[:pitch]
        arr[i, :len(body)] = np.frombuffer(body, dtype=np.uint8)
    d = torch.from_numpy(arr).cuda().reshape(-1)
    pat = b"[a-z]+\\d+"
    rx = M.compile_regex(pat)
    prefix, spans, total = rx._dev_findall(M.DeviceBatch.strided(d, pitch, length=pitch))
    prefix, spans = prefix.cpu().numpy(), spans.cpu().numpy()
    for i in range(130):
        have = [tuple(int(x) for x in r) for r in spans[prefix[i]:prefix[i + 1]]]
        assert have == O.findall(pat, arr[i].tobytes()), i


def test_full_size_c2_properties():
    """BASELINE.json config 2 at full size: 2^20 texts x 1 KiB."""
    _need_gpu()
    n, L = 1 << 20, 1024
    d = make_c2_batch(n, L, seed=20260102, device="cuda")
    rx = M.compile_regex(b"[a-z]+\\d+")
    batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
    prefix, spans, total = rx._dev_findall(batch)
    assert total > n  # the token texts alone carry dozens of matches each
    sp = spans[:total].to(torch.int64)
    counts = prefix[1:] - prefix[:-1]
    owner = torch.repeat_interleave(torch.arange(n, device="cuda"), counts)
    # spans lie inside their text, are non-empty, ordered and non-overlapping per text
    assert bool((sp[:, 0] >= 0).all()) and bool((sp[:, 1] <= L).all())
    assert bool((sp[:, 1] > sp[:, 0]).all())
    same = owner[1:] == owner[:-1]
    assert bool((sp[1:, 0][same] >= sp[:-1, 1][same]).all())
    # every match is [a-z]+ followed by [0-9]+ and is maximal on both sides
    flat = d.reshape(-1).to(torch.int64)
    base = owner * L
    first = flat[base + sp[:, 0]]
    last = flat[base + sp[:, 1] - 1]
    assert bool(((first >= 97) & (first <= 122)).all())
    assert bool(((last >= 48) & (last <= 57)).all())
    has_prev = sp[:, 0] > 0
    prev = flat[(base + sp[:, 0] - 1).clamp(min=0)]
    assert not bool((has_prev & (prev >= 97) & (prev <= 122)).any())
    has_next = sp[:, 1] < L
    nxt = flat[(base + sp[:, 1]).clamp(max=flat.numel() - 1)]
    assert not bool((has_next & (nxt >= 48) & (nxt <= 57)).any())
    # the generic kernel (count only) agrees text by text
    with generic_kernels():
        counts2 = rx.count(batch)
        assert M.load_library().mrx_last_kernel_name() in (b"k_step_count", b"k_findall_count")   # (stepper in its LZ form; start-accepting plans: generic)
    assert bool((counts2.to(torch.int64) == counts).all())
    counts3 = rx.count(batch)   # count-only mode of the streaming kernel (no records, no decode)
    assert M.load_library().mrx_last_kernel_name() == b"k_stream_count"
    assert bool((counts3.to(torch.int64) == counts).all())
    # and the oracle agrees on a sample of every text kind
    host = d[:: n // 256].cpu().numpy()
    pre = prefix.cpu().numpy()
    sph = spans[:total].cpu().numpy()
    for j, i in enumerate(range(0, n, n // 256)):
        have = [tuple(int(x) for x in r) for r in sph[pre[i]:pre[i + 1]]]
        assert have == O.findall(b"[a-z]+\\d+", host[j].tobytes()), i


# ---- bitset NFA (the PikeVM / LazyDFA path without a table) -----
BITSET_CASES = [
    # (pattern, lazydfa_semantics, forced, alphabet)
    (b"(a|b)*a(a|b){12}", False, False, b"ab"),               # 2^13 DFA states: table refused, bitset runs
    (b"(a|b)*a(a|b){12}", False, False, b"abc"),
    (b"(a|b)*a[ab]{13}c", False, False, b"abbc"),
    (b"(a|b)*a(a|b){40}c", False, False, b"aabbc"),           # 85 positions: two words
    (b"(a|b)*a(a|b){70}", False, False, b"ab", 8),            # 144 positions: four words (slow in the oracle)
    (b"(a|b)x", False, True, b"abx "),                        # fits the table; forced onto the bitset walk
    (b"(ab)+c", False, True, b"abc"),
    (b"(foo|bar)+baz?", False, True, b"fobarz "),
    (b"x*y?z+(ab|cd)*", False, True, b"xyzabcd "),            # start set accepts? (no: z+ required)
    (b"(a|b)*", True, True, b"abc"),                          # start set accepts: empty matches
    (b"^(a|b)+c", False, True, b"abc"),                       # '^' passes at every start (pikevm.mojo:714)
    (b"(x|y|foo|bar)+", True, True, b"xyfobar "),             # config 5, LazyDFA reading
    (b"(\\d{3})(\\d{3})(\\d{4})", True, True, b"0123456789 -"),  # config 4 program on the NFA kernel
    (b"[a-z]+\\d+", True, True, b"abz019 -"),
]


@pytest.mark.parametrize("case", BITSET_CASES, ids=lambda c: c[0].decode() + "/" + c[3].decode())
def test_bitset_nfa_matches_oracle(case):
    _need_gpu()
    from mrx_ref.hybrid import CompiledRegex as OracleRegex
    pat, lazy, forced, al = case[:4]
    fewer = case[4] if len(case) > 4 else 1
    rng = np.random.default_rng(zlib.crc32(pat + al))
    texts = _random_texts(rng, 300 // fewer, 120, al) + _random_texts(rng, 40 // fewer, 600, al)
    rx = M.compile_regex(pat, lazydfa_semantics=lazy, bitset_nfa=forced)
    d = rx.describe()
    assert "device.bitset=yes" in d, d
    o = OracleRegex(pat, force_nfa=lazy)
    got_all = rx.findall_lists(texts)
    s, e = rx.match_next(texts)
    fs, fe = rx.match_first(texts)
    for i, t in enumerate(texts):
        assert got_all[i] == o.match_all(t), (pat, i, t)
        w = o.match_next(t, 0)
        assert (int(s[i]), int(e[i])) == (w if w else (-1, -1)), (pat, i, t)
        w = o.match_first(t, 0)
        w = w if (w and w[0] == 0) else None
        assert (int(fs[i]), int(fe[i])) == (w if w else (-1, -1)), (pat, i, t)
    if forced:
        # the table walk and the bitset walk are the same function of the text
        tab = M.compile_regex(pat, lazydfa_semantics=lazy)
        assert "device.bitset" not in tab.describe()
        assert tab.findall_lists(texts) == got_all


@pytest.mark.parametrize("pat,lazy,width", [
    (b"(\\d{3})(\\d{3})(\\d{4})", True, 10),      # config 4's program
    (b"(a|b)x", False, 2),
    (b"([a-f]|x)[0-9][0-9]", False, 3),
    (b"(a|b)(c|d)[0-9a-f][0-9a-f](x|y)", True, 5),
    (b"(ab)+c", False, 0),                        # a loop: several lengths
    (b"(foo|ba)z", False, 0),                     # branches of different lengths
])
@pytest.mark.parametrize("csr", [False, True])
def test_bitset_programs_of_one_match_length_need_no_second_pass(pat, lazy, width, csr):
    """When every match of a bitset program has the same length, the union pass (k_bscan) takes a match end iff the match
    does not begin inside the one taken before -- count, findall and search without any walk per start.  Equal to the
    walks (mrx_debug_multiwalk(2)) on every text and to the oracle on a sample; programs of several lengths keep the
    walks."""
    _need_gpu()
    from mrx_ref.hybrid import CompiledRegex as OracleRegex
    rx = M.compile_regex(pat, lazydfa_semantics=lazy, bitset_nfa=True)
    d = rx.describe()
    assert "device.bitset=yes" in d and ("fixed_len=%d\n" % width) in d, d
    rng = np.random.default_rng(zlib.crc32(pat) + csr)
    al = np.frombuffer(b"ab0123456789fxz -" + bytes(c for c in pat if chr(c).isalnum()) * 3, dtype=np.uint8)
    n, pitch = 700, 272
    arr = rng.choice(al, size=(n, pitch)).astype(np.uint8)
    arr[::3] = rng.choice(np.frombuffer(b"0123456789abfoz", dtype=np.uint8), size=arr[::3].shape)   # matches back to back
    lens = rng.integers(0, pitch + 1, size=n).astype(np.int32)
    lens[:8] = [0, 1, pitch, max(width, 1), max(width - 1, 0), 16, 17, 128]
    texts = [arr[i, :lens[i]].tobytes() for i in range(n)]
    if csr:
        batch = M.DeviceBatch.from_texts(texts)
    else:
        batch = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch, length=pitch, lens=torch.from_numpy(lens).cuda())
    lib = M.load_library()

    def run():
        prefix, spans, total = rx._dev_findall(batch)
        k1 = lib.mrx_last_kernel_name()
        cnt = rx.count(batch)
        k2 = lib.mrx_last_kernel_name()
        s, e = rx.match_next(batch)
        k3 = lib.mrx_last_kernel_name()
        return (prefix.cpu().numpy(), spans[:total].cpu().numpy(), cnt.cpu().numpy(), s.cpu().numpy(), e.cpu().numpy()), (k1, k2, k3)

    got, names = run()
    if width:
        assert names == (b"k_bscan_fixed", b"k_bscan_fixed", b"k_bscan_fixed_search"), names
    else:
        assert b"k_bscan_fixed" not in names and b"k_bscan_fixed_search" not in names, names
    with multiwalk(2):
        want, names2 = run()
        assert names2 == (b"k_bstep_count", b"k_bstep_count", b"k_bstep_search"), names2
    for a, b in zip(got, want):
        assert np.array_equal(a, b), pat
    assert np.array_equal(np.diff(got[0]), got[2])
    o = OracleRegex(pat, force_nfa=lazy)
    for i in range(0, n, 7):
        have = [tuple(int(x) for x in r) for r in got[1][got[0][i]:got[0][i + 1]]]
        assert have == o.match_all(texts[i]), (pat, i)
        w = o.match_next(texts[i], 0)
        assert (int(got[3][i]), int(got[4][i])) == (w if w else (-1, -1)), (pat, i)


@pytest.mark.parametrize("pat", [b"^[a-z]+$", b"(foo|bar)$", b"a+$|b", b"[0-9]+x?$", b"(ab|a)c*$", b"x*y$|z", b"(a|b)+c$",
                                 b"[a-c]+[0-9]*$", b"(\\d+|[a-f]+)$"])
def test_end_anchor_on_the_lazydfa_search_with_a_per_text_cache(pat):
    """'$' programs whose search / findall / sub the reference runs on the LazyDFA (matcher.mojo:401-431): a cached
    transition remembers whether '$' held when it was FIRST computed (pikevm.mojo:869-942), so upstream answers depend
    on earlier calls.  The product answers every text as a freshly compiled pattern would (walk_lazy_end, per-text
    masks of which (state, last byte) pairs were first computed where); the oracle restates the same contract with
    the real lazy cache, emptied at the start of every call."""
    _need_gpu()
    from mrx_ref import hybrid as H
    rx = M.compile_regex(pat)
    d = rx.describe()
    if "engine_type=NFA" not in d:
        pytest.skip("routed to the DFA engine")
    assert "device.lazy_end_cache=yes" in d and "support.search=yes" in d, d
    rng = np.random.default_rng(zlib.crc32(pat))
    al = b"abcfoobar0123xyz" + bytes(c for c in pat if chr(c).isalnum()) * 2
    texts = _random_texts(rng, 300, 24, al) + _random_texts(rng, 40, 300, al)
    # the same tail twice / a last byte that occurs nowhere else / only as the last byte of a repeated word
    texts += [b"abc", b"abcabc", b"abcabd", b"foo", b"foofoo", b"barfoo", b"xfoo bar", b"aaa", b"aab", b"ba", b"12x12x", b"12x123",
              b"ac", b"acac", b"abcc", b"", b"y", b"xxy", b"xyxy", b"z", b"abc1", b"abc1abc1", b"abc12", b"ff", b"12", b"1f1f"]
    o = H.CompiledRegex(pat)
    got = rx.findall_lists(texts)
    s, e = rx.match_next(texts)
    sub = rx.sub(b"<>", texts)
    cnt = rx.count(M.DeviceBatch.from_texts(texts)).cpu().numpy()
    nmatch = 0
    for i, t in enumerate(texts):
        assert got[i] == o.match_all(t), (pat, t)
        w = o.match_next(t, 0)
        assert (int(s[i]), int(e[i])) == (w if w else (-1, -1)), (pat, t)
        assert sub[i] == o.sub(b"<>", t), (pat, t)
        assert int(cnt[i]) == len(got[i])
        nmatch += len(got[i])
    assert nmatch > 0
    assert M.load_library().mrx_last_kernel_name() in (b"k_step_count", b"k_findall_count")   # (stepper in its LZ form; start-accepting plans: generic)


@pytest.mark.parametrize("pat", [b"[a-z]+\\d+", b"\\w+\\d{2}", b"(x|y|foo|bar)+", b"[A-Z]{10,20}[0-9]{15,25}|ab"])
def test_findall_with_offsets_known_on_the_host(pat):
    """mrx_findall_known_dev: the caller passes offsets[n] and the longest text's length (exact, or upper bounds) and
    the call needs no read-back before its scan -- same CSR as mrx_findall_dev."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat) + 11)
    al = b"abcxyz0123456789 fooQRSTUVWXYZ" + bytes(c for c in pat if chr(c).isalnum())
    texts = _random_texts(rng, 3000, 300, al) + _random_texts(rng, 8, 9000, al) + [b"", b"ab12"]
    rx = M.compile_regex(pat)
    plain = M.DeviceBatch.from_texts(texts)
    exact_end, exact_max = plain._end_offset, plain._max_len
    assert exact_end == sum(len(t) for t in texts) and exact_max == max(len(t) for t in texts)
    plain._end_offset = plain._max_len = None            # mrx_findall_dev: reads both from the device
    want = rx._dev_findall(plain)
    for end, mx in ((exact_end, exact_max), (exact_end + 12345, exact_max * 3 + 7)):
        b = M.DeviceBatch(plain.data, plain.offsets)
        b._end_offset, b._max_len = end, mx
        got = rx._dev_findall(b)
        assert got[2] == want[2] and torch.equal(got[0], want[0]) and torch.equal(got[1][:got[2]], want[1][:want[2]]), (pat, end, mx)
        pre = torch.empty_like(want[0]); sp = torch.empty_like(want[1])
        rx.findall_async(b, (pre, sp))
        torch.cuda.synchronize()
        assert torch.equal(pre, want[0]) and torch.equal(sp[:want[2]], want[1][:want[2]])


FIRST_PATTERNS = STREAM_PATTERNS + [b"[a-z]*[0-9]*", b"^abc", b"a*", b"\\w+@\\w+\\.com", b"[a-c]+x[0-9]+y",
                                    b"\\d{3}-\\d{4}", b"hello world this is long", b"^[a-z]+\\d*", b"x?y?z?",
                                    b"[a-z]+\\s+[a-z]+\\s+[0-9]+"]


@pytest.mark.parametrize("pat", FIRST_PATTERNS)
@pytest.mark.parametrize("n,pitch,var", [(1, 16, False), (65, 80, True), (777, 208, True), (130, 1024, False)])
def test_streaming_match_first_equals_generic_and_oracle(pat, n, pitch, var):
    """regex.match_first on a fixed-pitch batch runs the anchored automaton on the streaming
    kernel; it must agree with the generic lane-per-text kernel on every text and with the
    oracle on a sample."""
    _need_gpu()
    rng = np.random.default_rng(n * 977 + pitch + zlib.crc32(pat))
    al = np.frombuffer(b"abcdxyz0123456789 -bcab" + bytes(c for c in pat if chr(c).isalnum() or c in b" @.-") * 2,
                       dtype=np.uint8)
    arr = rng.choice(al, size=(n, pitch)).astype(np.uint8)
    for i in range(0, n, 3):   # texts that begin like a match: long letter runs, then digits
        k = int(rng.integers(0, pitch))
        arr[i, :k] = ord("q")
        arr[i, k:k + int(rng.integers(0, 40))] = ord("7")
    for i in range(1, n, 11):  # texts that begin with the pattern's own literal bytes
        lit = bytes(c for c in pat if chr(c).isalnum() or c in b" ")[:pitch]
        arr[i, :len(lit)] = np.frombuffer(lit, dtype=np.uint8)
    lens = rng.integers(0, pitch + 1, size=n).astype(np.int32) if var else None
    rx = M.compile_regex(pat)
    assert "device.first_stream=yes" in rx.describe()
    d = torch.from_numpy(arr).cuda().reshape(-1)
    dl = torch.from_numpy(lens).cuda() if var else None
    batch = M.DeviceBatch.strided(d, pitch, length=pitch, lens=dl)
    fs, fe = rx.match_first(batch)
    assert M.load_library().mrx_last_kernel_name() == b"k_stream_first"
    fs, fe = fs.cpu().numpy(), fe.cpu().numpy()
    texts = [arr[i, : (lens[i] if var else pitch)].tobytes() for i in range(n)]
    with generic_kernels():
        gs, ge = rx.match_first(texts)
        assert M.load_library().mrx_last_kernel_name() == b"k_match"
    assert np.array_equal(fs, gs) and np.array_equal(fe, ge), pat
    cs, ce = rx.match_first(texts)   # CSR batch -> streaming kernel, ragged frame
    assert M.load_library().mrx_last_kernel_name() == b"k_stream_first"
    assert np.array_equal(cs, gs) and np.array_equal(ce, ge), pat
    for i in range(0, n, 5):
        w = O.match_first(pat, texts[i])
        assert (int(fs[i]), int(fe[i])) == (w if w else (-1, -1)), (pat, i)


@pytest.mark.parametrize("pat", [b"[a-z]+\\d+", b"\\d+", b"hello", b"(\\d{3})(\\d{3})(\\d{4})", b"(x|y|foo|bar)+",
                                 b"(a|b)x", b"[a-z]+[0-9]+x"])
@pytest.mark.parametrize("shift", [0, 1, 7, 15])
def test_streaming_csr_ragged_unaligned(pat, shift):
    """CSR batches on the streaming kernel: arbitrary text lengths (0 .. 3000 bytes, several
    chunks), arbitrary alignment of every text and of the data pointer itself; findall, search
    and match_first against the generic kernels on every text and the oracle on a sample."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat) + shift)
    al = b"abcdxyz0123456789 -bcab" + bytes(c for c in pat if chr(c).isalnum()) * 2
    lens = np.concatenate([rng.integers(0, 40, size=150), rng.integers(0, 400, size=100),
                           rng.integers(1000, 3000, size=12), [0, 0, 1, 15, 16, 17, 127, 128, 129, 0]])
    rng.shuffle(lens)
    texts = [bytes(rng.choice(np.frombuffer(al, dtype=np.uint8), size=int(k)).tolist()) for k in lens]
    for i in range(0, len(texts), 9):      # long letter runs that cross chunk boundaries
        k = len(texts[i])
        texts[i] = (b"q" * (k // 2) + b"7" * (k // 6) + texts[i])[:k]
    data, offsets = M.api.pack_texts(texts)
    buf = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
    buf[shift:shift + data.size] = torch.from_numpy(data).cuda()
    batch = M.DeviceBatch(buf[shift:shift + data.size], torch.from_numpy(offsets).cuda())
    assert batch.data.data_ptr() % 16 == shift
    rx = M.compile_regex(pat)
    lib = M.load_library()
    prefix, spans, total = rx._dev_findall(batch)
    assert lib.mrx_last_kernel_name() in STREAM_FINDALL
    ss, se = rx.match_next(batch)
    assert lib.mrx_last_kernel_name() == b"k_stream_search"
    fs, fe = rx.match_first(batch)
    assert lib.mrx_last_kernel_name() == b"k_stream_first"
    with generic_kernels():
        gp, gsp, gtotal = rx._dev_findall(batch)
        assert lib.mrx_last_kernel_name() == b"k_findall_count"
        gss, gse = rx.match_next(batch)
        gfs, gfe = rx.match_first(batch)
    assert total == gtotal and torch.equal(prefix, gp) and torch.equal(spans[:total], gsp[:total])
    assert torch.equal(ss, gss) and torch.equal(se, gse)
    assert torch.equal(fs, gfs) and torch.equal(fe, gfe)
    pre, sp = prefix.cpu().numpy(), spans.cpu().numpy()
    for i in range(0, len(texts), 6):
        have = [tuple(int(x) for x in r) for r in sp[pre[i]:pre[i + 1]]]
        assert have == O.findall(pat, texts[i]), (pat, i)


ONEPASS_PATTERNS = [b"^[a-z]+[0-9]+$", b"^a|b$", b"^\\d+$", b"[a-z]+$", b"^\\w+$", b"^\\s$", b"^xa|yb$",
                    b"^\\+?1?[\\s.-]?\\(?([2-9]\\d{2})\\)?[\\s.-]?([2-9]\\d{2})[\\s.-]?(\\d{4})$"]


@pytest.mark.parametrize("pat", ONEPASS_PATTERNS)
def test_onepass_match_first_matches_oracle(pat):
    """NFA-routed '$' patterns that compile one-pass (matcher.mojo:378-379,
    onepass.mojo:440-488): match_first / is_match on the OnePass tables, CSR and fixed pitch."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat))
    rx = M.compile_regex(pat)
    assert "onepass=yes" in rx.describe()
    texts = _random_texts(rng, 150, 40, b"ab019 nz") + _random_texts(rng, 50, 300, b"abcxyz0123456789") + [
        b"", b"a", b"b", b"ab", b"na", b"nb", b"abc123", b"abc123x", b"12345", b" ", b"\t", b"hello_world1",
        b"6502530000", b"+1 (650) 253-0000", b"1-650-253-0000", b"650.253.0000", b"(650)253-0000",
        b"650 253 0000x", b"+1 (150) 253-0000", b"q" * 700, b"q" * 700 + b"1", b"7" * 513]
    fs, fe = rx.match_first(texts)
    assert M.load_library().mrx_last_kernel_name() == b"k_stream_first"
    flags = rx.is_match(texts)
    for i, t in enumerate(texts):
        w = O.match_first(pat, t)
        assert (int(fs[i]), int(fe[i])) == (w if w else (-1, -1)), (pat, t)
        assert bool(flags[i]) == (w is not None), (pat, t)
    # fixed pitch, aligned and not
    for pitch in (64, 50):
        arr = np.full((len(texts), pitch), ord("!"), dtype=np.uint8)
        lens = np.zeros(len(texts), np.int32)
        for i, t in enumerate(texts):
            k = min(len(t), pitch)
            arr[i, :k] = np.frombuffer(t[:k], dtype=np.uint8)
            lens[i] = k
        batch = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch,
                                      lens=torch.from_numpy(lens).cuda())
        s2, e2 = rx.match_first(batch)
        s2, e2 = s2.cpu().numpy(), e2.cpu().numpy()
        for i, t in enumerate(texts):
            w = O.match_first(pat, t[:pitch])
            assert (int(s2[i]), int(e2[i])) == (w if w else (-1, -1)), (pat, pitch, t)
    # search / findall of these patterns run on the LazyDFA upstream, with a transition cache that remembers where a
    # transition was first computed: served per text as a freshly compiled pattern would answer (round 3;
    # test_end_anchor_on_the_lazydfa_search_with_a_per_text_cache)
    ss, se = rx.match_next(texts)
    for i, t in enumerate(texts):
        w = O.search(pat, t)
        assert (int(ss[i]), int(se[i])) == (w if w else (-1, -1)), (pat, t)


CHAIN_SUBS = [(b"(\\w+) (\\w+)", b"\\2 \\1"), (b"(\\w+) (\\w+)", b"<\\2|\\1|\\2>"), (b"(\\w+) (\\w+)", b"\\3x\\1"),
              (b"([a-z]+)(\\d+)", b"\\2\\1"), (b"([a-z]+)-(\\d{2,4})", b"[\\2:\\1]"), (b"(\\d+)\\.(\\d+)", b"\\2,\\1"),
              (b"([a-c]{2,3})(x+)(\\d)", b"\\3\\2\\1"), (b"((\\d+)-([a-z]+))", b"\\3=\\2 (\\1)"), (b"(?:([a-z])(\\d+)) ", b"\\1"),
              (b"(\\d+)", b"<\\1>"), (b"([a-z]+)@([a-z]+)\\.(com|org)", b"\\2")]


@pytest.mark.parametrize("pat,repl", CHAIN_SUBS)
@pytest.mark.parametrize("count", [0, 2])
def test_sub_with_groups_of_a_chain_from_spans_equals_the_interpreter_and_oracle(pat, repl, count):
    """regex.sub with \\1..\\9 on a deterministic chain (describe(): chain_groups=yes): the matches are the plain search's
    spans, the groups the runs of the leaves' classes (k_subc_sizes / k_subc_emit), against the flat-program interpreter
    (k_sub: NFAEngine.match_next_with_groups, matcher.mojo:1781-1822) on every text and the oracle on a sample.  Patterns
    outside the form (an alternation, here) keep the interpreter; so does a batch with a text beyond the 4 KiB tile."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat + repl) + count)
    al = b"abcxyz0123456789 -.@" + bytes(c for c in pat if chr(c).isalnum()) * 2
    texts = _random_texts(rng, 300, 90, al) + _random_texts(rng, 60, 1500, al) + _random_texts(rng, 8, 4000, al) + [
        b"", b"a", b"hello world", b"hello world foo bar baz", b" hello  world ", b"ab12cd345", b"ab-123 cd-12345 e-1",
        b"3.14 2.718.1", b"abxx1 abcx9 abcdx1", b"12-ab 7-q", b"a1 b22 c333 ", b"joe@example.com ann@site.org x@y.net",
        b"word " * 800, b"w " * 2000, b"x" * 4096, b"ab 12 " * 600, b"q" * 3000 + b" zz yy"]
    rx = M.compile_regex(pat)
    lib = M.load_library()
    form = "chain_groups=yes" in rx.describe()
    with generic_kernels():
        want = rx.sub(repl, texts, count)
        assert lib.mrx_last_kernel_name() == b"k_sub_size"
    got = rx.sub(repl, texts, count)
    if form:   # (templates under which every match gains the same number of bytes skip the measuring pass: both forms)
        lib.mrx_debug_chain_sub_general(1)
        try:
            assert rx.sub(repl, texts, count) == got
        finally:
            lib.mrx_debug_chain_sub_general(0)
    fits = max(len(w) for w in want) <= 4096   # (an output beyond the tile also hands the call to the interpreter)
    assert lib.mrx_last_kernel_name() == (b"k_subc_emit" if form and fits else b"k_sub_size"), rx.describe()
    if form and not fits:
        short = [t for t, w in zip(texts, want) if len(w) <= 4096]
        assert rx.sub(repl, short, count) == [w for w in want if len(w) <= 4096]
        assert lib.mrx_last_kernel_name() == b"k_subc_emit"
    for i, (a, b) in enumerate(zip(got, want)):
        assert a == b, (pat, repl, count, i, texts[i][:80], len(texts[i]), a[:80], b[:80])
    for i in list(range(0, len(texts), 5)) + list(range(len(texts) - 17, len(texts))):
        assert got[i] == O.sub(pat, repl, texts[i], count), (pat, repl, texts[i][:80], count)
    # texts at every alignment of the input and of the output
    shifted = [texts[(i * 7) % 300][: 30 + i] for i in range(80)]
    with generic_kernels():
        want = rx.sub(repl, shifted, count)
    assert rx.sub(repl, shifted, count) == want
    # one text beyond the tile: the whole call is the interpreter's
    long_batch = texts[:20] + [b"ab 12 " * 900]
    with generic_kernels():
        want = rx.sub(repl, long_batch, count)
    assert rx.sub(repl, long_batch, count) == want
    assert lib.mrx_last_kernel_name() == b"k_sub_size"


def _random_chain_with_groups(rng):
    """A random chain of classes and literals with quantifiers, capture groups around runs of its elements (nested now
    and then), and a replacement template over the groups."""
    atoms = ["\\w", "\\d", "\\s", "[a-z]", "[a-c]", "[0-9a-f]", "[^ ]", " ", "-", "\\.", "@", "x", "a", ":"]
    quants = ["", "", "+", "+", "{2}", "{1,3}", "{2,}", "{3,5}"]
    n = int(rng.integers(1, 7))
    elems = [atoms[int(rng.integers(len(atoms)))] + quants[int(rng.integers(len(quants)))] for _ in range(n)]
    opens, closes = [0] * (n + 1), [0] * (n + 1)
    ngroups = int(rng.integers(1, 4))
    for _ in range(ngroups):
        a = int(rng.integers(0, n))
        b = int(rng.integers(a + 1, n + 1))
        opens[a] += 1
        closes[b] += 1
    # (groups opened at a and closed at b in any order nest or overlap: close the inner ones first by emitting every
    # close in front of the opens of the same position -- overlapping pairs are then simply another nesting)
    pat = ""
    depth = 0
    for i in range(n):
        c = min(closes[i], depth)
        pat += ")" * c
        depth -= c
        pat += "(" * opens[i]
        depth += opens[i]
        pat += elems[i]
    pat += ")" * depth
    refs = [b"\\1", b"\\2", b"\\3", b"\\4", b"<", b">", b"-", b"", b"::", b"x"]
    repl = b"".join(refs[int(rng.integers(len(refs)))] for _ in range(int(rng.integers(1, 6))))
    if b"\\" not in repl:
        repl += b"\\1"
    return pat.encode(), repl


def test_sub_with_groups_on_generated_chains_equals_the_oracle():
    """Generated chains with groups (the fuzz campaign's generators almost never produce one): regex.sub with a group
    template against the oracle on every text, whichever kernel the plan takes -- k_subc_emit where build_plan proves
    the chain's matches to be the table walk's (chain_groups=yes), the interpreter elsewhere -- and, for the proven ones,
    against the interpreter as well."""
    _need_gpu()
    rng = np.random.default_rng(int(os.environ.get("MRX_CHAIN_FUZZ_SEED", "20261005")))   # (other seeds: the round's soak runs)
    lib = M.load_library()
    al = b"abcxyz0123456789 -.@:af"
    texts = _random_texts(rng, 100, 60, al) + _random_texts(rng, 12, 700, al) + [
        b"", b"a", b" ", b"ab 12", b"hello world foo", b"aa-bb.cc@dd:ee", b"abc 123 abc 123 " * 40, b"x" * 300, b"a1 " * 500]
    proven = checked = 0
    # (the oracle's Python backtracker needs minutes on chains like \\w+\\w{2,}...; its C twin, equal to it on every reference
    # vector and on generated patterns -- tests/test_oracle_c.py --, takes over for this test, as in tests/big_fuzz.py)
    import mrx_ref.hybrid as _H
    monkey = _H.USE_C_BACKTRACK
    _H.USE_C_BACKTRACK = True
    try:
        proven, checked = _generated_chains_body(rng, lib, texts)
    finally:
        _H.USE_C_BACKTRACK = monkey
    assert proven >= 40 and checked > 8000, (proven, checked)


def _generated_chains_body(rng, lib, texts):
    proven = checked = 0
    for _ in range(int(os.environ.get("MRX_CHAIN_FUZZ_N", "400"))):
        pat, repl = _random_chain_with_groups(rng)
        try:
            rx = M.compile_regex(pat)
        except M.RegexSyntaxError:
            continue
        if " chain=1" not in rx.describe():   # (programs that backtrack are the interpreter's, with its own tests -- and
            continue                          # quadratic and worse on texts like these, upstream as here)
        for count in (0, 2):
            try:
                got = rx.sub(repl, texts, count)
            except M.UnsupportedPattern:
                break
            kernel = lib.mrx_last_kernel_name()
            form = "chain_groups=yes" in rx.describe() and "fixed_total=-1" in rx.describe()   # (fixed-width groups: k_subs_wave)
            fits = max(len(g) for g in got) <= 4096   # (an output beyond the tile hands the call to the interpreter)
            if form and not fits:
                assert kernel == b"k_sub_size", (pat, repl, kernel)
            elif form:
                assert kernel == b"k_subc_emit", (pat, repl, kernel)
                with generic_kernels():
                    assert rx.sub(repl, texts, count) == got, (pat, repl, count)
                lib.mrx_debug_chain_sub_general(1)
                try:
                    assert rx.sub(repl, texts, count) == got, (pat, repl, count)
                finally:
                    lib.mrx_debug_chain_sub_general(0)
            else:
                assert kernel != b"k_subc_emit", (pat, repl)
            for t, g in list(zip(texts, got))[:: 1 if form else 4]:   # (the interpreter has its own tests)
                try:
                    w = O.sub(pat, repl, t, count)
                except (O.UnsupportedByOracle, O.ReferenceDoesNotTerminate):
                    continue
                assert g == w, (pat, repl, count, t[:80], g[:80], w[:80], form)
                checked += 1
            proven += form and fits
    return proven, checked


@pytest.mark.parametrize("pat,repl", [(b"[a-z]+\\d+", b"#"), (b"[a-z]+\\d+", b""), (b"\\d", b""), (b"\\d+", b"<NUM>"),
                                      (b"(\\d{3})(\\d{3})(\\d{4})", b"\\1-\\2-\\3"),
                                      (b"(\\d{3})(\\d{3})(\\d{4})", b"(\\1) \\2-\\3 ext \\7"),
                                      (b"(x|y|foo|bar)+", b"_"), (b"hello", b"HELLO WORLD")])
@pytest.mark.parametrize("count", [0, 1, 3])
def test_sub_from_spans_equals_generic_and_oracle(pat, repl, count):
    """regex.sub assembled from streaming findall spans (k_subs_*) vs the generic lane-per-text
    sub kernel on every text and vs the oracle's _sub_impl on a sample.  Every form of the assembly:
    k_subs_wave with 16 / 32 / 64 / 256 lanes per text (texts beyond the group's LDS tiles -- frame over
    32 G + 16 bytes or output over 64 G -- fall to k_subs_emit inside the same call), k_subs_emit alone,
    and the form picked from the average text length."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat + repl) + count)
    al = b"abcxyz0123456789 -" + bytes(c for c in pat if chr(c).isalnum()) * 2
    texts = _random_texts(rng, 300, 90, al) + _random_texts(rng, 40, 1500, al) + _random_texts(rng, 6, 5000, al) + \
        _random_texts(rng, 3, 12000, al) + [b"ab12 " * 1638, b"0" * 8192, b"x1" * 4200] + [
        b"", b"6502530000", b"Call 6502530000 or 4155551234 today.", b"123", b"1", b"a1b2c3", b"hellohello",
        b"q" * 600 + b"1", b"9" * 333, b"1 2 3 4 " * 200, b"ab1 " * 127, b"ab1 " * 128, b"x" * 511 + b"1", b"y1" * 256,
        b"7" * 1024, b"hello" * 300, b"6502530000" * 51, b"a" * 1023 + b"1", b"foo bar " * 260]
    rx = M.compile_regex(pat)
    lib = M.load_library()
    with generic_kernels():
        want = rx.sub(repl, texts, count)
    for g in (-1, 16, 32, 64, 256, 0):
        with subs_group(g):
            got = rx.sub(repl, texts, count)
            assert lib.mrx_last_kernel_name() == (b"k_subs_emit" if g == 0 else b"k_subs_wave")
        for i, (a, b) in enumerate(zip(got, want)):
            assert a == b, (pat, repl, count, g, i, texts[i][:60], len(texts[i]))
    for i in range(0, len(texts), 7):
        assert want[i] == O.sub(pat, repl, texts[i], count), (pat, repl, texts[i], count)
    # texts at every alignment of the input (CSR: back to back) with one lane group per text
    shifted = [texts[i % len(texts)][: 40 + i] for i in range(64)]
    with generic_kernels():
        want = rx.sub(repl, shifted, count)
    for g in (16, 64, 256):
        with subs_group(g):
            assert rx.sub(repl, shifted, count) == want, (pat, repl, count, g)


@pytest.mark.parametrize("pat,repl", [(b"[a-z]+\\d+", b"#"), (b"\\d", b""), (b"\\d+", b"<NUM>"), (b"[0-9]+", b"#"),
                                      (b"(\\d{3})(\\d{3})(\\d{4})", b"\\1-\\2-\\3"), (b"(x|y|foo|bar)+", b"_"),
                                      (b"hello", b"HELLO WORLD"), (b"\\d{3}-\\d{3}-\\d{4}", b"XXX-XXX-XXXX"), (b"\\s+", b" ")])
@pytest.mark.parametrize("count", [0, 5])
def test_sub_emit_with_a_workgroup_per_text(pat, repl, count):
    """k_subs_emit<kBlock> (long texts: a workgroup per text, replacements staged window by window)
    against the 16-lanes-per-text form and the oracle: dense replacements (more per 4 KiB window than
    the window stages), empty replacements of touching matches, texts shorter than one round."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat + repl) + count + 1)
    al = b"abcxyz0123456789 -" + bytes(c for c in pat if chr(c).isalnum()) * 2
    texts = _random_texts(rng, 40, 90, al) + _random_texts(rng, 12, 9000, al) + _random_texts(rng, 3, 40000, al) + [
        b"", b"1", b"a1", b"1 2 3 4 5 6 7 8 9 " * 900, b"7" * 20000, b"a1" * 9000 + b"  " * 500,
        b"Call 6502530000 or 415-555-1234 today.  hello " * 400]
    rx = M.compile_regex(pat)
    lib = M.load_library()
    with long_text_kernels(1):
        got = rx.sub(repl, texts, count)
        assert lib.mrx_last_kernel_name() == b"k_subs_emit_long"
    with long_text_kernels(2), subs_group(0):
        want = rx.sub(repl, texts, count)
        assert lib.mrx_last_kernel_name() == b"k_subs_emit"
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, (pat, repl, count, i, len(texts[i]))
    for i in list(range(0, 40, 9)) + list(range(40, len(texts))):
        assert got[i] == O.sub(pat, repl, texts[i], count), (pat, repl, i, count)
    # by average length: a batch of long texts takes the workgroup form on its own
    long_only = [t for t in texts if len(t) >= 9000]
    got2 = rx.sub(repl, long_only, count)
    assert lib.mrx_last_kernel_name() == b"k_subs_emit_long"
    assert got2 == [g for g, t in zip(got, texts) if len(t) >= 9000]


@pytest.mark.parametrize("seed", [20260503, 20260504, 20260505, 20260506])
def test_generated_patterns_results_match_oracle(seed):
    """4 x 300 generated patterns (tests/pattern_gen.py) x 60 random texts: every operation the
    product accepts must return the oracle's result; refusals must coincide with the oracle's."""
    _need_gpu()
    from pattern_gen import patterns
    rng = np.random.default_rng(seed + 1)
    texts = (_random_texts(rng, 25, 48, b"abcxyz019 -@.") + _random_texts(rng, 15, 160, b"abcfoobarhellocatdog0123456789 xyz@.-")
             + _random_texts(rng, 10, 40, b"ab01") + [b"", b"a", b"foo", b"hello", b"abc123", b"foobar baz", b"cat dog",
                                                      b"http://id.no", b"aaa", b"xyz 999", b"q" * 200 + b"1"])
    checked = {"findall": 0, "search": 0, "match_first": 0, "refused": 0}
    for p in patterns(seed, 300):
        pb = p.encode()
        try:
            rx = M.compile_regex(pb)
        except M.RegexSyntaxError:
            with pytest.raises(Exception):
                O.compile_regex(pb)
            continue
        for op in ("findall", "search", "match_first"):
            try:
                if op == "findall":
                    got = rx.findall_lists(texts)
                elif op == "search":
                    s, e = rx.match_next(texts)
                    got = [(int(a), int(b)) if a >= 0 else None for a, b in zip(s, e)]
                else:
                    s, e = rx.match_first(texts)
                    got = [(int(a), int(b)) if a >= 0 else None for a, b in zip(s, e)]
            except M.UnsupportedPattern as exc:
                checked["refused"] += 1
                # The oracle restates the recursive backtracking matcher and answers (nearly) everything; the
                # product refuses, per pattern and with the reason, what the reference runs on that matcher
                # when the pattern is outside its flat-program form, LazyDFA '$' searches of more than 64 states,
                # SIMD-width-dependent nibble-table false positives and tables beyond its budgets.
                reason = str(exc)
                assert any(k in reason for k in ("flat-program form does not cover", "nibble-table",
                                                 "per-text transition cache is tracked for at most 64", "state budget", "DFA states",
                                                 "LDS staging budget")), (p, op, reason)
                continue
            for t, g in zip(texts, got):
                assert g == getattr(O, op)(pb, t), (p, op, t)
            checked[op] += 1
    assert checked["findall"] > 100 and checked["match_first"] > 150, checked
    assert checked["refused"] < 0.25 * (checked["findall"] + checked["search"] + checked["match_first"]), checked


@pytest.mark.parametrize("seed", [20260601, 20260602, 20260603])
def test_generated_patterns_streaming_equals_generic(seed):
    """The streamability proof under fire: for every generated pattern that the plan marks
    streamable, the single-pass kernel (findall, search, match_first, count) must equal the
    generic restart-per-position kernels on texts built from the pattern's own alphabet."""
    _need_gpu()
    from pattern_gen import patterns
    lib = M.load_library()
    rng = np.random.default_rng(seed)
    n, pitch = 192, 272
    base = np.frombuffer(b"abcxyz019 -@.fobrhelcatdg", dtype=np.uint8)
    nstream = nfirst = 0
    for p in patterns(seed, 400):
        pb = p.encode()
        try:
            rx = M.compile_regex(pb)
        except M.RegexSyntaxError:
            continue
        d = rx.describe()
        streamable = "device.streamable=yes" in d
        first_stream = "device.first_stream=yes" in d
        if not (streamable or first_stream):
            continue
        lit = np.frombuffer(bytes(c for c in pb if chr(c).isalnum() or c in b" -@."), dtype=np.uint8)
        al = np.concatenate([base, lit, lit]) if lit.size else base
        arr = rng.choice(al, size=(n, pitch)).astype(np.uint8)
        for i in range(0, n, 4):     # long single-byte runs: worst case for restart-per-position
            arr[i, : int(rng.integers(0, pitch))] = al[int(rng.integers(0, al.size))]
        lens = rng.integers(0, pitch + 1, size=n).astype(np.int32)
        batch = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch,
                                      lens=torch.from_numpy(lens).cuda())
        if streamable:
            nstream += 1
            pre, sp, tot = rx._dev_findall(batch)
            assert lib.mrx_last_kernel_name() in STREAM_FINDALL
            ss, se = rx.match_next(batch)
            cnt = rx.count(batch)
        if first_stream:
            nfirst += 1
            fs, fe = rx.match_first(batch)
            assert lib.mrx_last_kernel_name() == b"k_stream_first"
        with generic_kernels():
            if streamable:
                gpre, gsp, gtot = rx._dev_findall(batch)
                gss, gse = rx.match_next(batch)
            if first_stream and "onepass=yes" not in d:
                gfs, gfe = rx.match_first(batch)
        if streamable:
            assert tot == gtot and torch.equal(pre, gpre) and torch.equal(sp[:tot], gsp[:tot]), p
            assert torch.equal(ss, gss) and torch.equal(se, gse), p
            assert torch.equal(cnt.to(torch.int64), pre[1:] - pre[:-1]), p
        if first_stream and "onepass=yes" not in d:
            assert torch.equal(fs, gfs) and torch.equal(fe, gfe), p
    assert nstream > 40 and nfirst > 150, (nstream, nfirst)


EXACT_LITERALS = [b"hello world this is long", b"abcabcabcabcabcabcabcabc", b"aaaaaaaaaaaaaaaaaaaaaa",
                  b"aaaaaaaaaaaaaaaaaaaab", b"xyxyxyxyxyxyxyxyxyxyxyxyxy"]


@pytest.mark.parametrize("pat", [b"555-123-4567", b"555\\.123\\.4567", b"aab", b"abab", b"aaa", b"abcab", b"xx"])
def test_self_overlapping_pure_literal_streams(pat):
    """DFAEngine's pure-literal loops (dfa.mojo:2053-2073) for a literal whose prefix is also a suffix:
    the KMP automaton with the full state falling back to the start (matches never overlap); findall,
    search and count against the generic kernels on every text and the oracle on a sample."""
    _need_gpu()
    rx = M.compile_regex(pat)
    d = rx.describe()
    assert rx.get_engine_type() == "DFA" and "device.streamable=yes" in d
    lib = M.load_library()
    lit = pat.replace(b"\\", b"")
    rng = np.random.default_rng(zlib.crc32(pat) + 3)
    al = bytes(set(lit)) + b" z"
    texts = []
    for _ in range(200):
        parts = []
        for _ in range(int(rng.integers(0, 7))):
            k = rng.random()
            if k < 0.4:
                parts.append(lit)
            elif k < 0.7:   # runs of the literal's own prefix: candidates that overlap
                parts.append(lit[: max(1, len(lit) // 3)] * int(rng.integers(1, 30)))
            else:
                parts.append(bytes(rng.choice(np.frombuffer(al, dtype=np.uint8), size=int(rng.integers(0, 40))).tolist()))
        texts.append(b"".join(parts))
    texts += [b"", lit, lit[:-1], lit + lit, lit[1:] + lit, lit[:1] * 100, (lit[:2] * 50) + lit]
    got = rx.findall_lists(texts)
    assert lib.mrx_last_kernel_name() in STREAM_FINDALL
    s, e = rx.match_next(texts)
    assert lib.mrx_last_kernel_name() == b"k_stream_search"
    with generic_kernels():
        want = rx.findall_lists(texts)
        gs, ge = rx.match_next(texts)
    assert got == want and s.tolist() == gs.tolist() and e.tolist() == ge.tolist()
    assert sum(len(g) for g in got) > 100
    for i in range(0, len(texts), 3):
        assert got[i] == O.findall(pat, texts[i]), (pat, texts[i])
        m = O.search(pat, texts[i])
        assert (None if s[i] < 0 else (int(s[i]), int(e[i]))) == m
        assert rx.sub(b"<>", [texts[i]], 0)[0] == O.sub(pat, b"<>", texts[i], 0)


@pytest.mark.parametrize("pat", [b"[a-z]+\\d+", b"\\d+", b"hello", b"[0-9]{4}", b"[A-Z]{3}", b"a{2,4}", b"555-123-4567",
                                 b"[0-9]{10}", b"(x|y|foo|bar)+", b"a{1}b{2}c{3}d{4}", b"[a-zA-Z0-9]+"])
def test_long_texts_in_pieces(pat):
    """Long texts cut at synchronising bytes (k_stream_findall VIRT): findall equals the uncut walk and
    the oracle -- matches across cuts, texts without any synchronising byte (they stay whole), empty
    and short texts between long ones, CSR and fixed-pitch batches."""
    _need_gpu()
    rx = M.compile_regex(pat)
    d = rx.describe()
    if "device.streamable=yes" not in d or "sync_bytes=0" in d:
        pytest.skip("no streaming automaton with synchronising bytes")
    lib = M.load_library()
    rng = np.random.default_rng(zlib.crc32(pat) + 5)
    al = b"abxyfor0123456789HELO -.," + bytes(c for c in pat if chr(c).isalnum()) * 2
    runs = [bytes([c]) * 900 for c in b"a5xA"]          # long runs: cuts with no synchronising byte nearby
    texts = (_random_texts(rng, 20, 150, al) + _random_texts(rng, 30, 3000, al) + _random_texts(rng, 6, 20000, al) +
             [b"", b"a1", runs[0] + b"12 " + runs[1], runs[2] + b" " + runs[3] + b"7",
              b"ab 12 " + runs[0] * 40 + b"77 x1 " + runs[1] * 30 + b" a5 b6",   # > 64 cuts in a row that cannot be made
              (b"hello abc123 555-123-4567 2024 ABCD xyfoobar aaaa abbcccdddd " * 120),
              bytes(rng.choice(np.frombuffer(b"abc123 ", dtype=np.uint8), size=5000).tolist()) + runs[1] * 3])
    with long_text_kernels(1):
        got = rx.findall_lists(texts)
        assert lib.mrx_last_kernel_name() == b"k_stream_findall_pieces"
    with long_text_kernels(2):
        want = rx.findall_lists(texts)
        assert lib.mrx_last_kernel_name() in STREAM_FINDALL
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, (pat, i, len(texts[i]), g[:5], w[:5])
    for i in list(range(0, 50, 7)) + list(range(50, len(texts))):
        assert got[i] == O.findall(pat, texts[i]), (pat, i)
    assert sum(len(g) for g in got) > 50
    data, offsets = M.api.pack_texts(texts)
    b = M.DeviceBatch(torch.from_numpy(data).cuda(), torch.from_numpy(offsets).cuda())
    with long_text_kernels(1):
        cnt = rx.count(b).cpu().numpy()
        assert lib.mrx_last_kernel_name() == b"k_stream_count_pieces"
        sub = rx.sub(b"<#>", texts, 0)
        ss, se = rx.match_next(texts)
        searched_in_pieces = lib.mrx_last_kernel_name() == b"k_stream_search_pieces"
    assert cnt.tolist() == [len(g) for g in got]
    assert searched_in_pieces or "findall_only=1" in d
    # search = the first findall match on these plans (no empty matches, same route)
    if searched_in_pieces:
        with long_text_kernels(2):
            ws, we = rx.match_next(texts)
        assert ss.tolist() == ws.tolist() and se.tolist() == we.tolist()
        for i in range(50, len(texts)):
            m = O.search(pat, texts[i])
            assert (None if ss[i] < 0 else (int(ss[i]), int(se[i]))) == m, (pat, i)
    for i in range(50, len(texts)):
        assert sub[i] == O.sub(pat, b"<#>", texts[i], 0), (pat, i)
    # fixed pitch with per-text lengths (unaligned pitch: frame form)
    for pitch in (2048, 3001):
        n = 41
        arr = rng.choice(np.frombuffer(al, dtype=np.uint8), size=(n, pitch)).astype(np.uint8)
        lens = rng.integers(0, pitch + 1, size=n).astype(np.int32)
        sb = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch, length=pitch,
                                   lens=torch.from_numpy(lens).cuda())
        with long_text_kernels(1):
            pre, sp, tot = rx._dev_findall(sb)
            assert lib.mrx_last_kernel_name() == b"k_stream_findall_pieces"
        with long_text_kernels(2):
            pre2, sp2, tot2 = rx._dev_findall(sb)
        assert tot == tot2 and torch.equal(pre, pre2) and torch.equal(sp[:tot], sp2[:tot2])


def test_outliers_of_a_ragged_batch_are_cut_into_pieces():
    """150 000 short texts and three of 100-300 KB in one CSR batch: without pieces the three long ones
    would each be one lane's work; the batch is cut although it has plenty of texts (per-text piece
    counts: the short texts are one piece each)."""
    _need_gpu()
    pat = b"[a-z]+\\d+"
    rx = M.compile_regex(pat)
    lib = M.load_library()
    rng = np.random.default_rng(77)
    al = np.frombuffer(b"abcxyz0123456789 -", dtype=np.uint8)
    lens = rng.integers(0, 120, size=150000)
    for pos, ln in ((5, 300000), (70001, 100000), (149999, 200000)):
        lens[pos] = ln
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    data = rng.choice(al, size=int(offsets[-1])).astype(np.uint8)
    b = M.DeviceBatch(torch.from_numpy(data).cuda(), torch.from_numpy(offsets).cuda())
    pre, sp, tot = rx._dev_findall(b)
    assert lib.mrx_last_kernel_name() == b"k_stream_findall_pieces"
    cnt = rx.count(b)      # (count / search of a batch this large do not look for outliers: no host sync)
    ss, se = rx.match_next(b)
    with long_text_kernels(2):
        pre2, sp2, tot2 = rx._dev_findall(b)
        assert lib.mrx_last_kernel_name() in STREAM_FINDALL
        cnt2 = rx.count(b)
        ws, we = rx.match_next(b)
    assert tot == tot2 and torch.equal(pre, pre2) and torch.equal(sp[:tot], sp2[:tot2])
    assert torch.equal(cnt, cnt2) and torch.equal(ss, ws) and torch.equal(se, we)
    t = data[offsets[5]:offsets[6]].tobytes()
    a, z = int(pre[5].item()), int(pre[6].item())
    assert [tuple(x) for x in sp[a:z].cpu().numpy().tolist()] == O.findall(pat, t)


@pytest.mark.parametrize("pat", [b"\\d{3}-\\d{3}-\\d{4}", b"\\w+\\d{2}", b"(foo|foobar)x", b"[a-z]+@[a-z]+"])
def test_outliers_of_a_ragged_batch_on_the_stepper(pat):
    """The stepper's routes on a ragged batch with a few very long texts: the lane-per-text kernel takes
    the short ones, the wavefront-per-text kernel the long ones (Layout::split), in the same call."""
    _need_gpu()
    rx = M.compile_regex(pat)
    if "step_search=1" not in rx.describe():
        pytest.skip("not a stepper plan")
    lib = M.load_library()
    rng = np.random.default_rng(zlib.crc32(pat) + 9)
    al = np.frombuffer(b"abfox0123456789@.- " + bytes(c for c in pat if chr(c).isalnum()) * 2, dtype=np.uint8)
    lens = rng.integers(0, 100, size=140000)
    for pos, ln in ((0, 90000), (77777, 40000), (139999, 150000)):
        lens[pos] = ln
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    data = rng.choice(al, size=int(offsets[-1])).astype(np.uint8)
    for o in (offsets[0] + 500, offsets[77777] + 20000):      # something to find in the long texts
        data[o:o + 30] = np.frombuffer(b" 555-123-4567 foobarx ab12 a@b", dtype=np.uint8)
    b = M.DeviceBatch(torch.from_numpy(data).cuda(), torch.from_numpy(offsets).cuda())
    with no_streaming_kernels():
        pre, sp, tot = rx._dev_findall(b)
        assert lib.mrx_last_kernel_name() == b"k_step_count+k_req_wave"
        cnt = rx.count(b)
        assert lib.mrx_last_kernel_name() == b"k_step_count+k_req_wave"
        ss, se = rx.match_next(b)
        assert lib.mrx_last_kernel_name() == b"k_step_search+k_req_wave_search"
    with no_streaming_kernels(), long_text_kernels(2):
        pre2, sp2, tot2 = rx._dev_findall(b)
        assert lib.mrx_last_kernel_name() == b"k_step_count"
        cnt2 = rx.count(b)
        ws, we = rx.match_next(b)
    assert tot == tot2 and torch.equal(pre, pre2) and torch.equal(sp[:tot], sp2[:tot2])
    assert torch.equal(cnt, cnt2) and torch.equal(ss, ws) and torch.equal(se, we)
    assert int(cnt[0].item()) >= 1 and int(cnt[77777].item()) >= 1


@pytest.mark.parametrize("pat", [b"[a-z]+", b"\\w+", b"\\d+", b"a+", b"[a-zA-Z0-9]+"])
@pytest.mark.parametrize("pitch", [4096, 3001, 64])
def test_match_first_of_a_class_run_with_a_wavefront_per_text(pat, pitch):
    """k_first_run (match_first = the run of class bytes at 0, one wavefront per text) against the anchored
    walk of the streaming kernel and the oracle: runs that end in the first block, in a later one, at the
    end of the text, empty runs, empty texts, unaligned pitch."""
    _need_gpu()
    rx = M.compile_regex(pat)
    assert "class_run=1" in rx.describe()
    lib = M.load_library()
    rng = np.random.default_rng(zlib.crc32(pat) + pitch)
    n = 70
    arr = rng.choice(np.frombuffer(b"abcxyzABC0123456789_ -", dtype=np.uint8), size=(n, pitch)).astype(np.uint8)
    fill = {b"\\d+": b"7", b"a+": b"a"}.get(pat, b"q")
    lens = rng.integers(0, pitch + 1, size=n).astype(np.int32)
    for i in range(0, n, 3):       # long runs: up to the whole text
        k = int(rng.integers(0, pitch + 1))
        arr[i, :k] = fill[0]
    lens[0], lens[1] = pitch, 0
    arr[0, :] = fill[0]
    sb = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch, length=pitch,
                               lens=torch.from_numpy(lens).cuda())
    with long_text_kernels(1):
        s1, e1 = rx.match_first(sb)
        assert lib.mrx_last_kernel_name() == b"k_first_run"
    with long_text_kernels(2):
        s2, e2 = rx.match_first(sb)
        assert lib.mrx_last_kernel_name() == b"k_stream_first"
    assert torch.equal(s1, s2) and torch.equal(e1, e2)
    s1, e1 = s1.cpu().numpy(), e1.cpu().numpy()
    for i in range(n):
        w = O.match_first(pat, arr[i, :lens[i]].tobytes())
        assert (int(s1[i]), int(e1[i])) == (w if w else (-1, -1)), (pat, i)
    assert int(e1[0]) == pitch and int(e1[1]) == -1


@pytest.mark.parametrize("pat", [b"[a-z]+", b"\\d+", b"hello", b"[a-z]+\\d+", b"^\\d+$", b"^[a-z]+[0-9]+$"])
def test_is_match_on_fixed_pitch_batches(pat):
    """mrx_is_match_strided_dev equals the CSR entry point and the oracle (is_match quirk included)."""
    _need_gpu()
    rx = M.compile_regex(pat)
    rng = np.random.default_rng(zlib.crc32(pat) + 2)
    n, pitch = 90, 41
    arr = rng.choice(np.frombuffer(b"abhelo0123456789+-. ", dtype=np.uint8), size=(n, pitch)).astype(np.uint8)
    lens = rng.integers(0, pitch + 1, size=n).astype(np.int32)
    texts = [arr[i, :lens[i]].tobytes() for i in range(n)]
    sb = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch, length=pitch, lens=torch.from_numpy(lens).cuda())
    got = rx.is_match(sb).cpu().numpy().astype(bool).tolist()
    assert got == [bool(x) for x in rx.is_match(texts)]
    orx = O.compile_regex(pat)
    assert got == [bool(orx.is_match(t)) for t in texts]


@pytest.mark.parametrize("pat", EXACT_LITERALS)
def test_exact_literal_kmp_streaming(pat):
    """HybridMatcher's exact-literal bypass (matcher.mojo:768-781, 815-847) on the streaming kernel:
    KMP automaton of the literal, findall returns EVERY occurrence (overlapping ones too), search
    the first; against the generic kernels on every text and the oracle on a sample."""
    _need_gpu()
    rx = M.compile_regex(pat)
    d = rx.describe()
    assert "ExactLiteral" in rx.get_engine_type() and "device.streamable=yes" in d
    lib = M.load_library()
    rng = np.random.default_rng(zlib.crc32(pat))
    al = bytes(set(pat)) + b" z"
    texts = []
    for _ in range(200):
        parts = []
        for _ in range(int(rng.integers(0, 6))):
            k = rng.random()
            if k < 0.45:
                parts.append(pat)
            elif k < 0.7:   # a run of the literal's own period: overlapping occurrences
                parts.append(pat[: max(1, len(pat) // 8)] * int(rng.integers(1, 40)))
            else:
                parts.append(bytes(rng.choice(np.frombuffer(al, dtype=np.uint8), size=int(rng.integers(0, 50))).tolist()))
        texts.append(b"".join(parts))
    texts += [b"", pat, pat[:-1], pat + pat, pat[1:] + pat]
    got = rx.findall_lists(texts)
    assert lib.mrx_last_kernel_name() in STREAM_FINDALL
    s, e = rx.match_next(texts)
    assert lib.mrx_last_kernel_name() == b"k_stream_search"
    cnt_total = sum(len(g) for g in got)
    with generic_kernels():
        want = rx.findall_lists(texts)
        gs, ge = rx.match_next(texts)
    assert got == want and cnt_total > 100
    assert np.array_equal(s, gs) and np.array_equal(e, ge)
    for i in range(0, len(texts), 5):
        assert got[i] == O.findall(pat, texts[i]), (pat, texts[i])
        w = O.search(pat, texts[i])
        assert (int(s[i]), int(e[i])) == (w if w else (-1, -1))
    # sub keeps its own non-overlapping search loop (generic kernel)
    out = rx.sub(b"#", texts[:40])
    for i in range(40):
        assert out[i] == O.sub(pat, b"#", texts[i]), (pat, texts[i])


@pytest.mark.parametrize("seed", [20260701, 20260702, 20260703])
def test_generated_patterns_stepper_equals_literal_restatement(seed):
    """k_step_* (one flattened loop) against the literal restatement of the reference's nested loops
    on every generated PF_STEPPABLE pattern: findall, count and search, ragged texts built from the
    pattern's own alphabet; plus the oracle on a sample."""
    _need_gpu()
    from pattern_gen import patterns
    lib = M.load_library()
    rng = np.random.default_rng(seed)
    base = b"abcxyz019 -@.fobrhelcatdg"
    nstep = nreq = 0
    for p in patterns(seed, 400):
        pb = p.encode()
        try:
            rx = M.compile_regex(pb)
        except M.RegexSyntaxError:
            continue
        dsc = rx.describe()
        if "device.steppable=yes" not in dsc and "device.steppable=required-byte route" not in dsc:
            continue
        nstep += 1
        nreq += "required-byte route" in dsc
        al = base + bytes(c for c in pb if chr(c).isalnum() or c in b" -@.") * 2
        texts = _random_texts(rng, 90, 70, al) + _random_texts(rng, 12, 400, al)
        for j in range(0, len(texts), 6):
            k = len(texts[j])
            texts[j] = (bytes([al[int(rng.integers(0, len(al)))]]) * (k // 2) + texts[j])[:k]
        try:
            big = "big_table=1" in dsc   # > 96 states: only the wavefront kernel has the table form
            with no_streaming_kernels(), long_text_kernels(1 if big else 0):
                got = rx.findall_lists(texts)
                assert lib.mrx_last_kernel_name() == (b"k_req_wave" if big else b"k_step_count")
                gs, ge = rx.match_next(texts)
        except M.UnsupportedPattern:   # tables beyond the LDS staging budget
            continue
        with generic_kernels():
            want = rx.findall_lists(texts)
            assert lib.mrx_last_kernel_name() == b"k_findall_count"
            ws, we = rx.match_next(texts)
        assert got == want, p
        assert np.array_equal(gs, ws) and np.array_equal(ge, we), p
        for j in range(0, len(texts), 17):
            assert got[j] == O.findall(pb, texts[j]), (p, texts[j])
    assert nstep > 150, (nstep, nreq)


@pytest.mark.parametrize("seed", [41, 42, 43])
def test_generated_patterns_multiwalk_equals_stepper_and_oracle(seed):
    """k_mwalk (several walks side by side, one pass) against the windowed stepper's restart-per-position loop on
    every generated pattern that has the multi-walk table: findall (slot rows and the second walk), count and
    search, ragged CSR texts that overlap the pattern's own pieces; the oracle on a sample."""
    _need_gpu()
    from pattern_gen import patterns, patterns2
    lib = M.load_library()
    rng = np.random.default_rng(seed)
    base = b"abcxyz019 -@.fobrhelcatdg"
    nmw = nreq = nback = 0
    # (required-byte plans: findall / count take the route's own table, search the plain one)
    extra = ["[a-z]+@[a-z]+\\.com", "\\d{3}-\\d{4}", "[0-9]+\\.[0-9]+", "[a-z0-9._%+-]+@[a-z0-9.-]+\\.[a-z]{2,}", "\\d{3}-\\d{3}-\\d{4}",
             "[a-c]+:[0-9]+", "\\w+@\\w+", "[89]00\\d{6}"]
    for p in patterns(seed, 260) + patterns2(seed, 140) + extra:
        pb = p.encode()
        try:
            rx = M.compile_regex(pb)
        except (M.RegexSyntaxError, M.UnsupportedPattern):
            continue
        dsc = rx.describe()
        if "multiwalk=yes" not in dsc and "multiwalk_req=yes" not in dsc and "backset=yes" not in dsc:
            continue
        nmw += 1
        nreq += "multiwalk_req=yes" in dsc
        nback += "backset=yes" in dsc and "multiwalk=yes" not in dsc
        al = base + bytes(c for c in pb if chr(c).isalnum() or c in b" -@.") * 2
        texts = _random_texts(rng, 90, 70, al) + _random_texts(rng, 12, 700, al) + [b"", pb[:1], bytes(al[:3]) * 40]
        for j in range(0, len(texts), 5):   # long runs of one byte: walks that overlap themselves
            k = len(texts[j])
            texts[j] = (bytes([al[int(rng.integers(0, len(al)))]]) * (k // 2) + texts[j])[:k]
        with long_text_kernels(2):
            got = rx.findall_lists(texts)
            if "tries_walk=yes" in dsc and "multiwalk_req=yes" not in dsc and "required-byte route" not in dsc:
                # (round 4: walks that stay within seven bytes of their match -- one pass with the pending tries)
                assert lib.mrx_last_kernel_name() == b"k_mwalk", (p, lib.mrx_last_kernel_name())
            elif "multiwalk=yes" not in dsc and "multiwalk_req=yes" not in dsc:   # marks of the match starts + the stepper
                # (tables of more than 96 states: the marked stepper in its class-indexed form)
                assert lib.mrx_last_kernel_name() in (b"k_backscan+k_step_count", b"k_req_wave"), (p, lib.mrx_last_kernel_name())
            elif "multiwalk_req=yes" in dsc or "required-byte route" not in dsc:
                assert lib.mrx_last_kernel_name() == b"k_mwalk", (p, lib.mrx_last_kernel_name())
            gs, ge = rx.match_next(texts)
            gc = rx.count(M.DeviceBatch.from_texts(texts)).cpu().numpy()
            with multiwalk(2):
                want = rx.findall_lists(texts)
                assert lib.mrx_last_kernel_name() not in (b"k_mwalk", b"k_backscan+k_step_count")
                ws, we = rx.match_next(texts)
        texts_h = texts   # (texts with the required byte doubled / hits inside matches / hits without a run in front)
        if "multiwalk_req=yes" in dsc:
            rb = int(re.search(r"required_byte=(-?\d+)", dsc).group(1))
            more = [bytes([rb]) * 3 + t[:40] + bytes([rb]) + t[40:80] + bytes([rb, rb]) + t[80:] for t in texts[:40]]
            with long_text_kernels(2):
                g2 = rx.findall_lists(more)
                with multiwalk(2):
                    w2 = rx.findall_lists(more)
            assert g2 == w2, (p, [(t, a, b) for t, a, b in zip(more, g2, w2) if a != b][:2])
            for j in range(0, len(more), 7):
                assert g2[j] == O.findall(pb, more[j]), (p, more[j])
        assert got == want, (p, [(t, g, w) for t, g, w in zip(texts, got, want) if g != w][:3])
        assert np.array_equal(gs, ws) and np.array_equal(ge, we), p
        assert [int(x) for x in gc] == [len(g) for g in got], p
        for j in range(0, len(texts), 13):
            assert got[j] == O.findall(pb, texts[j]), (p, texts[j])
        # whole wavefronts of full rows (fixed pitch, no lens): the kernels' paths without the per-byte frame test, the
        # packed-start form (mrx_debug_multiwalk(3): without it), rows without a match beside rows full of them
        blob = b"".join(texts) + bytes(al) * 8
        rows = np.frombuffer((blob * (128 * 256 // len(blob) + 1))[:128 * 256], dtype=np.uint8).reshape(128, 256).copy()
        rows[5::7] = ord("~")
        fb = M.DeviceBatch.strided(torch.from_numpy(rows).cuda().reshape(-1), 256, length=256)
        res = []
        for mode in (0, 3, 2):
            with long_text_kernels(2), multiwalk(mode):
                pre_, sp_, tot_ = rx._dev_findall(fb)
                s_, e_ = rx.match_next(fb)
                c_ = rx.count(fb)
            res.append((pre_.cpu().numpy(), sp_[:tot_].cpu().numpy(), s_.cpu().numpy(), e_.cpu().numpy(), c_.cpu().numpy()))
        for other in res[1:]:
            for a_, b_ in zip(res[0], other):
                assert np.array_equal(a_, b_), p
        assert np.array_equal(np.diff(res[0][0]), res[0][4]), p
    assert nmw > 60 and nreq >= 5 and nback >= 15, (nmw, nreq, nback)


@pytest.mark.parametrize("pat,repl", [(b"\\w+\\d{2}", b"<W>"), (b"\\d+(\\.\\d+)?", b"N"), (b"(foo|foobar)", b""),
                                      (b"[a-z]+(-[a-z]+)*", b"_"),
                                      # required-byte plans: sub walks match_next, not the findall route
                                      (b"[a-z]+@[a-z]+", b"<mail>"), (b"\\d{3}-\\d{4}", b"###-####")])
@pytest.mark.parametrize("count", [0, 2])
def test_sub_from_stepper_spans(pat, repl, count):
    """Non-streamable but steppable plans: sub is assembled from the windowed stepper's findall."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat) + count)
    texts = _random_texts(rng, 200, 100, b"abcfoxr0123456789.- ") + _random_texts(rng, 20, 900, b"abfoxr019.-  ") + [
        b"", b"foox", b"foobarx", b"3.14 2.", b"a-b-c", b"ab12"]
    rx = M.compile_regex(pat)
    d = rx.describe()
    assert "device.streamable=no" in d and "step_search=1" in d
    texts = texts + [b"aaa@bbb@ccc x@y", b"12345-6789 555-1234", b"a@b", b"555-12345"]
    got = rx.sub(repl, texts, count)
    assert M.load_library().mrx_last_kernel_name() == b"k_subs_wave"
    with generic_kernels():
        want = rx.sub(repl, texts, count)
    assert got == want
    for i in range(0, len(texts), 5):
        assert got[i] == O.sub(pat, repl, texts[i], count), (pat, texts[i])


@pytest.mark.parametrize("pat", [b"[^a-z]+", b"[^0-9]+x", b".+", b"a.c", b"\\w+", b"[\\x80-\\xff]+", b"\\S+" if False else b"\\s+\\w",
                                 b"[^abc]{2}", b"(a|[^a])b"])
def test_full_byte_range_texts(pat):
    """Texts over all 256 byte values (NUL and bytes >= 0x80 included): tables are 256 wide
    upstream (dfa.mojo:215-254) and nothing may sign-extend a byte."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat))
    texts = []
    for _ in range(150):
        k = int(rng.integers(0, 300))
        t = rng.integers(0, 256, size=k).astype(np.uint8)
        if k > 8 and rng.random() < 0.5:           # sprinkle the pattern's own ASCII bytes in
            idx = rng.integers(0, k, size=k // 3)
            t[idx] = rng.choice(np.frombuffer(b"abcx019 \\n_", dtype=np.uint8), size=idx.size)
        texts.append(t.tobytes())
    texts += [bytes(range(256)), bytes([0]) * 40, bytes([255]) * 40 + b"x", b""]
    try:
        rx = M.compile_regex(pat)
    except M.RegexSyntaxError:
        with pytest.raises(Exception):
            O.compile_regex(pat)
        return
    for op in ("findall", "search", "match_first"):
        try:
            if op == "findall":
                got = rx.findall_lists(texts)
            else:
                s, e = (rx.match_next if op == "search" else rx.match_first)(texts)
                got = [(int(a), int(b)) if a >= 0 else None for a, b in zip(s, e)]
        except M.UnsupportedPattern:
            continue
        for t, g in zip(texts, got):
            assert g == getattr(O, op)(pat, t), (pat, op, t)


def test_stepper_slot_overflow_is_rewalked():
    """k_wstep parks up to 32 spans per text; texts with more are re-walked by the emit pass."""
    _need_gpu()
    pat = b"\\d+(\\.\\d+)?"
    rx = M.compile_regex(pat)
    assert "device.streamable=no" in rx.describe() and "device.steppable=yes" in rx.describe()
    texts = [b"1 2.5 x " * 40, b"7", b"", b"3.14 " * 33, b"9 " * 32, b"no digits here", b"4 " * 31, b"1.2.3.4 " * 50]
    texts = texts * 20
    got = rx.findall_lists(texts)
    for t, g in zip(texts[:8], got[:8]):
        assert g == O.findall(pat, t), t
    assert got[8:16] == got[:8]
    assert len(got[0]) == 80 and len(got[4]) == 32 and len(got[6]) == 31


def test_concurrent_calls_on_one_handle():
    """include/mrx.h: a handle is immutable after mrx_compile(), so concurrent batch calls on one
    handle are safe (per-thread scratch arenas, no shared mutable state).  ctypes drops the GIL
    during the calls, so these threads really overlap."""
    _need_gpu()
    import threading
    rx = M.compile_regex(b"[a-z]+\\d+")
    rng = np.random.default_rng(5)
    batches, want = [], []
    for k in range(4):
        texts = _random_texts(rng, 400 + 50 * k, 300, b"abcxyz0123 ")
        b = M.DeviceBatch.from_texts(texts)
        batches.append(b)
        want.append(rx._dev_findall(b))
    errs = []

    def work(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(25):
                    pre, sp, tot = rx._dev_findall(batches[k])
                    wp, ws, wt = want[k]
                    if tot != wt or not torch.equal(pre, wp) or not torch.equal(sp[:tot], ws[:wt]):
                        errs.append(k)
                        return
                    cnt = rx.count(batches[k])
                    if not torch.equal(cnt.to(torch.int64), wp[1:] - wp[:-1]):
                        errs.append(-k - 1)
                        return
        except Exception as exc:  # noqa: BLE001
            errs.append(repr(exc))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs


def test_findall_count_only_and_capacity_error():
    """span_cap == 0: totals and CSR offsets without spans; a too-small buffer reports the need."""
    _need_gpu()
    import ctypes as C
    rx = M.compile_regex(b"[a-z]+\\d+")
    texts = [b"a1 b2 c3", b"", b"zz9"] * 50
    b = M.DeviceBatch.from_texts(texts)
    lib = M.load_library()
    prefix = torch.empty(b.n + 1, dtype=torch.int64, device="cuda")
    total = C.c_int64(0)
    rc = lib.mrx_findall_dev(rx._h, C.c_void_p(b.data.data_ptr()), C.c_void_p(b.offsets.data_ptr()), b.n,
                             C.c_void_p(prefix.data_ptr()), None, 0, C.byref(total), None)
    assert rc == M.api.MRX_E_CAPACITY and total.value == 200      # need reported, nothing written
    assert prefix[-1].item() == 200 and prefix[3].item() == 4
    spans = torch.full((10, 2), -5, dtype=torch.int32, device="cuda")
    rc = lib.mrx_findall_dev(rx._h, C.c_void_p(b.data.data_ptr()), C.c_void_p(b.offsets.data_ptr()), b.n,
                             C.c_void_p(prefix.data_ptr()), C.c_void_p(spans.data_ptr()), 10, C.byref(total), None)
    assert rc == M.api.MRX_E_CAPACITY and total.value == 200
    assert spans[:4].tolist() == [[0, 2], [3, 5], [6, 8], [0, 3]]   # what fits is written, in order


@pytest.mark.parametrize("pat", [b"\\d{3}-\\d{4}", b"\\w+@\\w+\\.com", b"[a-z]+@[a-z]+", b"[a-c]+x[0-9]+y", b"bar\\d[xyz]",
                                 b"[a-zA-Z0-9._%+-]+@[a-zA-Z0-9.-]+\\.[a-z]{2,}", b"[0-9]+:[0-9]+"])
def test_required_byte_route_on_the_stepper(pat):
    """HybridMatcher._match_all_required_byte (matcher.mojo:864-898) as the stepper's second route:
    findall / count against the literal restatement on every text and the oracle on a sample --
    including its quirks (the match must pass the hit; starts back up into earlier matches)."""
    _need_gpu()
    rx = M.compile_regex(pat)
    d = rx.describe()
    if "required-byte route" not in d:
        pytest.skip("pattern is not on the required-byte route: " + d.split("device.steppable")[1][:60])
    lib = M.load_library()
    rng = np.random.default_rng(zlib.crc32(pat))
    al = b"abcxyz0123456789@.-: " + bytes(c for c in pat if chr(c).isalnum() or c in b"@.-:") * 3
    texts = _random_texts(rng, 250, 80, al) + _random_texts(rng, 30, 700, al) + [
        b"", b"555-1234", b"12345-6789", b"call 555-1234 or 555-12345-6789", b"a@b.com", b"aaa@bbb@ccc.com x@y.com",
        b"user@example.com,other@test.org", b"@", b"-", b"12:30 1:2:3", b"bar5x bar9q", b"@@@@", b"a@" * 50]
    with no_streaming_kernels():
        got = rx.findall_lists(texts)
        assert lib.mrx_last_kernel_name() == b"k_step_count"
    with generic_kernels():
        want = rx.findall_lists(texts)
        assert lib.mrx_last_kernel_name() == b"k_findall_count"
    assert got == want
    for i in range(0, len(texts), 4):
        assert got[i] == O.findall(pat, texts[i]), (pat, texts[i])
    assert sum(len(g) for g in got) >= 1


@pytest.mark.parametrize("pat", [b"\\d{3}-\\d{4}", b"\\w+@\\w+\\.com", b"[a-z]+@[a-z]+", b"[a-c]+x[0-9]+y", b"bar\\d[xyz]",
                                 b"[a-zA-Z0-9._%+-]+@[a-zA-Z0-9.-]+\\.[a-z]{2,}", b"[0-9]+:[0-9]+",
                                 b"\\d{3}-\\d{3}-\\d{4}", b"[0-9]+\\.[0-9]+"])
def test_required_byte_route_one_wavefront_per_text(pat):
    """k_req_wave (the required-byte route with a wavefront sweeping each text) against the lane-per-text
    stepper and the oracle: short texts, texts of several 1 KiB blocks, long runs that make the back-up
    and the walks leave the three-block window, CSR (unaligned) and fixed-pitch batches, findall with
    more matches than slots, count."""
    _need_gpu()
    rx = M.compile_regex(pat)
    if "required-byte route" not in rx.describe():
        pytest.skip("pattern is not on the required-byte route")
    lib = M.load_library()
    rng = np.random.default_rng(zlib.crc32(pat) + 7)
    al = b"abcxyz0123456789@.-: " + bytes(c for c in pat if chr(c).isalnum() or c in b"@.-:") * 3
    texts = (_random_texts(rng, 60, 90, al) + _random_texts(rng, 24, 5000, al) + _random_texts(rng, 6, 40000, al) + [
        b"", b"-", b"@", b"555-1234", b"a@b.com",
        b"7" * 5000 + b"-1234 " + b"5" * 3000 + b"-" + b"8" * 4000,          # back-up over runs longer than the window
        b"ab" * 2000 + b"@" + b"cd" * 3000 + b".com " + b"x@y.com" * 300,     # walks longer than the window
        (b"call 555-123-4567 or 12:30 or a@b.com, 1.5 and cx9y bar5x " * 200)])
    with long_text_kernels(3):
        got = rx.findall_lists(texts)
        assert lib.mrx_last_kernel_name() == b"k_req_wave"
    with long_text_kernels(2):
        want = rx.findall_lists(texts)
        assert lib.mrx_last_kernel_name() in (b"k_step_count", b"k_mwalk")   # (k_mwalk: the route's one-pass table)
        with multiwalk(2):
            want2 = rx.findall_lists(texts)
            assert lib.mrx_last_kernel_name() == b"k_step_count"
    assert got == want == want2
    for i in list(range(0, 60, 5)) + list(range(60, len(texts))):
        assert got[i] == O.findall(pat, texts[i]), (pat, i)
    assert max(len(g) for g in got) > 32          # more matches than slots: the emit pass ran
    # device-resident batches: CSR and fixed pitch (aligned and not), count
    data, offsets = M.api.pack_texts(texts)
    b = M.DeviceBatch(torch.from_numpy(data).cuda(), torch.from_numpy(offsets).cuda())
    with long_text_kernels(3):
        cnt = rx.count(b).cpu().numpy()
        assert lib.mrx_last_kernel_name() == b"k_req_wave"
    assert cnt.tolist() == [len(g) for g in got]
    for pitch in (4096, 4099):
        n = 37
        arr = rng.choice(np.frombuffer(al, dtype=np.uint8), size=(n, pitch)).astype(np.uint8)
        lens = rng.integers(0, pitch + 1, size=n).astype(np.int32)
        sb = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch, length=pitch,
                                   lens=torch.from_numpy(lens).cuda())
        pre, sp, tot = rx._dev_findall(sb)           # by average length: 4 KiB rows are cut into pieces, or take the wavefront kernel
        assert lib.mrx_last_kernel_name() in (b"k_req_wave", b"k_step_count_pieces", b"k_mwalk_pieces"), lib.mrx_last_kernel_name()
        with long_text_kernels(3):
            pre3, sp3, tot3 = rx._dev_findall(sb)
            assert lib.mrx_last_kernel_name() == b"k_req_wave"
        assert tot == tot3 and torch.equal(pre, pre3) and torch.equal(sp[:tot], sp3[:tot3])
        with long_text_kernels(2):
            pre2, sp2, tot2 = rx._dev_findall(sb)
        assert tot == tot2 and torch.equal(pre, pre2) and torch.equal(sp[:tot], sp2[:tot2])


@pytest.mark.parametrize("pat", [b"\\d+(\\.\\d+)?", b"\\w+\\d{2}", b"(foo|foobar)", b"\\(?\\d{3}\\)?[\\s.-]?\\d{3}[\\s.-]?\\d{4}",
                                 b"[A-Z]{2,4}[0-9]{3,5}", b"(hello|world|test|demo|sample)[0-9]{3}[a-z]{2}",
                                 b"8(?:00|33|44|55|66|77|88)[2-9]\\d{6}", b"[a-z]+@[a-z]+", b"\\d{3}-\\d{3}-\\d{4}",
                                 # more than 96 states: class-indexed table (PF_STEP_BIG), wavefront kernel only
                                 b"(?:2(?:0[1-35-9]|1[02-9]|2[03-57-9]|3[1459]|4[08]|5[1-46]|6[0279]|7[0269]|8[13])|3(?:0[1-47-9]|1[02-9]|"
                                 b"2[0135-79]|3[0-24679]|4[167]|5[0-2]|6[01349]|8[056])|4(?:0[124-9]|1[02-579]|2[3-5]|3[0245]|4[023578]|58|"
                                 b"6[349]|7[0589]|8[04])|5(?:0[1-47-9]|1[0235-8]|20|3[0149]|4[01]|5[179]|6[1-47]|7[0-5]|8[0256]))[2-9]\\d{6}"])
def test_stepper_one_wavefront_per_text(pat):
    """k_req_wave<., 0> (DFAEngine.match_all / match_next with a wavefront sweeping each text) against the
    lane-per-text stepper and the oracle: findall, count, search and sub (which walks match_next even on
    required-byte plans)."""
    _need_gpu()
    rx = M.compile_regex(pat)
    d = rx.describe()
    if "step_search=1" not in d:
        pytest.skip("pattern is not on the stepper")
    req = "required-byte route" in d
    lib = M.load_library()
    rng = np.random.default_rng(zlib.crc32(pat) + 11)
    al = b"abfoxHELOWRDT0123456789@.-() " + bytes(c for c in pat if chr(c).isalnum()) * 2
    texts = (_random_texts(rng, 50, 90, al) + _random_texts(rng, 20, 5000, al) + _random_texts(rng, 4, 30000, al) + [
        b"", b"8", b"foo", b"5" * 9000 + b"-" + b"6" * 3000, b"AB" * 4000 + b"12345 " + b"hello123ab" * 500,
        b"foo" * 700, b"a@b " * 600, b"x1.5" * 500, b"x1 " * 900,   # a match every 3-4 bytes: more than a wide slot row holds
        (b"(555) 123-4567 8005551234 555-123-4567 2125551234 foobar x1.5 ABC1234 world456cd a@b " * 150)])
    with no_streaming_kernels(), long_text_kernels(3):
        got = rx.findall_lists(texts)
        k1 = lib.mrx_last_kernel_name()
        ss, se = rx.match_next(texts)
        k2 = lib.mrx_last_kernel_name()
        sub = rx.sub(b"<#>", texts, 3)
    # the second implementation: the lane-per-text stepper, or the literal restatement for big tables
    with (generic_kernels() if "big_table=1" in d else no_streaming_kernels()), long_text_kernels(2):
        want = rx.findall_lists(texts)
        ws, we = rx.match_next(texts)
        wsub = rx.sub(b"<#>", texts, 3)
    assert k1 == b"k_req_wave" and k2 == b"k_req_wave_search"
    assert got == want and ss.tolist() == ws.tolist() and se.tolist() == we.tolist() and sub == wsub
    for i in list(range(0, 50, 5)) + list(range(50, len(texts))):
        assert got[i] == O.findall(pat, texts[i]), (pat, i)
        m = O.search(pat, texts[i])
        assert (None if ss[i] < 0 else (int(ss[i]), int(se[i]))) == m, (pat, i)
        assert sub[i] == O.sub(pat, b"<#>", texts[i], 3), (pat, i)
    assert sum(len(g) for g in got) >= 1


@pytest.mark.parametrize("mw", [2, 0, 3])
@pytest.mark.parametrize("pat", [b"\\d+(\\.\\d+)?", b"\\w+\\d{2}", b"[a-z]+@[a-z]+", b"(foo|foobar)",
                                 b"[A-Z]{10,20}[0-9]{15,25}", b"foo|[a-z]{3}\\d|[ab]", b"[0-9]+\\.[0-9]+", b"(?:xy){4}@{2}"])
@pytest.mark.parametrize("n,pitch,var", [(130, 256, True), (70, 50, True), (64, 1024, False), (3, 7, True),
                                         (40, 2304, True), (33, 2051, False),   # >= 2 KiB: slot rows sized by the text
                                         (192, 512, False)])   # whole wavefronts of full rows: the kernels' paths without the per-byte frame test
def test_stepper_on_fixed_pitch_batches(pat, n, pitch, var, mw):
    """k_wstep's frame form (mw = 2) and k_mwalk's (mw = 0, plans that have the multi-walk table) on fixed-pitch
    batches (aligned and not, with and without lens)."""
    _need_gpu()
    rng = np.random.default_rng(n * 31 + pitch + zlib.crc32(pat))
    al = np.frombuffer(b"abcfoxr0123456789.-@ xyAZ" + bytes(c for c in pat if chr(c).isalnum()) * 2, dtype=np.uint8)
    arr = rng.choice(al, size=(n, pitch)).astype(np.uint8)
    if pitch >= 256:   # something for the counted repetitions to find
        for i in range(0, n, 4):
            k = int(rng.integers(0, pitch - 60))
            arr[i, k:k + 14] = ord("Q"); arr[i, k + 14:k + 34] = ord("7")
            arr[i, k + 40:k + 48] = np.frombuffer(b"xyxyxyxy", dtype=np.uint8); arr[i, k + 48:k + 50] = ord("@")
    lens = rng.integers(0, pitch + 1, size=n).astype(np.int32) if var else None
    rx = M.compile_regex(pat)
    assert "device.streamable=no" in rx.describe()
    d = torch.from_numpy(arr).cuda().reshape(-1)
    batch = M.DeviceBatch.strided(d, pitch, length=pitch, lens=torch.from_numpy(lens).cuda() if var else None)
    lib = M.load_library()
    has_mw = mw != 2 and "multiwalk=yes" in rx.describe()
    has_tries = mw != 2 and "tries_walk=yes" in rx.describe() and "required-byte route" not in rx.describe()   # (findall / count only)
    with long_text_kernels(2), multiwalk(mw):   # this test is about the lane-per-text kernels
        pre, sp, tot = rx._dev_findall(batch)
        has_bk = mw != 2 and not has_mw and "backset=yes" in rx.describe() and "required-byte route" not in rx.describe()
        assert lib.mrx_last_kernel_name() == (b"k_mwalk" if has_mw or has_tries else b"k_backscan+k_step_count" if has_bk else b"k_step_count")
        ss, se = rx.match_next(batch)
        if has_mw and b"@" not in pat:
            assert lib.mrx_last_kernel_name() == b"k_mwalk_search"
        cnt = rx.count(batch)
    with generic_kernels():
        gpre, gsp, gtot = rx._dev_findall(batch)
        gss, gse = rx.match_next(batch)
    assert tot == gtot and torch.equal(pre, gpre) and torch.equal(sp[:tot], gsp[:tot])
    assert torch.equal(ss, gss) and torch.equal(se, gse)
    assert torch.equal(cnt.to(torch.int64), pre[1:] - pre[:-1])
    pre_h, sp_h = pre.cpu().numpy(), sp.cpu().numpy()
    for i in range(0, n, 9):
        t = arr[i, : (lens[i] if var else pitch)].tobytes()
        assert [tuple(int(x) for x in r) for r in sp_h[pre_h[i]:pre_h[i + 1]]] == O.findall(pat, t), (pat, i)


@pytest.mark.parametrize("pat", [b"z*", b"x*", b"(abc)*", b"a+b*", b"", b"a**", b"http?", b"ca*t", b"(?:abc)?", b"cat|(dog){0,2}\\d?",
                                 b"1{2}.{3}|hello|xy|(?:abc)?", b"(ab|cd)*", b"(?:a|b)*c?", b"x?y?", b"(foo)?(bar)?",
                                 b"\\d*", b"[a-c]*x?", b"\\s?", b"a{0,2}", b"[abc]*[xyz]*", b"b{0,2}c?"])
def test_empty_match_plans_on_the_stepper(pat):
    """Plans whose start state accepts and that have no first-byte matcher (dfa.mojo:2118-2130,
    pikevm.mojo:805-817): every position yields a match, possibly empty; findall / count run the windowed
    stepper's EMPTY form (count, then emit) -- or, when no walk of the plan ever reads beyond its match (every state
    accepts: `empty_walk=1`), ONE pass on k_mwalk with up to two reports per byte (round 4) -- against the literal
    restatement on every text, the stepper, and the oracle."""
    _need_gpu()
    rx = M.compile_regex(pat)
    d = rx.describe()
    rng = np.random.default_rng(zlib.crc32(pat) + 3)
    al = b"abcxyzt ht p1dogcafoobar" + bytes(c for c in pat if chr(c).isalnum()) * 2
    texts = _random_texts(rng, 250, 40, al) + _random_texts(rng, 30, 700, al) + [
        b"", b"a", b"aa", b"aab", b"abcabc", b"abx", b"xyz", b"zzz", b"http", b"htt", b"ct", b"caaat", b"cat", b"dogdog1", b"11abc",
        b"hello", b"q" * 129, b"z" * 128, b"z" * 127 + b"q", b"ab" * 100, b"foobar", b"barfoo"]
    lib = M.load_library()
    batch = M.DeviceBatch.from_texts(texts)
    try:
        lists = rx.findall_lists(texts)
    except M.UnsupportedPattern:
        pytest.skip("search not supported for this plan")
    used = lib.mrx_last_kernel_name()
    cnt = rx.count(batch).cpu().numpy()
    used_cnt = lib.mrx_last_kernel_name()
    with generic_kernels():
        assert rx.findall_lists(texts) == lists, pat
        assert (rx.count(batch).cpu().numpy() == cnt).all(), pat
    assert [len(x) for x in lists] == [int(c) for c in cnt]
    for i in range(0, len(texts), 3):
        assert lists[i] == O.findall(pat, texts[i]), (pat, texts[i])
    for t, got in zip(texts[-22:], lists[-22:]):
        assert got == O.findall(pat, t), (pat, t)
    if "empty_walk=1" in d:
        assert used == b"k_mwalk" and used_cnt == b"k_mwalk", (used, used_cnt)
        lib.mrx_debug_multiwalk(2)   # the stepper's EMPTY form on the same batch
        try:
            assert rx.findall_lists(texts) == lists, pat
            assert lib.mrx_last_kernel_name() == b"k_estep_count"
            assert (rx.count(batch).cpu().numpy() == cnt).all(), pat
        finally:
            lib.mrx_debug_multiwalk(0)
    elif "empty_matches=1" in d:
        assert used == b"k_estep_count" and used_cnt == b"k_estep_count", (used, used_cnt)
    # fixed pitch, and the C ABI's capacity protocol on a batch where nearly every byte is a match
    if "empty_matches=1" in d:
        L = 96
        rows = [t[:L].ljust(L, b"q") for t in texts[:128]]
        sb = M.DeviceBatch.strided(torch.tensor(list(b"".join(rows)), dtype=torch.uint8, device="cuda"), L, length=L)
        pre, sp, tot = rx._dev_findall(sb)
        pre_h, sp_h = pre.cpu().numpy(), sp.cpu().numpy()
        for i in range(0, 128, 5):
            assert [tuple(int(x) for x in r) for r in sp_h[pre_h[i]:pre_h[i + 1]]] == O.findall(pat, rows[i]), (pat, rows[i])


@pytest.mark.parametrize("pat", [b"^abc$", b"xyz$", b"a$", b"^hello$", b"^a$", b"^ab?c$", b"(foo|bar)$", b"^(foo|bar)x$", b"[a-z]+$",
                                 b"^[a-z]*$", b"hello$"])
def test_end_anchored_dfa_plans_on_the_anchored_automaton(pat):
    """'$' on the DFA route (dfa.mojo:2019-2024: the last accepting position of the greedy walk must be the
    end of the text): match_first / is_match -- and search / findall / count of '^...$' plans, which only
    try position 0 -- run the anchored automaton with end-of-text flags on the streaming kernel.  Plans
    where the _try_match_simd shortcut or the pure-literal return comes first keep the literal restatement;
    every pattern here is checked against it and the oracle either way."""
    _need_gpu()
    rx = M.compile_regex(pat)
    d = rx.describe()
    rng = np.random.default_rng(zlib.crc32(pat))
    al = b"abcxyzfor helo" + bytes(c for c in pat if chr(c).isalnum())
    texts = _random_texts(rng, 200, 12, al) + _random_texts(rng, 40, 300, al) + [
        b"", b"abc", b"abcabc", b"xabc", b"abcx", b"xyz", b"xxyz", b"xyzx", b"a", b"aa", b"ba", b"ab", b"hello", b"hello!", b"ac",
        b"foo", b"bar", b"foox", b"barx", b"xfoo", b"q" * 200 + b"xyz", b"q" * 127 + b"a", b"abc" + b"\n", b"z" * 500]
    lib = M.load_library()
    batch = M.DeviceBatch.from_texts(texts)
    fs, fe = rx.match_first(texts)
    k_first = lib.mrx_last_kernel_name()
    im = rx.is_match(texts)
    try:
        ss, se = rx.match_next(texts)
        k_search = lib.mrx_last_kernel_name()
        lists = rx.findall_lists(texts)
        cnt = rx.count(batch).cpu().numpy()
        searchable = True
    except M.UnsupportedPattern:
        searchable = False
    with generic_kernels():
        gfs, gfe = rx.match_first(texts)
        gim = rx.is_match(texts)
        assert np.array_equal(fs, gfs) and np.array_equal(fe, gfe) and np.array_equal(np.asarray(im), np.asarray(gim)), pat
        if searchable:
            gss, gse = rx.match_next(texts)
            assert np.array_equal(ss, gss) and np.array_equal(se, gse), pat
            assert rx.findall_lists(texts) == lists and (rx.count(batch).cpu().numpy() == cnt).all(), pat
    orx = O.compile_regex(pat)
    for i, t in enumerate(texts):
        w = O.match_first(pat, t)
        assert (int(fs[i]), int(fe[i])) == (w if w else (-1, -1)), (pat, "match_first", t)
        assert bool(im[i]) == bool(orx.is_match(t, 0)), (pat, "is_match", t)
        if searchable:
            w = O.search(pat, t)
            assert (int(ss[i]), int(se[i])) == (w if w else (-1, -1)), (pat, "search", t)
            assert lists[i] == O.findall(pat, t), (pat, "findall", t)
    if "engine_type=DFA" in d and "pure_literal=0" in d and "has_matcher=0" in d:
        assert "device.first_stream=yes" in d and k_first == b"k_stream_first", (d, k_first)
        if "start_anchor=1" in d and searchable:
            assert k_search == b"k_stream_first", k_search


@pytest.mark.parametrize("pat", [b"^[a-z]+\\d+", b"^\\d+", b"^[a-z]+[0-9]+x", b"^(foo|bar)", b"^a+b", b"^[a-z]*[0-9]*", b"^hello"])
def test_start_anchored_search_uses_the_anchored_automaton(pat):
    """'^'-anchored DFA plans: match_next only tries position 0, so search runs the anchored automaton
    on the streaming kernel (pure literals keep the generic route: simd_search is not anchored)."""
    _need_gpu()
    rx = M.compile_regex(pat)
    rng = np.random.default_rng(zlib.crc32(pat))
    texts = _random_texts(rng, 200, 60, b"abfor019x ") + [b"", b"abc123", b"123", b"foo", b"barx", b"aab", b"xxhello", b"hello"]
    try:
        s, e = rx.match_next(texts)
    except M.UnsupportedPattern:
        pytest.skip("search not supported for this plan")
    used = M.load_library().mrx_last_kernel_name()
    with generic_kernels():
        gs, ge = rx.match_next(texts)
    assert np.array_equal(s, gs) and np.array_equal(e, ge), pat
    for i, t in enumerate(texts):
        w = O.search(pat, t)
        assert (int(s[i]), int(e[i])) == (w if w else (-1, -1)), (pat, t)
    if pat != b"^hello" and "engine_type=DFA" in rx.describe():
        assert used == b"k_stream_first", used


@pytest.mark.parametrize("pat", [b"[a-z]+\\d+", b"\\d+", b"hello", b"(\\d{3})(\\d{3})(\\d{4})", b"(x|y|foo|bar)+", b"a",
                                 b"abab", b"[a-c]+[0-9]+[x-z]+[0-9]+"])
@pytest.mark.parametrize("shape", ["strided", "strided_lens", "csr", "frame"])
def test_fused_findall_equals_three_launch_form(pat, shape):
    """ST_FUSED (opt-in: records kept by the wavefront, tickets, look-back over task and group words for
    the CSR base, spans written in place) against the default three-launch form on the same batches:
    identical CSR offsets and spans.  Sizes
    chosen so that several tickets, a partial last wavefront, empty texts, wavefronts below one LDS
    tile, across several tiles and on the direct-store path all occur."""
    _need_gpu()
    lib = M.load_library()
    rx = M.compile_regex(pat)
    assert "device.streamable=yes" in rx.describe()
    rng = np.random.default_rng(zlib.crc32(pat) + len(shape))
    al = np.frombuffer(b"abcxyz0189 -fobar5" + bytes(c for c in pat if chr(c).isalnum()) * 2, dtype=np.uint8)
    for n, pitch in ((1, 16), (63, 48), (64 * 9 + 5, 208), (4099, 64), (300, 1024), (70, 4096)):
        if shape == "frame":
            pitch += 3   # not a multiple of 16: frame form with a fixed pitch
        arr = rng.choice(al, size=(n, pitch)).astype(np.uint8)
        for i in range(0, n, 3):   # dense rows: a match every two or three bytes
            body = np.frombuffer((b"a1 " * pitch)[:pitch], dtype=np.uint8)
            arr[i] = body if i % 2 == 0 else np.frombuffer((b"x" * pitch), dtype=np.uint8)
        if shape in ("strided", "frame"):
            batch = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch, length=pitch)
        elif shape == "strided_lens":
            lens = rng.integers(0, pitch + 1, size=n).astype(np.int32)
            lens[:: 11] = 0
            batch = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch, lens=torch.from_numpy(lens).cuda())
        else:
            lens = rng.integers(0, pitch + 1, size=n)
            lens[:: 7] = 0
            batch = M.DeviceBatch.from_texts([arr[i, : lens[i]].tobytes() for i in range(n)])
        with long_text_kernels(2):   # (the 4 KiB texts would otherwise be cut into pieces)
            with fused_findall(), stream_bits(0):
                p1, s1, t1 = rx._dev_findall(batch)
                assert lib.mrx_last_kernel_name() == b"k_stream_findall_fused"
            lib.mrx_debug_dense_rows(2)   # (records: a handle that has seen a batch full of matches would take event rows)
            try:
                with stream_bits(0):
                    p3, s3, t3 = rx._dev_findall(batch)
            finally:
                lib.mrx_debug_dense_rows(0)
            assert lib.mrx_last_kernel_name() == b"k_stream_findall"
        assert t1 == t3 and torch.equal(p1, p3) and torch.equal(s1[:t1], s3[:t3]), (pat, shape, n, pitch)
        # a span buffer that is too small: what fits is written in order, the need is reported
        if t1 > 8:
            cap = t1 // 2
            pre = torch.empty(batch.n + 1, dtype=torch.int64, device="cuda")
            sp = torch.full((cap + 4, 2), -9, dtype=torch.int32, device="cuda")
            with long_text_kernels(2), fused_findall(), stream_bits(0):
                rx.findall_async(batch, (pre, sp[:cap]))
                assert lib.mrx_last_kernel_name() == b"k_stream_findall_fused"
            torch.cuda.synchronize()
            assert int(pre[-1]) == t1 and torch.equal(pre, p1)
            assert torch.equal(sp[:cap], s1[:cap]) and bool((sp[cap:] == -9).all())


STREAM_BITS_PATTERNS = [b"[a-z]+\\d+", b"\\d+", b"[0-9]+", b"hello", b"a", b"abab", b"[a-z]+", b"[a-c]+[0-9]+",
                        b"[^0-9]+", b"[a-z]+[0-9]+[a-z]+"]


@pytest.mark.parametrize("pat", STREAM_BITS_PATTERNS)
@pytest.mark.parametrize("shape", ["strided", "strided_short", "strided_lens"])
def test_stream_bits_equals_three_launch_form_and_oracle(pat, shape):
    """k_stream_bits (round 4: one launch, every text's event bits in registers, tickets + look-back, spans written at
    their final place; texts of at most 1 KiB at a 16-byte aligned pitch) against scan -> prefix sums -> decode on
    the same batches, and against the oracle text by text (DFAEngine.match_all, dfa.mojo:2028-2130).  Sizes chosen so
    that one to many tickets, a partial last task, a partial last 64-text group, empty texts, texts that end inside
    a chunk, all three register-file sizes (256 / 512 / 1024 bytes), tiles of more spans than one window holds and
    matches that run to the end of the text all occur."""
    _need_gpu()
    lib = M.load_library()
    rx = M.compile_regex(pat)
    if "device.streamable=yes" not in rx.describe():
        pytest.skip("not a streamable plan")
    rng = np.random.default_rng(zlib.crc32(pat) + len(shape))
    al = np.frombuffer(b"abcxyz0189 -fobar5" + bytes(c for c in pat if chr(c).isalnum()) * 2, dtype=np.uint8)
    took_bits = 0
    for n, pitch in ((1, 16), (63, 48), (64 * 9 + 5, 208), (4099, 64), (300, 1024), (257, 256), (1000, 512), (130, 528),
                     (64 * 4 * 7, 1024), (5000, 128), (3, 1024)):
        arr = rng.choice(al, size=(n, pitch)).astype(np.uint8)
        for i in range(0, n, 3):   # dense rows: a match every two or three bytes
            body = np.frombuffer((b"a1 " * pitch)[:pitch], dtype=np.uint8)
            arr[i] = body if i % 2 == 0 else np.frombuffer((b"x" * pitch), dtype=np.uint8)
        if n > 200:   # a whole 64-text group of the densest rows there are: several tile windows
            arr[64:128] = np.frombuffer((b"a1" * pitch)[:pitch], dtype=np.uint8)
        for i in range(1, n, 5):   # a match that runs to the end of the row
            arr[i, -3:] = np.frombuffer(b"a12", dtype=np.uint8)
        data = torch.from_numpy(arr).cuda().reshape(-1)
        if shape == "strided":
            lens = np.full(n, pitch)
            batch = M.DeviceBatch.strided(data, pitch, length=pitch)
        elif shape == "strided_short":
            L = max(0, pitch - 5)
            lens = np.full(n, L)
            batch = M.DeviceBatch.strided(data, pitch, length=L)
        else:
            lens = rng.integers(0, pitch + 1, size=n).astype(np.int32)
            lens[:: 11] = 0
            lens[1:: 11] = pitch
            batch = M.DeviceBatch.strided(data, pitch, lens=torch.from_numpy(lens).cuda())
        with stream_bits(1):
            p1, s1, t1 = rx._dev_findall(batch)
            k1 = lib.mrx_last_kernel_name()
        p3, s3, t3 = rx._dev_findall(batch)
        assert lib.mrx_last_kernel_name() != b"k_stream_bits"
        assert t1 == t3 and torch.equal(p1, p3) and torch.equal(s1[:t1], s3[:t3]), (pat, shape, n, pitch, k1)
        if k1 != b"k_stream_bits":
            continue
        took_bits += 1
        pre, sp = p1.cpu().numpy(), s1.cpu().numpy()
        for i in list(range(0, n, max(1, n // 40))) + [n - 1]:
            want = O.findall(pat, arr[i, : lens[i]].tobytes())
            have = [tuple(int(x) for x in r) for r in sp[pre[i]:pre[i + 1]]]
            assert have == want, (pat, shape, n, pitch, i)
        # a span buffer that is too small: what fits is written in order, the need is reported
        if t1 > 8:
            cap = t1 // 2
            pre_t = torch.empty(batch.n + 1, dtype=torch.int64, device="cuda")
            sp_t = torch.full((cap + 4, 2), -9, dtype=torch.int32, device="cuda")
            with stream_bits(1):
                rx.findall_async(batch, (pre_t, sp_t[:cap]))
            assert lib.mrx_last_kernel_name() == b"k_stream_bits"
            torch.cuda.synchronize()
            assert int(pre_t[-1]) == t1 and torch.equal(pre_t, p1)
            assert torch.equal(sp_t[:cap], s1[:cap]) and bool((sp_t[cap:] == -9).all())
    if pat in (b"[a-z]+\\d+", b"\\d+"):   # (automata of more than four states keep the three-launch form: hello, abab ...)
        assert took_bits > 0, "the headline plans must take the one-launch form"


def test_stream_bits_back_to_back_calls_on_two_streams():
    """Tickets and look-back words are zeroed per call and live in the per-stream scratch arena: many calls in a
    row on two streams (their launches overlap on the device, so neither grid is resident as a whole -- tasks
    are handed out by ticket) give the same answer, and the arena stops growing."""
    _need_gpu()
    lib = M.load_library()
    rx = M.compile_regex(b"[a-z]+\\d+")
    batch_t = make_c2_batch(1 << 15, 1024, seed=5, device="cuda")
    batch = M.DeviceBatch.strided(batch_t.reshape(-1), 1024, length=1024)
    p0, s0, t0 = rx._dev_findall(batch)
    assert lib.mrx_last_kernel_name() == b"k_stream_findall"
    size0 = None
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [(torch.empty_like(p0), torch.empty_like(s0)) for _ in streams]
    with stream_bits(1):
        for it in range(40):
            k = it % 2
            with torch.cuda.stream(streams[k]):
                rx.findall_async(batch, outs[k])
                assert lib.mrx_last_kernel_name() == b"k_stream_bits"
            if it == 9:
                torch.cuda.synchronize()
                size0 = lib.mrx_debug_scratch_bytes()
        torch.cuda.synchronize()
    assert lib.mrx_debug_scratch_bytes() == size0
    for pre, sp in outs:
        assert torch.equal(pre, p0) and torch.equal(sp[:t0], s0[:t0])


def test_fused_findall_back_to_back_calls_reuse_their_scratch():
    """The ticket word and descriptors are zeroed per call, the record regions are reused task after
    task and call after call: many calls in a row (two streams) give the same answer, and the arena
    stops growing."""
    _need_gpu()
    lib = M.load_library()
    rx = M.compile_regex(b"[a-z]+\\d+")
    batch_t = make_c2_batch(1 << 14, 1024, seed=5, device="cuda")
    batch = M.DeviceBatch.strided(batch_t.reshape(-1), 1024, length=1024)
    with stream_bits(0):
        p0, s0, t0 = rx._dev_findall(batch)
    assert lib.mrx_last_kernel_name() == b"k_stream_findall"
    size0 = None
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [(torch.empty_like(p0), torch.empty_like(s0)) for _ in streams]
    with fused_findall(1):
        for it in range(40):
            k = it % 2
            with torch.cuda.stream(streams[k]):
                rx.findall_async(batch, outs[k])
                assert lib.mrx_last_kernel_name() == b"k_stream_findall_fused"
            if it == 9:
                torch.cuda.synchronize()
                size0 = lib.mrx_debug_scratch_bytes()
        torch.cuda.synchronize()
    assert lib.mrx_debug_scratch_bytes() == size0
    for pre, sp in outs:
        assert torch.equal(pre, p0) and torch.equal(sp[:t0], s0[:t0])


def test_scratch_arena_survives_failed_calls():
    """Calls that fail after they have taken scratch (a CSR batch whose offsets[n] is negative, a span buffer
    that is too small) give their scratch back: the arena of the (thread, stream) does not grow over the
    good calls that follow."""
    _need_gpu()
    import ctypes as C
    lib = M.load_library()
    rx = M.compile_regex(b"[a-z]+\\d+")
    texts = _random_texts(np.random.default_rng(11), 4000, 200, b"abcxyz0123456789 ")
    batch = M.DeviceBatch.from_texts(texts)
    p0, s0, t0 = rx._dev_findall(batch)
    for _ in range(3):
        rx._dev_findall(batch)
    torch.cuda.synchronize()
    size0 = lib.mrx_debug_scratch_bytes()
    bad_off = batch.offsets.clone()
    bad_off[-1] = -5
    prefix = torch.empty(batch.n + 1, dtype=torch.int64, device="cuda")
    spans = torch.empty((max(t0, 64), 2), dtype=torch.int32, device="cuda")
    total = C.c_int64(0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for it in range(100):
        if it % 5 == 0:
            rc = lib.mrx_findall_dev(rx._h, C.c_void_p(batch.data.data_ptr()), C.c_void_p(bad_off.data_ptr()), batch.n,
                                     C.c_void_p(prefix.data_ptr()), C.c_void_p(spans.data_ptr()), spans.shape[0], C.byref(total), stream)
            assert rc != 0, "negative offsets[n] must be refused"
            rc = lib.mrx_findall_dev(rx._h, C.c_void_p(batch.data.data_ptr()), C.c_void_p(batch.offsets.data_ptr()), batch.n,
                                     C.c_void_p(prefix.data_ptr()), C.c_void_p(spans.data_ptr()), 8, C.byref(total), stream)
            assert rc != 0 and total.value == t0, "a span buffer of 8 entries is too small: MRX_E_CAPACITY with the needed size"
        p1, s1, t1 = rx._dev_findall(batch)
        assert t1 == t0
    torch.cuda.synchronize()
    assert torch.equal(p1, p0) and torch.equal(s1[:t0], s0[:t0])
    assert lib.mrx_debug_scratch_bytes() == size0, (lib.mrx_debug_scratch_bytes(), size0)


AT_PATTERNS = [b"hello", b"[a-z]+\\d+", b"\\d+", b"[0-9]*", b"[a-z]*[0-9]+", b"^abc", b"^[a-z]+", b"a$", b"^abc$", b".*", b"",
               b"(x|y|foo|bar)+", b"(\\d{3})(\\d{3})(\\d{4})", b"hello world this is long", b"\\w+@\\w+\\.com", b"\\d{3}-\\d{4}",
               b"(foo|foobar)x", b"\\d+(\\.\\d+)?", b"[a-c]+[x-z]?", b"^[a-z]+[0-9]+$", b"^\\d+$", b"(a|b)*c", b"^(a|b)*c",
               b"abab", b"[^0-9]+", b"x*", b"hello.*", b".*@b\\.com", b"^aaaa.*a$",
               # match_first on the backtracking matcher: a start behind the end is the program's to answer
               b"xy|[^0-9]{0,2}$", b"ab|x?", b"(a|b)?$", b"a*(b|c)*", b"ab|[a-z]*$"]


@pytest.mark.parametrize("pat", AT_PATTERNS)
def test_start_argument_matches_oracle(pat):
    """mrx_{match_first,search,is_match}_at_*: Engine.match_first(text, start) and friends
    (engine.mojo:4-37, matcher.mojo:1049-1115) for every start in [0, len + 2] and start = -1, one
    start for the whole batch and one per text, CSR and fixed-pitch batches, against the oracle's
    CompiledRegex.match_first / match_next / is_match with the same start."""
    _need_gpu()
    rx = M.compile_regex(pat)
    orx = O.compile_regex(pat)
    rng = np.random.default_rng(zlib.crc32(pat))
    al = b"abcxyz0189 -.@fobarhelo w" + bytes(c for c in pat if chr(c).isalnum()) * 2
    texts = _random_texts(rng, 96, 40, al)
    texts += [b"hello world this is long", b"abc", b"abcabc", b"123-4567 555-1234", b"foobarx foox", b"a@b.com xx@yy.com",
              b"6502530000 4155551234", b"ab12", b"", b"x", b"3.14 15"]
    n = len(texts)
    pitch = 48
    arr = np.zeros((n, pitch), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    for i, t in enumerate(texts):
        arr[i, : len(t)] = np.frombuffer(t, dtype=np.uint8)
        lens[i] = len(t)
    csr = M.DeviceBatch.from_texts(texts)
    strided = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch, lens=torch.from_numpy(lens).cuda())

    def want(op, t, s):
        try:
            if op == "is_match":
                return bool(orx.is_match(t, s)) if s >= 0 else False
            r = (orx.match_first if op == "match_first" else orx.match_next)(t, s) if s >= 0 else None
            return r if r else (-1, -1)
        except UnsupportedByOracle:
            return "unsupported"

    per_text = rng.integers(-1, 44, size=n).astype(np.int32)
    per_text[:8] = [0, 1, 2, 0, 40, 41, -1, 3]
    cases = [("scalar", s) for s in (0, 1, 2, 3, 5, 11, 24, 25, 39, 40, 41, 42, -1)] + [("per_text", per_text)]
    for op in ("match_first", "search", "is_match"):
        for kind, start in cases:
            for batch in (csr, strided):
                try:
                    got = rx._at(op, batch, start)
                except M.UnsupportedPattern as exc:
                    # (operations the reference runs on its backtracking matcher with absolute positions)
                    if "per-text transition cache is tracked for at most 64" in str(exc):
                        continue
                    assert want(op, texts[0], 0) == "unsupported" or "start != 0 on an operation" in str(exc), (pat, op)
                    if "start != 0 on an operation" in str(exc):
                        assert kind == "per_text" or start != 0
                    continue
                if op == "is_match":
                    got = got.cpu().numpy()
                else:
                    got = np.stack([got[0].cpu().numpy(), got[1].cpu().numpy()], axis=1)
                for i, t in enumerate(texts):
                    s = int(start[i]) if kind == "per_text" else start
                    w = want(op, t, s)
                    assert w != "unsupported", (pat, op)
                    h = bool(got[i]) if op == "is_match" else (int(got[i][0]), int(got[i][1]))
                    assert h == w, (pat, op, kind, s, t, h, w)


GROUP_PATTERNS = [
    (b"(\\w+) (\\w+)", b"\\2 \\1"), (b"(\\d+)", b"[\\1,\\1]"), (b"(?:hello) (\\w+)", b"\\1"), (b"(\\d+)", b"[\\1]"),
    (b"(\\w+)@(\\w+)\\.com", b"\\2 at \\1"), (b"([a-z]+)(\\d*)x", b"<\\1|\\2>"), (b"^(\\w+)\\s+(\\w+)$", b"\\2,\\1"),
    (b"(a*)(b+)c", b"\\2\\1"), (b"\\s*(\\d+)\\s*", b"(\\1)"), (b"([a-z0-9]+)-([^-]+)", b"\\2-\\1"),
    (b"(\\d{2,4})-(\\d+)", b"\\2/\\1"), (b"x(.*)y", b"[\\1]"), (b"([a-zA-Z0-9._%+-]+)@([a-zA-Z0-9.-]+)", b"\\1 AT \\2"),
    (b"(\\w+)\\s(\\s*)(\\w*)", b"\\3\\2\\1"), (b"((\\w+)-(\\d+))", b"\\3:\\2:\\1"), (b"(\\d+)\\.(\\d+)", b"\\2.\\1 \\9"),
    (b"(h.llo) (w.*d)", b"\\2 \\1"), (b"([A-Z][a-z]+) ([A-Z][a-z]+)", b"\\2, \\1"), (b"(\\s+)(\\S?)", b"_\\2"),
    # alternation (_match_or) and quantified groups (_match_group_with_quantifier / the zero-repetition rule for a
    # quantified group that is not the last child of its sequence)
    (b"(a|b)(c)", b"\\2\\1"), (b"(ab)+(c)", b"<\\1\\2>"), (b"(\\w+)|(\\d+)", b"[\\1|\\2]"), (b"((a)|b)x", b"\\2\\1"),
    (b"(cat|dog)s? (\\w+)", b"\\2 \\1"), (b"x(a|b)?bc", b"[\\1]"), (b"(\\d+)(ab)*", b"\\1"), (b"(ab|a)(bc|c)?", b"\\1-\\2"),
    (b"(a(b|c)d)+", b"\\2\\1"), (b"((\\w)(\\d))*", b"\\3\\2"), (b"(foo|bar|baz)=(\\d+|x)", b"\\2=\\1"), (b"(a|ab)(c|bcd)(d*)", b"\\3\\2\\1"),
    (b"([a-c]+|\\d)x(y|z)*", b"\\1\\2"), (b"(a+|b+)+", b"<\\1>"), (b"(?:(x)|(y)|(z))+", b"\\1\\2\\3"),
]


@pytest.mark.parametrize("pat,repl", GROUP_PATTERNS)
def test_general_capture_groups_match_the_backtracking_oracle(pat, repl):
    """a18: regex.sub with \\1..\\9 on patterns outside the fixed-width group form, and the group spans
    themselves (mrx_captures_*): NFAEngine.match_next_with_groups (nfa.mojo:500-574) -- the recursive
    backtracking matcher, run on the GPU as a flat program -- against the oracle's restatement of
    nfa.mojo:657-1731 (oracle/mrx_ref/backtrack.py)."""
    _need_gpu()
    rx = M.compile_regex(pat)
    assert "device.backtrack=yes" in rx.describe(), rx.describe()
    orx = O.compile_regex(pat)
    assert orx.fixed_total_width < 0
    rng = np.random.default_rng(zlib.crc32(pat))
    al = b"abcxyz0189 -.@helowrdHW_\t,+" + bytes(c for c in pat if chr(c).isalnum()) * 2
    texts = _random_texts(rng, 160, 60, al) + _random_texts(rng, 20, 300, al)
    texts += [b"hello world", b"42 and 99", b"abc 123 def 456", b"john@example.com mary@test.com", b"ab12x cdx", b"foo  bar",
              b"aabbc bc abc", b"  17  ", b"k9-zz-top", b"2024-12 19-1", b"xaayxbby", b"a.b@c.d e@f", b"3.14 2.718", b"",
              b"hello world said Hello World", b"Ada Lovelace and Alan Turing", b"z9", b"1", b"77"]
    for count in (0, 1):
        got = rx.sub(repl, texts, count)
        for i, t in enumerate(texts):
            try:
                want = orx.sub(repl, t, count)
            except O.ReferenceDoesNotTerminate:   # the prefilter handed back a match in front of pos: nothing to agree with
                continue
            assert got[i] == want, (pat, repl, count, t, got[i])
    caps = rx.captures(texts)
    g = rx.num_groups
    bt = orx.matcher.nfa_matcher.backtrack
    for i, t in enumerate(texts):
        m, groups = bt.match_next_with_groups(t, 0)
        want = [(-1, -1)] * (g + 1)
        if m is not None:
            for gid, gs, ge in groups:
                if 1 <= gid <= g:
                    want[gid - 1] = (gs, ge)      # the last entry of a group wins (matcher.mojo:1797-1802)
            want[g] = m
        assert [tuple(int(x) for x in r) for r in caps[i]] == want, (pat, t)


@pytest.mark.parametrize("pat", [b"(" * 17 + b"a" + b")" * 17 + b"(b)", b"".join(bytes([c]) + b"*" for c in b"abcdefghijklmnopqrstuvwxyzABCDEFG") + b"(z)"])
def test_capture_groups_outside_the_flat_form_are_refused(pat):
    """What the flat program does not hold (groups nested deeper than 16, more than 30 open choices) is
    refused, never guessed."""
    _need_gpu()
    rx = M.compile_regex(pat)
    assert "device.backtrack=no" in rx.describe()
    with pytest.raises(M.UnsupportedPattern):
        rx.sub(b"\\1", [b"abc"])


BACKTRACKER_ROUTED = [b"hello.*", b".*@example\\.com", b".*world", b"^aaaa.*a$", b"a.*b$", b"hello.*world", b"\\w+@example\\.com$",
                      b".*\\d+", b"x.*", b"^\\s*hello.*", b"[a-z]+ing\\b" if False else b"[a-z]+ing.*",
                      # with alternation / quantified groups (reference vectors '^na|nb$', '.*(com|it)', '^x(a|b)?bc$')
                      b"^na|nb$", b".*(com|it)", b"^x(a|b)?bc$", b"^x(a)?ac$", b"^(a|b)*a.*a$", b"hello(a|b)*world", b"(ab)+c.*",
                      b".*(ing|ed)$", b"^(foo|bar).*x$"]


@pytest.mark.parametrize("pat", BACKTRACKER_ROUTED)
def test_backtracker_routed_operations_match_oracle(pat):
    """match_first / search / findall / is_match / sub of NFA-routed patterns that NFAMatcher hands to
    NFAEngine -- literal-prefiltered searches, leading / trailing '.*' fast paths, '$' patterns that are
    not one-pass (matcher.mojo:361-431, nfa.mojo:169-498) -- served by the backtracking matcher's flat
    program on the generic kernels, against the oracle's restatement (backtrack.py)."""
    _need_gpu()
    rx = M.compile_regex(pat)
    d = rx.describe()
    orx = O.compile_regex(pat)
    if orx.matcher.use_dfa:
        pytest.skip("DFA-routed after all")
    rng = np.random.default_rng(zlib.crc32(pat))
    al = b"abxhelowrd @.cmpying019\n " + bytes(c for c in pat if chr(c).isalnum()) * 2
    texts = _random_texts(rng, 200, 50, al) + _random_texts(rng, 20, 400, al)
    texts += [b"hello world", b"say hello there world", b"bob@example.com, eve@example.com", b"aaaaa", b"aaaaba", b"a b\nab",
              b"hello\nworld hello world", b"", b"x", b"singing and dancing", b"  hello you", b"line1\nuser@example.com"]
    # the batch-wide literal pass (bt_prepass / k_litscan) in front of the lanes: occurrences at chunk borders,
    # at the very end, only behind a newline, long texts, the literal alone
    # (kept to a few hundred bytes: a pattern without a literal backtracks through '.*' from every start)
    texts += [b"q" * 127 + b"hello", b"q" * 120 + b"hello world", b"z" * 300 + b"@example.com", b"p" * 400 + b"\n" + b"hello w" + b"r" * 200,
              b"hello", b"world", b"@example.com", b"k" * 500 + b"hello" + b"m" * 300 + b"world", b"w" * 255 + b"x", b"ing",
              b"sing" * 70, b"hell" * 40 + b"o", b"\n" * 10 + b"hello.*", b"aaaa" + b"b" * 200 + b"a"]
    if "literal_opt=0" in d and pat.startswith(b".*"):
        # no literal to prefilter with: every start backtracks through '.*' (quadratic upstream, in the oracle's
        # Python and on one GPU lane alike) -- short texts only
        texts = [t for t in texts if len(t) <= 100]
    supported_search = "support.search=yes" in d
    supported_first = "support.match_first=yes" in d
    assert supported_search or supported_first, d
    if supported_first:
        fs, fe = rx.match_first(texts)
        im = rx.is_match(texts)
    if supported_search:
        ss, se = rx.match_next(texts)
        lists = rx.findall_lists(texts)
        subs = rx.sub(b"<>", texts)
    for i, t in enumerate(texts):
        if supported_first:
            w = O.match_first(pat, t)
            assert (int(fs[i]), int(fe[i])) == (w if w else (-1, -1)), (pat, "match_first", t)
            assert bool(im[i]) == bool(orx.is_match(t, 0)), (pat, "is_match", t)
        if supported_search:
            w = O.search(pat, t)
            assert (int(ss[i]), int(se[i])) == (w if w else (-1, -1)), (pat, "search", t)
            assert lists[i] == O.findall(pat, t), (pat, "findall", t)
            try:
                assert subs[i] == O.sub(pat, b"<>", t), (pat, "sub", t)
            except O.ReferenceDoesNotTerminate:
                pass
    if supported_search:   # and without the batch-wide literal pass (every lane looks for itself)
        batch = M.DeviceBatch.from_texts(texts)
        cnt = rx.count(batch).cpu().numpy()
        with generic_kernels():
            ss2, se2 = rx.match_next(texts)
            assert rx.findall_lists(texts) == lists and rx.sub(b"<>", texts) == subs
            assert (rx.count(batch).cpu().numpy() == cnt).all()
        assert (np.asarray(ss2) == np.asarray(ss)).all() and (np.asarray(se2) == np.asarray(se)).all()
        assert [len(x) for x in lists] == [int(c) for c in cnt]
        for mode in (0, 1):   # the literal pass one lane per text / in 208-byte pieces (by default: 2 KiB pieces here)
            with litscan_pieces(mode):
                ss3, se3 = rx.match_next(texts)
                assert rx.findall_lists(texts) == lists and rx.sub(b"<>", texts) == subs, (pat, mode)
                assert (rx.count(batch).cpu().numpy() == cnt).all()
            assert (np.asarray(ss3) == np.asarray(ss)).all() and (np.asarray(se3) == np.asarray(se)).all(), (pat, mode)


@pytest.mark.parametrize("pat", [b"hello.*", b".*@example\\.com", b"hello.*world", b".*world"])
def test_literal_pass_in_pieces_on_long_texts(pat):
    """Few long texts: the backtracking matcher's literal pass (first / last occurrence, newline flag) cut into
    2 KiB pieces whose answers meet per text -- occurrences across the cuts, only in the first or last piece, none --
    against one lane per text and the oracle."""
    _need_gpu()
    rx = M.compile_regex(pat)
    rng = np.random.default_rng(zlib.crc32(pat) + 5)
    texts = []
    for i in range(40):
        L = int(rng.integers(3000, 30000))
        t = bytearray(rng.choice(np.frombuffer(b"abcdefg xyz.@", dtype=np.uint8), size=L).tobytes())
        if i % 4 != 3:
            for word in (b"hello", b"world", b"@example.com"):
                for _ in range(int(rng.integers(0, 3))):
                    at = int(rng.integers(0, L - 16))
                    if i % 4 == 1:
                        at = min(L - 16, (at // 2048) * 2048 + 2048 - int(rng.integers(0, len(word) + 1)))   # on a cut
                    t[at:at + len(word)] = word
        if i % 5 == 0:
            t[int(rng.integers(0, L))] = 10
        texts.append(bytes(t))
    texts += [b"hello world" + b"q" * 5000, b"q" * 5000 + b"hello world", b"q" * 2044 + b"hello" + b"q" * 2043 + b"world",
              b"u@example.com" + b"\n" * 4000, b"q" * 6000]
    ss, se = rx.match_next(texts)
    lists = rx.findall_lists(texts)
    with litscan_pieces(0):
        ss0, se0 = rx.match_next(texts)
        assert rx.findall_lists(texts) == lists
    assert (np.asarray(ss0) == np.asarray(ss)).all() and (np.asarray(se0) == np.asarray(se)).all()
    for i, t in enumerate(texts):
        w = O.search(pat, t)
        assert (int(ss[i]), int(se[i])) == (w if w else (-1, -1)), (pat, i)
        assert lists[i] == O.findall(pat, t), (pat, i)


@pytest.mark.parametrize("pat", [b"[a-z]+\\d+", b"(\\d{3})(\\d{3})(\\d{4})", b"hello"])
def test_split_findall_equals_one_batch(pat):
    """mrx_debug_split_findall(1): a fixed-pitch batch of 2^18 texts and more as two halves on two streams (decode of
    the first half under the scan of the second; off by default, it measured slower) -- same CSR as the one-batch form."""
    _need_gpu()
    lib = M.load_library()
    n, L = (1 << 18) + 77, 64
    g = torch.Generator(device="cuda")
    g.manual_seed(zlib.crc32(pat))
    al = torch.tensor(list(b"abcxyz0123456789 helo-"), dtype=torch.uint8, device="cuda")
    data = al[torch.randint(0, al.numel(), (n, L), generator=g, device="cuda")]
    lens = torch.randint(0, L + 1, (n,), generator=g, device="cuda", dtype=torch.int32)
    rx = M.compile_regex(pat)
    for batch in (M.DeviceBatch.strided(data.reshape(-1), L, length=L), M.DeviceBatch.strided(data.reshape(-1), L, lens=lens)):
        pre0, sp0, tot0 = rx._dev_findall(batch)
        lib.mrx_debug_split_findall(1)
        try:
            pre1, sp1, tot1 = rx._dev_findall(batch)
            pre2, sp2, tot2 = rx._dev_findall(batch)   # back to back: the side stream and its events are reused
        finally:
            lib.mrx_debug_split_findall(0)
        assert tot0 == tot1 == tot2 and tot0 > 0
        assert torch.equal(pre0, pre1) and torch.equal(sp0[:tot0], sp1[:tot0]) and torch.equal(sp0[:tot0], sp2[:tot0])


@pytest.mark.parametrize("pat", [b"\\d+(\\.\\d+)?", b"[A-Z]{2,4}[0-9]{3,5}", b"\\d{3}-\\d{3}-\\d{4}", b"[0-9]+\\.[0-9]+", b"[a-z]+@[a-z]+",
                                 b"(foo|foobar)x", b"[0-9]+:[0-9]+", b"\\(?\\d{3}\\)?[\\s.-]?\\d{3}[\\s.-]?\\d{4}"])
def test_stepper_plans_in_disjoint_pieces(pat):
    """findall / count of plans that do not stream, long texts cut at bytes on which every walk dies and none begins
    (pieces of 200 bytes under mrx_debug_long_text_kernels(1)): against one lane per whole text and the oracle --
    matches next to the cuts, texts without any synchronising byte for longer than the look-back, the last piece of
    the batch, CSR and fixed pitch."""
    _need_gpu()
    rx = M.compile_regex(pat)
    d = rx.describe()
    if "device.streamable=yes" in d or "sync_bytes=0 " in d:
        pytest.skip("streams, or has no synchronising byte")
    lib = M.load_library()
    rng = np.random.default_rng(zlib.crc32(pat) + 9)
    al = b"abcfoxAZ0123456789.-@: ()" + bytes(c for c in pat if chr(c).isalnum())
    texts = _random_texts(rng, 60, 2500, al) + _random_texts(rng, 40, 150, al) + [
        b"", b"12", b"ABC1234", b"7" * 1000, b"A" * 700 + b"1234", b"555-123-4567 " * 90, b"3.14 " * 300 + b"2.", b"x" * 199 + b"12:30" * 50,
        b"a@b " * 260, b"foobarx" * 120, b"12345.6789" * 77 + b" 1.5"]
    batch = M.DeviceBatch.from_texts(texts)
    with long_text_kernels(2):
        want = rx.findall_lists(texts)
        wcnt = rx.count(batch).cpu().numpy()
    with long_text_kernels(1):
        got = rx.findall_lists(texts)
        used = lib.mrx_last_kernel_name()
        cnt = rx.count(batch).cpu().numpy()
    assert used.endswith(b"_pieces"), used
    assert (cnt == wcnt).all()
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, (pat, i, len(texts[i]), g[:4], w[:4])
    for i in list(range(0, len(texts), 9)) + list(range(len(texts) - 11, len(texts))):
        assert got[i] == O.findall(pat, texts[i]), (pat, i)
    # fixed pitch, default thresholds: texts of 12 KiB of a plain-route plan take the pieces on their own
    L = 12288
    rows = [(t * (L // max(len(t), 1) + 1))[:L] for t in texts[:24] if len(t) >= 8]
    sb = M.DeviceBatch.strided(torch.tensor(list(b"".join(rows)), dtype=torch.uint8, device="cuda"), L, length=L)
    pre, sp, tot = rx._dev_findall(sb)
    if "required-byte route" not in d:
        assert lib.mrx_last_kernel_name().endswith(b"_pieces"), lib.mrx_last_kernel_name()
    pre_h, sp_h = pre.cpu().numpy(), sp.cpu().numpy()
    for i in range(0, len(rows), 5):
        assert [tuple(int(x) for x in r) for r in sp_h[pre_h[i]:pre_h[i + 1]]] == O.findall(pat, rows[i]), (pat, i)


@contextlib.contextmanager
def subs_group(lanes):
    """Lanes per text in k_subs_wave (16 / 32 / 64), 0 = k_subs_emit for every text, -1 = by average length."""
    lib = M.load_library()
    lib.mrx_debug_subs_group(lanes)
    try:
        yield
    finally:
        lib.mrx_debug_subs_group(-1)


@contextlib.contextmanager
def litscan_pieces(mode):
    """The backtracking matcher's literal pass: 0 = one lane per text, 1 = always in 208-byte pieces."""
    lib = M.load_library()
    lib.mrx_debug_litscan_pieces(mode)
    try:
        yield
    finally:
        lib.mrx_debug_litscan_pieces(2)


@contextlib.contextmanager
def dynamic_texts(mode):
    """1 = ragged batches always on k_stream_dyn (texts handed to lanes as they fall free), 2 = never."""
    lib = M.load_library()
    lib.mrx_debug_dynamic_texts(mode)
    try:
        yield
    finally:
        lib.mrx_debug_dynamic_texts(0)


@pytest.mark.parametrize("pat", [b"[a-z]+\\d+", b"\\d+", b"hello", b"(\\d{3})(\\d{3})(\\d{4})", b"(x|y|foo|bar)+", b"a", b"abab",
                                 b"[a-c]+[0-9]+[x-z]+[0-9]+", b"[a-z]+", b"(cat|dog)+"])
@pytest.mark.parametrize("lens_kind", ["uniform_ragged", "short", "with_long", "chunk_multiples"])
def test_dynamic_text_assignment_equals_static(pat, lens_kind):
    """k_stream_dyn (a lane takes the wavefront's next text when its own ends; records carry the text's
    index in a 256-text task) against k_stream_findall's one-text-per-lane form on the same ragged CSR
    batches: findall (offsets and spans), count and search identical; oracle on a sample.  Lengths cover
    empty texts, texts that end exactly on a 128-byte chunk boundary, texts of several chunks among
    short ones, a partial last task and fewer than 64 texts."""
    _need_gpu()
    lib = M.load_library()
    rx = M.compile_regex(pat)
    d = rx.describe()
    assert "device.streamable=yes" in d
    if "reset_byte=-1" in d:
        pytest.skip("no reset byte: the plan keeps the static form")
    rng = np.random.default_rng(zlib.crc32(pat) + len(lens_kind))
    al = np.frombuffer(b"abcxyz0189 -fobarhelcatdg5" + bytes(c for c in pat if chr(c).isalnum()) * 2, dtype=np.uint8)
    for n in (40, 256 * 3 + 17, 5000):
        if lens_kind == "uniform_ragged":
            lens = rng.integers(0, 600, size=n)
        elif lens_kind == "short":
            lens = rng.integers(0, 40, size=n)
        elif lens_kind == "with_long":
            lens = rng.integers(0, 200, size=n)
            lens[rng.integers(0, n, size=max(1, n // 50))] = rng.integers(1500, 6000, size=max(1, n // 50))
        else:
            lens = rng.choice([0, 16, 112, 128, 129, 256, 384, 127, 1], size=n)
        lens[:: 9] = 0
        texts = []
        for L in lens:
            t = rng.choice(al, size=int(L)).astype(np.uint8)
            if L > 3 and rng.random() < 0.3:
                t[-3:] = np.frombuffer(b"a12"[: 3], dtype=np.uint8)    # a match that runs to the end of the text
            texts.append(t.tobytes())
        batch = M.DeviceBatch.from_texts(texts)
        with long_text_kernels(2):
            with dynamic_texts(1):
                p1, s1, t1 = rx._dev_findall(batch)
                assert lib.mrx_last_kernel_name() == b"k_stream_findall_dyn"
                c1 = rx.count(batch)
                assert lib.mrx_last_kernel_name() == b"k_stream_count_dyn"
                a1, b1 = rx.match_next(batch)
                assert lib.mrx_last_kernel_name() == b"k_stream_search_dyn"
            with dynamic_texts(2):
                p2, s2, t2 = rx._dev_findall(batch)
                assert lib.mrx_last_kernel_name() == b"k_stream_findall"
                c2 = rx.count(batch)
                a2, b2 = rx.match_next(batch)
        assert t1 == t2 and torch.equal(p1, p2) and torch.equal(s1[:t1], s2[:t2]), (pat, lens_kind, n)
        assert torch.equal(c1, c2) and torch.equal(a1, a2) and torch.equal(b1, b2), (pat, lens_kind, n)
        pre, sp = p1.cpu().numpy(), s1.cpu().numpy()
        for i in range(0, n, max(1, n // 40)):
            have = [tuple(int(x) for x in r) for r in sp[pre[i]:pre[i + 1]]]
            assert have == O.findall(pat, texts[i]), (pat, lens_kind, n, i)


@pytest.mark.parametrize("pat,repl", [(b"[a-z]+\\d+", b"#"), (b"(\\d{3})(\\d{3})(\\d{4})", b"\\1-\\2-\\3"), (b"\\d+(\\.\\d+)?", b"N"),
                                      (b"(\\w+) (\\w+)", b"\\2 \\1")])
def test_sub_on_fixed_pitch_batches(pat, repl):
    """mrx_sub_strided_dev (matcher.mojo:1857-1917 over a fixed-pitch batch): rows without padding take the CSR
    fast paths through offsets written on the device, padded rows the lane-per-text kernels; both equal mrx_sub_dev
    on the same texts and the oracle."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat))
    al = np.frombuffer(b"abcxy 0123456789.-", dtype=np.uint8)
    rx = M.compile_regex(pat)
    for n, pitch, var in ((200, 64, False), (100, 96, True), (70, 50, True), (33, 1024, False)):
        arr = rng.choice(al, size=(n, pitch)).astype(np.uint8)
        lens = rng.integers(0, pitch + 1, size=n).astype(np.int32) if var else None
        texts = [arr[i, : (lens[i] if var else pitch)].tobytes() for i in range(n)]
        d = torch.from_numpy(arr).cuda().reshape(-1)
        sb = M.DeviceBatch.strided(d, pitch, length=pitch, lens=torch.from_numpy(lens).cuda() if var else None)
        for count in (0, 2):
            so, sd = rx.sub_dev(repl, sb, count)
            co, cd = rx.sub_dev(repl, M.DeviceBatch.from_texts(texts), count)
            assert torch.equal(so, co) and torch.equal(sd, cd), (pat, n, pitch, var, count)
            so_h, sd_h = so.cpu().numpy(), sd.cpu().numpy().tobytes()
            for i in range(0, n, 7):
                assert sd_h[so_h[i]:so_h[i + 1]] == O.sub(pat, repl, texts[i], count), (pat, texts[i])


@pytest.mark.parametrize("pat", [b"(x|y|foo|bar)+", b"[a-z]+\\d+", b"\\d+"])
def test_dense_matches_decode_in_row_windows(pat):
    """k_decode's dense path (a wavefront's 64 texts hold more spans than three LDS tiles): row windows per text,
    one coalesced store per row and batch, spans behind a row's window stored directly -- a match every one to three
    bytes, texts of very different density in one wavefront (one text far denser than the batch's average), fixed
    pitch and CSR, against the generic kernels on every text and the oracle on a sample."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat) + 3)
    dense = {b"(x|y|foo|bar)+": b"xyxyfoobarxy ", b"[a-z]+\\d+": b"a1b2c3 d4", b"\\d+": b"1 2 3 45 6 "}[pat]
    n, L = 192, 4096
    arr = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        al = np.frombuffer(dense if i % 3 else dense + b" " * 40, dtype=np.uint8)   # every third text is sparse
        arr[i] = rng.choice(al, size=L)
    arr[5] = np.frombuffer((b"x1" * (L // 2))[:L], dtype=np.uint8) if pat != b"\\d+" else np.frombuffer((b"1 " * (L // 2))[:L], dtype=np.uint8)
    rx = M.compile_regex(pat)
    d = torch.from_numpy(arr).cuda().reshape(-1)
    lens = rng.integers(L // 2, L + 1, size=n).astype(np.int32)
    batches = [M.DeviceBatch.strided(d, L, length=L),
               M.DeviceBatch.strided(d, L, length=L, lens=torch.from_numpy(lens).cuda()),
               M.DeviceBatch.from_texts([arr[i, : lens[i]].tobytes() for i in range(n)])]
    for bi, b in enumerate(batches):
        pre, sp, tot = rx._dev_findall(b)
        with generic_kernels():
            gpre, gsp, gtot = rx._dev_findall(b)
        assert tot == gtot and tot > 3 * 3072 * (n // 64), (pat, bi, tot)
        assert torch.equal(pre, gpre) and torch.equal(sp[:tot], gsp[:tot]), (pat, bi)
    pre_h, sp_h = pre.cpu().numpy(), sp.cpu().numpy()
    for i in (0, 3, 5, 64, 191):
        t = arr[i, : lens[i]].tobytes()
        assert [tuple(int(x) for x in r) for r in sp_h[pre_h[i]:pre_h[i + 1]]] == O.findall(pat, t), (pat, i)


@pytest.mark.parametrize("pat", [b"[, ]+", b"\\d+", b"ab", b"[a-z]+\\d+", b"x*", b"(\\d{3})(\\d{3})(\\d{4})", b"hello.*world"])
@pytest.mark.parametrize("maxsplit", [0, 1, 3, -1])
def test_split_behind_the_c_abi_equals_the_oracle(pat, maxsplit):
    """regex.split (matcher.mojo:1357-1393) through mrx_split_batch / mrx_split_dev / mrx_split_strided_dev: pieces as
    byte ranges per text, against the oracle's split text by text -- separators at the very start and end, adjacent
    separators (empty pieces), texts without a separator, empty texts, every maxsplit rule (0 no limit, n at most n
    splits, negative: none)."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat) + maxsplit + 7)
    al = b"ab12, xhelloworld" + bytes(c for c in pat if chr(c).isalnum())
    texts = _random_texts(rng, 400, 90, al) + [b"", b",", b",,a,,", b"a,b", b"12ab34", b"hello big world, hello world"]
    rx = M.compile_regex(pat)
    try:
        want = [O.split(pat, t, maxsplit) for t in texts]
    except UnsupportedByOracle:
        pytest.skip("oracle does not cover this pattern")
    assert rx.split(texts, maxsplit) == want
    batch = M.DeviceBatch.from_texts(texts)
    prefix, pieces, total = rx.split_dev(batch, maxsplit)
    pre, pc = prefix.cpu().numpy(), pieces.cpu().numpy()
    assert total == int(pre[-1]) == sum(len(w) for w in want)
    for i, t in enumerate(texts):
        assert [t[int(a):int(b)] for a, b in pc[pre[i]:pre[i + 1]]] == want[i], (pat, maxsplit, i, t)
    # fixed pitch with per-text lengths
    pitch = 96
    arr = np.zeros((len(texts), pitch), dtype=np.uint8)
    lens = np.array([len(t) for t in texts], dtype=np.int32)
    for i, t in enumerate(texts):
        arr[i, : len(t)] = np.frombuffer(t, dtype=np.uint8)
    sb = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), pitch, lens=torch.from_numpy(lens).cuda())
    p2, c2, t2 = rx.split_dev(sb, maxsplit)
    assert t2 == total and torch.equal(p2, prefix) and torch.equal(c2[:t2], pieces[:total])
    # a piece buffer that is too small is reported, with the need when no limit is on
    if total > 4:
        with pytest.raises(M.MrxError):
            rx.split_dev(batch, maxsplit, piece_cap=total - 1)


@pytest.mark.parametrize("pat", [b"[a-z]+[0-9]+$", b"(foo|[0-9]+)$", b"[a-c]+[0-9]*$", b"(a|b)+c$", b"x*y$|z[0-9]", b"(ab|a)c*$",
                                 b"[a-z]+[0-9]+x?$", b"(\\d+|[a-f]+)$", b"[a-z]+$"])
def test_lazy_end_stepper_ends_a_walk_where_it_meets_the_last_failed_one(pat):
    """Round 4: the LZ stepper walks the last failed walk again beside the lane and ends the current walk the moment it
    stands in that walk's state at that walk's position (same future: no accepting state any more) -- upstream's
    restart-per-position loop (pikevm.mojo:754-867) without its quadratic cost on long failing walks.  Same answers
    as the literal restatement (generic kernels: every restart walks to its death, walk_lazy_end) on texts made of
    long runs, with the text's last byte value also inside the text / only at its end / in every state's reach, and as
    the oracle on a sample."""
    _need_gpu()
    from mrx_ref import hybrid as H
    lib = M.load_library()
    rx = M.compile_regex(pat)
    d = rx.describe()
    if "device.lazy_end_cache=yes" not in d:
        pytest.skip("not a '$' program on the LazyDFA search")
    rng = np.random.default_rng(zlib.crc32(pat) + 4)
    al = b"abcf019xyz " + bytes(c for c in pat if chr(c).isalnum())
    texts = []
    for L in (40, 200, 1024):
        for _ in range(120):
            kind = rng.integers(0, 6)
            if kind == 0:     # one long run of one class, then a byte that kills / ends it
                t = bytes(rng.choice(np.frombuffer(b"abc", dtype=np.uint8), size=L - 1).tolist()) + bytes([rng.choice(list(b"!1c9 "))])
            elif kind == 1:   # letters then digits to the end (a match that runs to the end), last digit also inside
                k = int(rng.integers(1, L - 1))
                t = bytes(rng.choice(np.frombuffer(b"abc", dtype=np.uint8), size=k).tolist()) + bytes(rng.choice(np.frombuffer(b"019", dtype=np.uint8), size=L - k).tolist())
            elif kind == 2:   # tokens
                t = b" ".join(bytes(rng.choice(np.frombuffer(b"abcf", dtype=np.uint8), size=int(rng.integers(1, 9))).tolist()) +
                              bytes(rng.choice(np.frombuffer(b"019", dtype=np.uint8), size=int(rng.integers(0, 4))).tolist())
                              for _ in range(L // 6))[:L]
            elif kind == 3:   # runs of runs: abab...c, xxxy
                t = (bytes(rng.choice(np.frombuffer(b"ab", dtype=np.uint8), size=L // 2).tolist()) + b"c" +
                     b"x" * (L // 4) + b"y" + bytes(rng.choice(np.frombuffer(b"z0f", dtype=np.uint8), size=L // 8).tolist()))[:L]
            else:
                t = bytes(rng.choice(np.frombuffer(al, dtype=np.uint8), size=L).tolist())
            texts.append(t)
    texts += [b"", b"a", b"a1", b"aa1a1", b"abcabc", b"foofoo", b"xxyxxy", b"a" * 700 + b"1", b"a" * 700 + b"!", b"ab" * 300 + b"c"]
    batch = M.DeviceBatch.from_texts(texts)

    def run():
        prefix, spans, total = rx._dev_findall(batch)
        k1 = lib.mrx_last_kernel_name()
        cnt = rx.count(batch)
        s, e = rx.match_next(batch)
        return (prefix.cpu().numpy(), spans[:total].cpu().numpy(), cnt.cpu().numpy(), s.cpu().numpy(), e.cpu().numpy()), k1

    got, k1 = run()
    with generic_kernels():
        want, k2 = run()
    assert k2 in (b"k_findall_count", b"k_findall"), k2
    for a, b in zip(got, want):
        assert np.array_equal(a, b), (pat, k1)
    o = H.CompiledRegex(pat)
    for i, t in enumerate(texts):
        if len(t) > 220 and i % 9:
            continue   # (the Python oracle is quadratic on these too)
        have = [tuple(int(x) for x in r) for r in got[1][got[0][i]:got[0][i + 1]]]
        assert have == o.match_all(t), (pat, i, t[:60])
        w = o.match_next(t, 0)
        assert (int(got[3][i]), int(got[4][i])) == (w if w else (-1, -1)), (pat, i)


@pytest.mark.parametrize("pat", [b"(x|y|foo|bar)+", b"[a-z]+\\d+", b"\\d+", b"hello", b"[0-9a-f]", b"(\\d{3})(\\d{3})(\\d{4})", b"a+b"])
@pytest.mark.parametrize("L", [2048, 2048 + 128, 4096, 8192 + 4096])
def test_findall_by_event_rows_equals_records_and_oracle(pat, L):
    """mrx_debug_dense_rows(1): the scan writes 8 bytes of event words per 32 text bytes at a fixed pitch and
    k_decode_rows (32 or 64 lanes per text) derives starts, indices and the match that ends with the text -- against
    the record form (mrx_debug_dense_rows(2)) span by span, and against the oracle on a sample.  Texts of exactly one round,
    a round and a bit, and three rounds; matches across lane, round and text ends; texts without a match; a match every
    byte; a batch that is not a multiple of 64 texts (nor of the texts a wavefront takes)."""
    _need_gpu()
    lib = M.load_library()
    rng = np.random.default_rng(zlib.crc32(pat) + L)
    n = 64 * 5 + 37
    kinds = [b"xyfoobar ", b"abcdefgh0123456789 ", b"0123456789abcdef", b"hello wrd", b"ab"]
    data = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        al = np.frombuffer(kinds[i % len(kinds)], dtype=np.uint8)
        data[i] = al[rng.integers(0, len(al), L)]
    data[3] = ord("x")                      # one match over the whole text: ends with it, no event
    data[4] = ord("7")
    data[5] = ord(" ")                      # no match at all
    data[6, :] = np.frombuffer((b"ab12 " * (L // 5 + 1))[:L], dtype=np.uint8)
    data[7, -3:] = np.frombuffer(b"a12", dtype=np.uint8)   # a match that ends with the text behind others
    data[8, :] = np.frombuffer((b"1234567890" * (L // 10 + 1))[:L], dtype=np.uint8)
    d = torch.from_numpy(data).cuda()
    rx = M.compile_regex(pat)
    if "device.streamable=yes" not in rx.describe():
        pytest.skip("does not stream")
    batch = M.DeviceBatch.strided(d.reshape(-1), L, length=L)
    lib.mrx_debug_dense_rows(2)
    lib.mrx_debug_long_text_kernels(2)   # (a few hundred texts of some KiB would go by pieces)
    try:
        pre0, sp0, tot0 = rx._dev_findall(batch)
        k0 = lib.mrx_last_kernel_name()
        lib.mrx_debug_dense_rows(1)
        pre1, sp1, tot1 = rx._dev_findall(batch)
        k1 = lib.mrx_last_kernel_name()
        # a span buffer that is too small: clipped, the total still reported
        small = torch.empty((max(tot0 // 2, 1), 2), dtype=torch.int32, device="cuda")
        prefix = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        rx.findall_async(batch, (prefix, small))
        torch.cuda.synchronize()
    finally:
        lib.mrx_debug_dense_rows(0)
        lib.mrx_debug_long_text_kernels(0)
    assert k1 == b"k_stream_findall_rows" and k0 != k1, (k0, k1)
    assert tot0 == tot1 and torch.equal(pre0, pre1) and torch.equal(sp0[:tot0], sp1[:tot0])
    assert torch.equal(prefix, pre0) and torch.equal(small, sp0[:small.shape[0]])
    pre, sp = pre1.cpu().numpy(), sp1.cpu().numpy()
    for i in list(range(0, 12)) + list(range(n - 5, n)):
        want = O.findall(pat, data[i].tobytes())
        got = [tuple(x) for x in sp[pre[i]:pre[i + 1]].tolist()]
        assert got == want, (pat, L, i, got[:3], want[:3])


def test_event_rows_are_chosen_after_a_dense_batch_and_dropped_after_a_sparse_one():
    """Default mode: the handle's previous eligible call decides (its total travels to pinned memory behind the call's work;
    nobody waits for it).  First call: records; after a batch with a match every few bytes: rows; after a batch with few
    matches: records again."""
    _need_gpu()
    lib = M.load_library()
    rx = M.CompiledRegex(b"(x|y|foo|bar)+")   # (a handle of its own: the cached one has seen other tests' batches)
    n, L = (1 << 17) + 64, 4096   # (more than 2^17 texts: fewer go by pieces)
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    al = torch.tensor(list(b"xyfoobar "), dtype=torch.uint8, device="cuda")
    dense = al[torch.randint(0, al.numel(), (n, L), generator=g, device="cuda")]
    sparse = torch.full((n, L), ord("."), dtype=torch.uint8, device="cuda")
    sparse[:, 100] = ord("x")
    bd = M.DeviceBatch.strided(dense.reshape(-1), L, length=L)
    bs = M.DeviceBatch.strided(sparse.reshape(-1), L, length=L)
    names = []
    want = None
    for batch in (bd, bd, bd, bs, bs, bd):
        pre, sp, tot = rx._dev_findall(batch, span_cap=n * 1024)   # (room for every span: a retry would be a call of its own)
        names.append(lib.mrx_last_kernel_name())
        torch.cuda.synchronize()
        if batch is bd:
            if want is None:
                want = (pre.clone(), sp[:tot].clone())
            assert torch.equal(pre, want[0]) and torch.equal(sp[:tot], want[1])
    assert names[0] != b"k_stream_findall_rows"
    assert names[2] == b"k_stream_findall_rows", names      # the first call's total has arrived by the third at the latest
    assert names[3] == b"k_stream_findall_rows"             # the sparse batch: the last answer still says dense
    assert names[5] != b"k_stream_findall_rows", names      # ... and its own total says otherwise


@pytest.mark.parametrize("pat,repl", [(b"[a-z]+\\d+", b"#"), (b"(\\d{3})(\\d{3})(\\d{4})", b"\\1-\\2-\\3"), (b"hello", b"<bye>"), (b"\\d+", b"")])
def test_sub_with_known_totals_equals_sub(pat, repl):
    """mrx_sub_known_dev (DeviceBatch.csr_known / from_texts): the same bytes as mrx_sub_dev, which reads offsets[n] and
    the longest text from the device first; upper bounds instead of the exact values change nothing either."""
    _need_gpu()
    rng = np.random.default_rng(zlib.crc32(pat) + 3)
    texts = _random_texts(rng, 700, 900, b"abchelo 0123456789-") + [b"", b"hello", b"5551234567", b"ab12" * 500, b"x" * 3000]
    data, offsets = M.api.pack_texts(texts)
    d = torch.from_numpy(data).cuda()
    o = torch.from_numpy(offsets).cuda()
    rx = M.compile_regex(pat)
    off0, out0 = rx.sub_dev(repl, M.DeviceBatch(d, o))
    for end, mx in ((int(offsets[-1]), 3000), (int(offsets[-1]) + 4096, 5000)):
        off1, out1 = rx.sub_dev(repl, M.DeviceBatch.csr_known(d, o, end, mx))
        assert torch.equal(off0, off1) and torch.equal(out0, out1)
    got = out0.cpu().numpy().tobytes()
    oo = off0.cpu().numpy()
    for i in (0, 1, 350, 699, 700, 701, 702, 703, 704):
        assert got[oo[i]:oo[i + 1]] == O.sub(pat, repl, texts[i]), (pat, i)


@pytest.mark.parametrize("seed", [20260701, 20260702])
def test_generated_patterns_event_rows_equal_records(seed):
    """Every generated pattern whose plan streams: findall by event rows (mrx_debug_dense_rows(1)) against the record form
    on texts of 2 KiB and 5 KiB built from the pattern's own alphabet -- long single-byte runs, texts that end in a
    match, a batch that is not a multiple of 64 texts; every automaton form the generator reaches (byte columns, code
    columns, class tables, pair tables)."""
    _need_gpu()
    from pattern_gen import patterns
    lib = M.load_library()
    rng = np.random.default_rng(seed)
    base = np.frombuffer(b"abcxyz019 -@.fobrhelcatdg", dtype=np.uint8)
    nstream = 0
    lib.mrx_debug_long_text_kernels(2)
    try:
        for p in patterns(seed, 300):
            pb = p.encode()
            try:
                rx = M.compile_regex(pb)
            except M.RegexSyntaxError:
                continue
            if "device.streamable=yes" not in rx.describe():
                continue
            nstream += 1
            lit = np.frombuffer(bytes(c for c in pb if chr(c).isalnum() or c in b" -@."), dtype=np.uint8)
            al = np.concatenate([base, lit, lit]) if lit.size else base
            for n, L in ((64 + 9, 2048), (70, 5 * 1024), (67, 2048 + 256)):
                arr = rng.choice(al, size=(n, L)).astype(np.uint8)
                for i in range(0, n, 4):
                    arr[i, : int(rng.integers(0, L))] = al[int(rng.integers(0, al.size))]
                batch = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), L, length=L)
                lib.mrx_debug_dense_rows(2)
                pre0, sp0, tot0 = rx._dev_findall(batch)
                lib.mrx_debug_dense_rows(1)
                pre1, sp1, tot1 = rx._dev_findall(batch)
                assert lib.mrx_last_kernel_name() == b"k_stream_findall_rows", p
                assert tot0 == tot1 and torch.equal(pre0, pre1) and torch.equal(sp0[:tot0], sp1[:tot0]), (p, n, L)
    finally:
        lib.mrx_debug_dense_rows(0)
        lib.mrx_debug_long_text_kernels(0)
    assert nstream > 40, nstream


def test_multiwalk_table_that_no_walk_ever_enters():
    """`[^0-9]id\\s+baz`: the first-byte filter (digits: A.6's "later transition overwrites" quirk) allows no byte the
    start row has a transition on, so the multi-walk table has one configuration and zero walks (mw_walks=0).  Found by
    the round-4 fuzz when 0 was briefly the marker of the empty-match walk: sub appended a replacement to every text."""
    _need_gpu()
    pat = b"[^0-9]id\\s+baz"
    rx = M.compile_regex(pat)
    assert "mw_walks=0" in rx.describe()
    rng = np.random.default_rng(5)
    texts = _random_texts(rng, 200, 60, b"abcidz 0129-baz.") + [b"", b"xid baz", b"0id  baz", b".id\tbaz 1id baz"]
    assert rx.findall_lists(texts) == [O.findall(pat, t) for t in texts]
    assert rx.sub(b"#", texts) == [O.sub(pat, b"#", t) for t in texts]
    batch = M.DeviceBatch.from_texts(texts)
    assert rx.count(batch).cpu().tolist() == [len(O.findall(pat, t)) for t in texts]


def test_pending_tries_walk_is_timed_against_marks_and_the_answers_stay_the_same():
    """PF_MW_TRIES plan on a batch of 256 texts and more: the handle takes the pending-tries walk and marks + stepper
    alternately on its first four calls (two of them timed) and keeps the faster route; every call returns the same CSR."""
    _need_gpu()
    lib = M.load_library()
    pat = b"foo|[a-z]{3}\\d|[ab]"
    rx = M.CompiledRegex(pat)   # (a handle of its own: the tuner's state is the handle's)
    assert "tries_walk=yes" in rx.describe()
    rng = np.random.default_rng(12)
    al = np.frombuffer(b"abcfoxyz0123 -", dtype=np.uint8)
    n, L = 8192, 208
    arr = al[rng.integers(0, len(al), (n, L))]
    batch = M.DeviceBatch.strided(torch.from_numpy(arr).cuda().reshape(-1), L, length=L)
    names, first = [], None
    for _ in range(7):
        pre, sp, tot = rx._dev_findall(batch, span_cap=n * L)   # (room for every span: a retry would be a call of its own)
        names.append(lib.mrx_last_kernel_name())
        torch.cuda.synchronize()
        if first is None:
            first = (pre.clone(), sp[:tot].clone())
        assert torch.equal(pre, first[0]) and torch.equal(sp[:tot], first[1])
    assert names[0] == b"k_mwalk" and names[1] == b"k_backscan+k_step_count", names     # the two candidates in turn
    assert names[5] == names[6] and names[6] in (b"k_mwalk", b"k_backscan+k_step_count"), names
    cnt = rx.count(batch).cpu()
    assert torch.equal(cnt.to(torch.int64), (first[0][1:] - first[0][:-1]).cpu())
    pre_h, sp_h = first[0].cpu().numpy(), first[1].cpu().numpy()
    for i in range(0, n, 997):
        assert [tuple(x) for x in sp_h[pre_h[i]:pre_h[i + 1]].tolist()] == O.findall(pat, arr[i].tobytes())
