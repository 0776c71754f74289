"""CPU: the product's C++ pattern compiler against the oracle's Python one.

Two independent implementations of the reference's front end, classifier,
router and table builders must produce the same routing decision and the same
tables (states, transitions, accept flags, matcher set, PikeVM program) for
every pattern the reference's own tests use plus a set of shape-covering extras.
Also checks that libmrx_hip.so loads and exports every symbol of include/mrx.h
(no compute calls: this runs without a GPU).
"""
import os
import re

import pytest

import mojo_regex_amd as M
from mojo_regex_amd import api
from mrx_ref import RegexSyntaxError as OracleSyntaxError
from mrx_ref.describe import describe
from vector_eval import load_vectors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EXTRA = [
    "(x|y|foo|bar)+z", "[a-z]{2,}", "[a-z]{0,3}", "[^abc]+\\d", "\\s*\\d+", "\\w+@\\w+\\.com",
    "[a-zA-Z0-9._%+-]+@[a-zA-Z0-9.-]+\\.[a-z]{2,}", "(?:00|33|44)\\d{3}", "a{3}b",
    "ab{3}cdefghijklmnopq", "hello world this is long", "a\\.bcdefghijklmnopq", "(abc)+", "(abc)*",
    "(abc)?", "(a|b)*", "(cat|dog)+", "(hello|help|helicopter)", ".+", ".?", ".", "^", "$", "^$", "",
    "[", "(", ")", "a{x}", "a|", "|", "\\t", "[a-Z]", "x[0-9]{2,4}y", "\\d{3}-\\d{4}",
    "([A-Z]{3}[0-9]{4})-([A-Z]{3}[0-9]{3})", "3[02]|40|[68]9", "(a|b)x", "h[ae]llo", "hello.world",
    "^[a-z]+$", "(\\d+)", "[+]*\\d+[-]*\\d+[-]*\\d+[-]*\\d+", "[0-9]+\\.?[0-9]*", "a*", "test+",
    "[a-c]+[x-z]?", "(?:a|b)+", "((a|b)|(c|d))", "a|b|c|d|e|f|g|h|i", "[a-z]+\\d+", "\\d+", "hello",
    "(\\d{3})(\\d{3})(\\d{4})", "(x|y|foo|bar)+", ".*", "a**", "a^b", "[(]", "[a|b]", "x{2,}", "x{,3}",
]


def _patterns():
    pats = sorted({v["pattern"] for v in load_vectors() if v.get("pattern") is not None})
    return pats + [p for p in EXTRA if p not in pats]


def _product_describe(p: bytes) -> str:
    try:
        return M.CompiledRegex(p).describe()
    except M.RegexSyntaxError as e:
        return "SYNTAX:" + str(e)


def _oracle_describe(p: bytes) -> str:
    try:
        return describe(p)
    except OracleSyntaxError as e:
        return "SYNTAX:" + str(e)


@pytest.mark.parametrize("pattern", _patterns())
def test_tables_match_oracle(pattern):
    pb = pattern.encode("utf-8")
    od, pd = _oracle_describe(pb), _product_describe(pb)
    if od.startswith("SYNTAX") or pd.startswith("SYNTAX"):
        assert od == pd
        return
    ol = od.strip().split("\n")
    pl = [l for l in pd.strip().split("\n")
          if not l.startswith(("support.", "device.", "nfa.has_filter"))]
    assert ol == pl


def test_library_exports_every_header_symbol():
    lib = M.load_library()
    hdr = open(os.path.join(ROOT, "include", "mrx.h")).read()
    declared = sorted(set(re.findall(r"\b(mrx_[a-z_]+)\s*\(", hdr)))
    assert len(declared) >= 24
    assert sorted(api.EXPORTED_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    hdr = open(os.path.join(ROOT, "include", "mrx_testing.h")).read()
    hooks = sorted(set(re.findall(r"\b(mrx_[a-z_]+)\s*\(", hdr)))
    assert sorted(api.TESTING_SYMBOLS) == hooks
    for name in hooks:
        assert hasattr(lib, name), name
    hdr = open(os.path.join(ROOT, "include", "mrx_comm.h")).read()
    comm = sorted(set(re.findall(r"\b(mrx_[a-z_]+)\s*\(", hdr)))
    assert sorted(api.COMM_SYMBOLS) == comm
    for name in comm:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.mrx_version()


def test_config_plans():
    """Routing + kernel plan of the five BASELINE.json configs."""
    want = {
        "hello": ("literal", "streamable=yes st_nstates=6 st_kind=3"),
        "[a-z]+\\d+": ("multi_class_sequence", "streamable=yes st_nstates=3 st_kind=1"),
        "\\d+": ("single_class", "streamable=yes st_nstates=2 st_kind=1"),
        "(\\d{3})(\\d{3})(\\d{4})": ("multi_class_sequence", "streamable=yes st_nstates=11 st_kind=2"),
        "(x|y|foo|bar)+": ("alternation", "streamable=yes st_nstates=6 st_kind=3"),
    }
    for pat, (shape, st) in want.items():
        d = M.CompiledRegex(pat).describe()
        assert "engine_type=DFA\n" in d and ("dfa.shape=%s\n" % shape) in d, d
        assert ("device." + st) in d, d
        assert "support.search=yes" in d and "support.match_first=yes" in d


def test_out_of_scope_patterns_fail_loudly_without_a_gpu():
    # routed by the reference to OnePass / the backtracking NFA: refused, never guessed
    rx = M.CompiledRegex("^aaaa.*a$")   # '$' program that is not one-pass: the backtracking matcher's flat program
    d = rx.describe()
    assert "support.match_first=yes" in d and "device.backtrack=yes" in d
    d = M.CompiledRegex("^(a|b)*a.*a$").describe()   # alternation / quantified groups: flat program since round 2
    assert "support.match_first=yes" in d and "device.backtrack=yes" in d
    d = M.CompiledRegex("^" + "(" * 17 + "a" + ")" * 17 + ".*a$").describe()   # outside the flat form: refused
    assert "support.match_first=reference routes" in d and "device.backtrack=no: groups nested deeper than 16" in d
    d = M.CompiledRegex("^" + "".join(c + "*" for c in "abcdefghijklmnopqrstuvwxyzABCDEFG") + ".*a$").describe()
    assert "device.backtrack=no: more than 30 open choices" in d
    d = M.CompiledRegex("^[a-z]+[0-9]+$").describe()   # one-pass: match_first on the OnePass tables
    assert "support.match_first=yes" in d and "onepass=yes" in d
    # its search runs on the LazyDFA upstream (matcher.mojo:401-431): served with a per-text transition cache (round 3)
    assert "support.search=yes" in d and "device.lazy_end_cache=yes" in d
    # ... as long as the states of both transition variants fit one 64-bit mask
    d = M.CompiledRegex("(a|b)*a(a|b){5}$").describe()
    assert "support.search=LazyDFA search with '$': the per-text transition cache is tracked for at most 64" in d, d
    rx = M.CompiledRegex("hello.world")   # literal-prefiltered backtracker search: flat program
    assert "support.search=yes" in rx.describe() and "literal_opt=1" in rx.describe()
    rx = M.CompiledRegex("hello(a|b)*world")   # the quantified group is not the last child: zero repetitions upstream
    assert "support.search=yes" in rx.describe() and "device.backtrack=yes" in rx.describe()
    with pytest.raises(M.RegexSyntaxError, match=r"Missing closing '\]'"):
        M.CompiledRegex("[abc")
    with pytest.raises(M.RegexSyntaxError, match="Unescaped closing parenthesis"):
        M.CompiledRegex("a)")


def test_arrow_large_binary_is_the_packed_batch_form():
    """an Arrow LargeBinary array's (offsets, data) buffers are the C ABI's
    packed batch form; slices and nulls included (CPU tensors here, no GPU needed)."""
    import numpy as np
    import pyarrow as pa
    texts = [b"hello123", b"", b"world456 test789", None, b"x"]
    arr = pa.array(texts, type=pa.large_binary())
    b = M.DeviceBatch.from_arrow(arr, device="cpu")
    want = [t or b"" for t in texts]
    data, offsets = api.pack_texts(want)
    assert b.n == 5 and np.array_equal(b.offsets.numpy(), offsets) and np.array_equal(b.data.numpy(), data)
    sl = M.DeviceBatch.from_arrow(arr.slice(2, 3), device="cpu")
    d2, o2 = api.pack_texts(want[2:5])
    assert np.array_equal(sl.offsets.numpy(), o2) and np.array_equal(sl.data.numpy(), d2)
    st = M.DeviceBatch.from_arrow(pa.array(["ab12", "cd"]), device="cpu")   # plain string array
    assert st.offsets.tolist() == [0, 4, 6] and bytes(st.data.numpy().tobytes()) == b"ab12cd"


def test_synchronising_bytes_and_long_text_plan_flags():
    """Plan facts the long-text kernels rely on (mrx_plan.cpp): which bytes take every state of the
    search automaton to the same state with the same start; the KMP automaton with restart for a
    literal whose prefix is also a suffix; the class-indexed stepper table beyond 96 states."""
    import re
    from mojo_regex_amd import api as M
    import bench_engine_cases as B

    def line(p, key):
        return re.search(key + r"=[^\n]*", M.CompiledRegex(p).describe()).group(0)

    assert "sync_bytes=220" in line(b"[a-z]+\\d+", "device.streamable")      # everything outside [a-z0-9]
    assert "sync_bytes=246" in line(b"\\d+", "device.streamable")            # every non-digit
    assert "sync_bytes=253" in line(b"hello", "device.streamable")           # non-literal bytes and 'h' (always prefix length 1)
    assert "sync_bytes=255" in line(b"a{2,4}", "device.streamable")
    assert "reset_byte=255" in line(b"[a-z]+\\d+", "device.streamable")        # all states -> idle, accepting ones emit
    assert "reset_byte=-1" in line(b"[^a]+b", "device.streamable")              # (not streamable at all)
    s = line(b"555-123-4567", "device.streamable")                             # "5" is prefix and suffix: KMP with restart
    assert s.startswith("device.streamable=yes") and "st_nstates=13" in s
    assert line(b"abab", "device.streamable").startswith("device.streamable=yes")
    assert "big_table=1" in line(B.NANPA_PATTERN, "device.steppable")          # 154 states > 96
    assert "big_table" not in line(b"\\w+\\d{2}", "device.steppable")


def test_backtracker_chain_classification():
    """DevPlan::bt_flags bit 5 (`chain=1` in mrx_describe): the flat program needs no choice stack -- no ALT / LOOP
    items, and no quantified leaf whose shorter counts could rescue what follows it."""
    def chain(p):
        d = M.CompiledRegex(p).describe()
        line = [x for x in d.split("\n") if x.startswith("device.backtrack=yes")]
        assert line, (p, d)
        return "chain=1" in line[0]
    for p in ("(\\w+) (\\w+)", "(\\d+)-(\\d+)", "hello.*", "x(.*)y", "(\\w+)@(\\w+)\\.com", "([a-z]+)(\\d*)x", "\\d+-\\d+x.*"):
        assert chain(p), p
    # '.*' / '\\w+' in front of something they can also match; alternation; a looping group
    for p in ("hello.*world", ".*@example\\.com", "(a|b)(c)", "(ab)+(c)", "\\w+ing.*", "\\w+s \\w+x.*"):
        assert not chain(p), p


def test_chain_groups_form_is_proven_on_the_tables():
    """HostPlan::chain (`chain_groups=yes|no: why` in mrx_describe): regex.sub with \\1..\\9 may take the plain search's
    spans only where the chain's matches are provably the table walk's -- a deterministic chain without anchors or a
    literal prefilter, every leaf's three membership tests equal, a leaf of variable count disjoint from the leaf behind
    it (also the last leaf of a group, which the matcher never backs off), and the chain's automaton equal to the
    engine's table pair by pair."""
    def verdict(p):
        d = M.CompiledRegex(p).describe()
        line = [x for x in d.split("\n") if x.startswith("device.backtrack=yes")]
        assert line, (p, d)
        return line[0].split("chain_groups=")[1]
    for p, leaves in (("(\\w+) (\\w+)", 3), ("([a-z]+)(\\d+)", 2), ("([a-z]+)-(\\d{2,4})", 3), ("(\\d+)\\.(\\d+)", 3),
                      ("((\\d+)-([a-z]+))", 3), ("\\w(a{2,})", 2), ("(@+(([a-z]{2,})))", 2)):
        assert verdict(p) == "yes leaves=%d" % leaves, (p, verdict(p))
    # the group's last leaf would have to give a byte back, which the matcher never does: its matches are not the walk's
    assert verdict("(\\w+)x").startswith("no: a leaf with a variable count shares a byte"), verdict("(\\w+)x")
    assert verdict("(a|b)(c)").startswith("no: not a deterministic chain")
    assert verdict("(\\s+)(x)").startswith("no: a leaf's three membership tests differ"), verdict("(\\s+)(x)")
    assert verdict("^(\\w+) (\\w+)").startswith("no: "), verdict("^(\\w+) (\\w+)")


def test_chain_groups_verdict_implies_the_greedy_leftmost_parse():
    """Where build_plan says chain_groups=yes, the reference's sub with group templates (the oracle's restatement of
    NFAEngine.match_next_with_groups, C twin of its backtracker) must be what a leftmost search with ONE greedy parse per
    start gives -- that is what the spans of the table walk plus runs of the leaves' classes compute on the GPU.  Python's
    `re` has exactly those semantics on such chains (ASCII classes on bytes patterns), so it stands in for them here:
    the verdict is checked without a GPU, on generated chains with random groups and templates."""
    import numpy as np
    import mrx_ref as O
    import mrx_ref.hybrid as H
    from test_gpu_parity import _random_chain_with_groups, _random_texts
    rng = np.random.default_rng(4242)
    al = b"abcxyz0123456789 -.@:af"
    texts = _random_texts(rng, 60, 60, al) + _random_texts(rng, 6, 400, al) + [b"", b"a", b"ab 12", b"hello world foo",
                                                                              b"aa-bb.cc@dd:ee", b"abc 123 abc 123 " * 20]
    before = H.USE_C_BACKTRACK
    H.USE_C_BACKTRACK = True
    proven = 0
    try:
        for _ in range(300):
            pat, repl = _random_chain_with_groups(rng)
            try:
                rx = M.CompiledRegex(pat.decode())
            except M.RegexSyntaxError:
                continue
            if "chain_groups=yes" not in rx.describe() or "fixed_total=-1" not in rx.describe():
                continue   # (patterns of the fixed-width (\\d{N}) form never take general groups: matcher.mojo:1726-1744)
            cre = re.compile(pat)
            # (a reference to a group the pattern lacks is empty upstream, an error in `re`)
            repl_py = re.sub(rb"\\([1-9])", lambda m: m.group(0) if int(m.group(1)) <= cre.groups else b"", repl)
            proven += 1
            for count in (0, 2):
                for t in texts:
                    assert O.sub(pat, repl, t, count) == cre.sub(repl_py, t, count), (pat, repl, count, t)
    finally:
        H.USE_C_BACKTRACK = before
    assert proven >= 60, proven


def test_fixed_width_group_patterns_that_are_nothing_but_groups():
    """HostPlan::fixed_pure (`device.sub_groups=fixed pure=1`): the pattern is (\\d{N}) / (\\d) groups end to end -- only
    then does every match hold all its group windows and equal the whole-text shortcut of regex.sub
    (matcher.mojo:1726-1744), so only then does a group template stay on the spans route unchecked."""
    def pure(p):
        line = [x for x in M.CompiledRegex(p).describe().split("\n") if x.startswith("device.sub_groups=")]
        return line[0].endswith("pure=1") if line else None
    for p in ("(\\d{3})(\\d{3})(\\d{4})", "(\\d)", "(\\d)(\\d{2})", "(\\d{10})"):
        assert pure(p) is True, p
    for p in ("(\\d)*", "(\\d)+", "(\\d)?", "(\\d)*a", "(\\d{4})-(\\d{2})-(\\d{2})", "^(\\d)+", "x(\\d)?", "(\\d{2})+", "(\\d)x"):
        assert pure(p) is False, p
    for p in ("(\\w+) (\\w+)", "[a-z]+(\\d)*", "hello"):
        assert pure(p) is None, p


def test_emptywalk2_table_run_on_the_host_equals_the_oracle():
    """The one-pass table of empty-match plans whose walks read beyond their match (build_emptywalk2(): pending tries
    behind the oldest walk, chased when it dies), run on the HOST through mrx_testing_emptywalk_findall: findall of
    hand-picked and generated patterns against the oracle, before any kernel sees the table."""
    import ctypes as C
    import numpy as np
    import mojo_regex_amd as M
    from mrx_ref import hybrid as O
    from pattern_gen import patterns, patterns2
    lib = M.load_library()

    def run(rx, t):
        buf = (C.c_int32 * (4 * (len(t) + 2)))()
        k = lib.mrx_testing_emptywalk_findall(rx._h, t, len(t), buf, len(buf) // 2)
        return None if k < 0 else [(buf[2 * i], buf[2 * i + 1]) for i in range(k)]

    rng = np.random.default_rng(7)
    texts = [b"", b"a", b"ab", b"aba", b"abab", b"abcabcab", b"abx", b"xyxy", b"foobar", b"fooba", b"barfoo", b"http", b"htt",
             b"catdogdog1", b"aabbb", b"abcab"]
    texts += [bytes(rng.choice(list(b"abcdxyfotrh p1g"), size=int(rng.integers(1, 40))).tolist()) for _ in range(60)]
    tables = 0
    for p in (b"(abc)*", b"a+b*", b"http?", b"ca*t", b"x?y?", b"(foo)?(bar)?", b"cat|(dog){0,2}\\d?"):
        rx = M.compile_regex(p)
        assert "empty_walk2=yes" in rx.describe(), p
        tables += 1
        for t in texts:
            assert run(rx, t) == O.findall(p, t), (p, t)
    seen = set()
    for gen, seeds in ((patterns, (20260503, 20260504)), (patterns2, (20260601,))):
        for sd in seeds:
            for ps in gen(sd, 300):
                if ps in seen:
                    continue
                seen.add(ps)
                p = ps.encode()
                try:
                    rx = M.compile_regex(p)
                except Exception:
                    continue
                if "empty_walk2=yes" not in rx.describe():
                    continue
                tables += 1
                al = b"abcxyz019 -@.fobrhelcatdg" + bytes(c for c in p if chr(c).isalnum()) * 2
                for _ in range(25):
                    t = bytes(rng.choice(list(al), size=int(rng.integers(0, 60))).tolist())
                    assert run(rx, t) == O.findall(p, t), (p, t)
    assert tables > 15, tables
