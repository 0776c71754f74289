"""The oracle against the reference's own known-answer tests (CPU only).

tests/golden/reference_vectors.json holds the vectors transcribed from the
reference's tests/test_*.mojo (file:line per vector).  This is what pins the
oracle: every vector must pass -- the DFA / LazyDFA / OnePass routes, the vectors the
hybrid router sends to the backtracking matcher, and (round 3) the 104 vectors of the
reference's tests/test_nfa.mojo, which drive NFAEngine directly (`nfa_match_first`,
`nfa_findall`: regex.nfa's module functions, nfa.mojo:1733-1769) and pin
oracle/mrx_ref/backtrack.py.
"""
import collections

import pytest

from vector_eval import load_vectors, evaluate, Unsupported
from mrx_ref import UnsupportedByOracle

VECS = load_vectors()


def test_fixture_is_large_enough():
    assert len(VECS) >= 580
    files = {v["file"] for v in VECS}
    assert "tests/test_matcher.mojo" in files and "tests/test_dfa.mojo" in files
    ops = collections.Counter(v["op"] for v in VECS)
    assert ops["nfa_match_first"] >= 94 and ops["nfa_findall"] >= 10   # tests/test_nfa.mojo


def test_oracle_passes_every_in_scope_reference_vector(oracle_backend):
    failures, unsupported, passed = [], collections.Counter(), 0
    for v in VECS:
        try:
            f = evaluate(oracle_backend, v)
        except (UnsupportedByOracle, Unsupported):
            unsupported[(v["op"], v["pattern"])] += 1
            continue
        if f:
            failures.extend(f)
        else:
            passed += 1
    assert not failures, "\n".join(failures[:20])
    # with the backtracking matcher restated (oracle/mrx_ref/backtrack.py) every vector is answered
    assert passed == len(VECS) and not unsupported, (passed, unsupported)


# the SURVEY.md Appendix B pins for the five BASELINE.json configs, spelled out
CONFIG_PINS = [
    ("match_first", b"hello", b"hello world", (0, 5)),
    ("match_first", b"hello", b"say hello there", None),
    ("search", b"hello", b"say hello world", (4, 9)),
    ("findall", b"a", b"banana", [(1, 2), (3, 4), (5, 6)]),
    ("findall", b"aa", b"aaaa", [(0, 2), (2, 4)]),
    ("match_first", b"[a-z]+[0-9]+", b"hello123", (0, 8)),
    ("match_first", b"[a-z]+[0-9]+", b"Hello123", None),
    ("findall", b"[a-z]+[0-9]+", b"hello123 world456 test789", [(0, 8), (9, 17), (18, 25)]),
    ("search", b"[a-z]+[0-9]+", b"QQab12ZZ", (2, 6)),
    ("findall", b"[a-z]+\\d+", b"hello123 world456 test789", [(0, 8), (9, 17), (18, 25)]),
    ("findall", b"\\d+", b"abc123def456ghi", [(3, 6), (9, 12)]),
    ("findall", b"[0-9]+", b"1 22 333", [(0, 1), (2, 4), (5, 8)]),
    ("search", b"[0-9]+", b"order 1234 shipped", (6, 10)),
    ("match_first", b"\\d+", b"", None),
    ("search", b"(\\d{3})(\\d{3})(\\d{4})", b"Call 6502530000 now", (5, 15)),
    ("search", b"a+b", b"xxaaabby", (4, 6)),          # quirk A.6 #2 (pinned upstream)
    ("search", b"a+b*", b"xxaaabby", (0, 0)),
    ("search", b"(a|b)x", b"zbxq", (1, 3)),            # LazyDFA path
    # PARITY-UNPINNED (source-derived): the '+' of a quantified literal
    # alternation is dropped by the DFA compiler (SURVEY.md A.2 cfg 5)
    ("findall", b"(x|y|foo|bar)+", b"xyfoo", [(0, 1), (1, 2), (2, 5)]),
    ("match_first", b"(x|y|foo|bar)+", b"xyfoo", (0, 1)),
]


@pytest.mark.parametrize("op,pat,text,want", CONFIG_PINS)
def test_config_pins(oracle_backend, op, pat, text, want):
    assert getattr(oracle_backend, op)(pat, text) == want


def test_config_routing_matches_survey_a2(oracle_backend):
    from mrx_ref import compile_regex
    want = {
        b"hello": ("literal", 6), b"[a-z]+\\d+": ("multi_class_sequence", 3),
        b"\\d+": ("single_class", 2), b"(\\d{3})(\\d{3})(\\d{4})": ("multi_class_sequence", 11),
        b"(x|y|foo|bar)+": ("alternation", 6),
    }
    for pat, (shape, nstates) in want.items():
        c = compile_regex(pat)
        assert c.matcher.use_dfa and c.matcher.get_engine_type() == "DFA"
        assert c.matcher.dfa.shape == shape and len(c.matcher.dfa.states) == nstates
        assert c.matcher.prefilter_literal is None and not c.matcher.is_exact_literal
    # the neighbour that does reach LazyDFA (SURVEY.md A.2 cfg 5, last cell)
    c = compile_regex(b"(x|y|foo|bar)+z")
    assert not c.matcher.use_dfa and c.matcher.nfa_matcher.lazy is not None


def test_sub_pins(oracle_backend):
    be = oracle_backend
    assert be.sub(b"(\\d{3})(\\d{3})(\\d{4})", b"\\1-\\2-\\3", b"6502530000") == b"650-253-0000"
    assert be.sub(b"(\\d{3})(\\d{3})(\\d{4})", b"\\1-\\2-\\3",
                  b"6502530000 and 4155551234", 1) == b"650-253-0000 and 4155551234"
    assert be.sub(b"(\\d{4})-(\\d{2})-(\\d{2})", b"\\2/\\3/\\1",
                  b"Date: 2026-04-12 is today") == b"Date: 04/12/2026 is today"
    assert be.sub(b"hello", b"\\0hi", b"hello world") == b"\\0hi world"


def test_onepass_engine_vectors():
    """The reference's own OnePass tests (tests/test_onepass.mojo), engine-level API."""
    import json
    import os
    from mrx_ref.frontend import parse
    from mrx_ref.pikevm import compile_ast
    from mrx_ref.onepass import compile_onepass
    path = os.path.join(os.path.dirname(__file__), "golden", "onepass_vectors.json")
    for v in json.load(open(path))["vectors"]:
        op = compile_onepass(compile_ast(parse(v["pattern"].encode())))
        if v["op"] == "compiles":
            assert (op is not None) == v["want"], v
            continue
        got = getattr(op, v["op"])(v["text"].encode(), 0)
        if "want_end" in v:
            assert got is not None and got[1] == v["want_end"], v
        else:
            assert got == (tuple(v["want"]) if v["want"] is not None else None), v


def test_onepass_routing_and_rejections():
    """NFAMatcher.match_first (matcher.mojo:361-380): '$' programs go to OnePass when they
    compile one-pass, otherwise to the backtracker (out of scope)."""
    from mrx_ref import hybrid as H
    assert H.match_first(b"^[a-z]+[0-9]+$", b"abc123") == (0, 6)
    assert H.match_first(b"^[a-z]+[0-9]+$", b"abc123x") is None
    assert H.match_first(b"^a|b$", b"a") == (0, 1)        # '^a' branch accepts mid-text (onepass.mojo:449-453)
    assert H.match_first(b"^a|b$", b"ab") == (0, 1)
    assert H.match_first(b"^\\d+$", b"12345") == (0, 5)
    # `.` and `a` both fire: not one-pass -> the backtracking matcher (oracle/mrx_ref/backtrack.py).  Its
    # greedy `.*` leaves nothing for the final `a` and a quantified leaf followed by siblings is only
    # backed off inside _match_with_backtracking: the reference's own answer, whatever re.match says.
    r = H.compile_regex(b"^aaaa.*a$")
    assert r.matcher.nfa_matcher.onepass is None
    # traced by hand: `.*` is a quantified leaf followed by siblings -> _match_with_backtracking
    # (nfa.mojo:1236-1311) gives back one byte at a time: count 1 -> 0, then `a` and `$` match at 4
    assert H.match_first(b"^aaaa.*a$", b"aaaaa") == (0, 5)
    assert H.match_first(b"^aaaa.*a$", b"aaaab") is None
    # search with '$' stays on the LazyDFA (matcher.mojo:401-431), whose cached transitions carry "'$' held" or
    # "'$' did not hold" from their FIRST computation (pikevm.mojo:869-942).  Restated with the cache empty at the
    # start of every call (oracle/mrx_ref/hybrid.py, module docstring); traced by hand through _run_lazy:
    #   "abc":    (S1, 'c') is first computed while the last byte is consumed -> closes with '$' -> match
    #   "abcabc": (S1, 'c') is first computed at position 2, inside the text -> cached without '$' -> the walk from 0
    #             ends without a match at 6, and so does every later start (the start closure passes '^' anywhere)
    #   "abcabd": 'd' only occurs as the last byte -> match
    assert H.search(b"^[a-z]+$", b"abc") == (0, 3)
    assert H.search(b"^[a-z]+$", b"abcabc") is None
    assert H.search(b"^[a-z]+$", b"abcabd") == (0, 6)
    assert H.search(b"^[a-z]+$", b"abc") == (0, 3)        # (the call before left nothing behind)
    H.FRESH_LAZY_CACHE = False                             # as upstream: the cache lives as long as the object
    try:
        H.clear_regex_cache()
        assert H.search(b"^[a-z]+$", b"abcabc") is None
        assert H.search(b"^[a-z]+$", b"abc") is None       # (S1, 'c') was cached without '$' by the call before
    finally:
        H.FRESH_LAZY_CACHE = True
        H.clear_regex_cache()


def test_hand_traced_backtracker_quirks(oracle_backend):
    """tests/golden/backtrack_quirk_vectors.json: NFAEngine quirks the reference's own tests do not reach,
    traced by hand through nfa.mojo (each vector cites the lines that decide it)."""
    import json
    import os
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "backtrack_quirk_vectors.json")))
    assert len(doc["vectors"]) >= 20
    for v in doc["vectors"]:
        got = getattr(oracle_backend, v["op"])(v["pattern"].encode("latin-1"), v["text"].encode("latin-1"))
        if v["op"] == "nfa_findall":
            assert [list(x) for x in got] == v["want"], (v, got)
        else:
            assert (list(got) if got is not None else None) == v["want"], (v, got)


def test_optimizer_vectors_pin_classifier_and_literal_helpers():
    """tests/test_optimizer.mojo: classify / is_literal_pattern / get_literal_string /
    pattern_has_anchors -- the oracle's analysis.py, and (classification) the product's plan."""
    import json
    import os
    from mrx_ref import analysis as A
    from mrx_ref.frontend import parse
    import mojo_regex_amd as M
    names = {A.SIMPLE: "SIMPLE", A.MEDIUM: "MEDIUM", A.COMPLEX: "COMPLEX"}
    path = os.path.join(os.path.dirname(__file__), "golden", "optimizer_vectors.json")
    vecs = json.load(open(path))["vectors"]
    assert len(vecs) >= 53
    for v in vecs:
        ast = parse(v["pattern"].encode())
        if v["fn"] == "classify":
            assert names[A.classify(ast)] in v["want"], v
            d = M.CompiledRegex(v["pattern"]).describe()
            assert any(("complexity=%s\n" % w) in d for w in v["want"]), (v, d[:120])
        elif v["fn"] == "is_literal_pattern":
            assert A.is_literal_pattern(ast) == v["want"], v
        elif v["fn"] == "get_literal_string":
            assert A.get_literal_string(ast) == v["want"].encode(), v
        elif v["fn"] == "has_literal_prefix":
            assert A.has_literal_prefix(ast) == v["want"], v
        elif v["fn"] == "search":
            from mrx_ref import hybrid as H
            assert H.search(v["pattern"].encode(), v["text"].encode())[0] == v["want_start"], v
        elif v["fn"] == "findall_count":
            from mrx_ref import hybrid as H
            assert len(H.findall(v["pattern"].encode(), v["text"].encode())) == v["want"], v
        else:
            assert list(A.pattern_has_anchors(ast)) == v["want"], v


def test_simd_class_vectors():
    """tests/test_simd.mojo: CharacterClassSIMD.contains / find_first_nibble_match (a9)."""
    import json
    import os
    from mrx_ref.dfa_engine import ClassMatcher
    from mrx_ref import hybrid as H
    path = os.path.join(os.path.dirname(__file__), "golden", "simd_vectors.json")
    for v in json.load(open(path))["vectors"]:
        if v["fn"] == "contains":
            assert ClassMatcher.for_class(v["class"].encode()).contains(ord(v["char"])) == v["want"], v
        elif v["fn"] == "find_first_nibble_match":
            t = v["text"].encode()
            assert ClassMatcher.for_class(v["class"].encode()).find_first_nibble_match(t, v["start"], len(t)) == v["want"], v
        else:
            t = v["text"].encode()
            assert [t[a:b].decode() for a, b in H.findall(v["pattern"].encode(), t)] == v["want"], v


def test_patterns_the_reference_requires_to_raise():
    """tests/test_lexer.mojo / tests/test_parser.mojo `with assert_raises()`: the oracle's front end
    and the product's (MRX_E_SYNTAX) both refuse these patterns, with the same message."""
    import json
    import os
    import mojo_regex_amd as M
    from mrx_ref import RegexSyntaxError as OracleSyntaxError
    from mrx_ref.frontend import parse
    path = os.path.join(os.path.dirname(__file__), "golden", "syntax_error_vectors.json")
    for v in json.load(open(path))["vectors"]:
        with pytest.raises(OracleSyntaxError) as eo:
            parse(v["pattern"].encode())
        with pytest.raises(M.RegexSyntaxError) as ep:
            M.CompiledRegex(v["pattern"])
        assert str(eo.value) == str(ep.value), v


def test_parser_ast_shape_vectors():
    """tests/test_parser.mojo: node types, child counts, quantifier bounds, values and negation
    flags at given child paths (tests/golden/ast_vectors.json, made by extract_ast_vectors.py)."""
    import json
    import os
    from mrx_ref import frontend as F
    path = os.path.join(os.path.dirname(__file__), "golden", "ast_vectors.json")
    vecs = json.load(open(path))["vectors"]
    assert len(vecs) >= 60
    for v in vecs:
        node = F.parse(v["pattern"].encode())
        for i in v["path"]:
            node = node.get_child(i)
        a = v["attr"]
        if a == "type":
            got = F.TYPE_NAMES[node.type]
        elif a == "children_len":
            got = node.get_children_len()
        elif a == "value":
            got = node.get_value().decode()
        else:
            got = getattr(node, a)
        assert got == v["want"], v


def test_lexer_token_vectors():
    """tests/test_lexer.mojo: token types / characters at given indexes."""
    import json
    import os
    from mrx_ref import frontend as F
    names = {"ELEMENT": F.T_ELEMENT, "COMMA": F.T_COMMA, "START": F.T_START, "END": F.T_END, "DASH": F.T_DASH,
             "SPACE": F.T_SPACE, "WILDCARD": F.T_WILDCARD, "LEFTPARENTHESIS": F.T_LPAREN,
             "RIGHTPARENTHESIS": F.T_RPAREN, "LEFTBRACKET": F.T_LBRACKET, "RIGHTBRACKET": F.T_RBRACKET,
             "LEFTCURLYBRACE": F.T_LCURLY, "RIGHTCURLYBRACE": F.T_RCURLY, "ASTERISK": F.T_ASTERISK,
             "PLUS": F.T_PLUS, "QUESTIONMARK": F.T_QMARK, "VERTICALBAR": F.T_VBAR}
    path = os.path.join(os.path.dirname(__file__), "golden", "lexer_vectors.json")
    for v in json.load(open(path))["vectors"]:
        toks = F.scan(v["pattern"].encode())
        if "len" in v:
            assert len(toks) == v["len"], v
            continue
        t = toks[v["index"]]
        assert t.type == names[v["type"]], (v, t)
        if "char" in v:
            assert t.char == ord(v["char"]), (v, t)
