"""CPU: the C half of the oracle (oracle/c/mrx_oracle.c) against the Python half
on the reference's vector patterns and on seeded random batches."""
import zlib

import numpy as np
import pytest

from mrx_ref import hybrid as O, UnsupportedByOracle
from mrx_ref.cfast import CDfa
from vector_eval import load_vectors

PATS = [b"hello", b"a", b"[a-z]+\\d+", b"\\d+", b"[0-9]*", b"[a-z]{3}", b"[^0-9]+",
        b"(\\d{3})(\\d{3})(\\d{4})", b"(x|y|foo|bar)+", b"a+b", b"a+b*", b"(abc)*", b"^a", b"a$",
        b"[a-z]*[0-9]+", b"[0-9]+\\.?[0-9]*", b"[A-Z][a-z]+[0-9]+", b"", b".+", b"x[0-9]{2,4}y",
        b"[a-zA-Z0-9._%+-]+x", b"\\s*\\d+"]


def _batch(rng, n, max_len, alphabet):
    al = np.frombuffer(alphabet, dtype=np.uint8)
    texts = [bytes(rng.choice(al, size=int(k)).tolist()) for k in rng.integers(0, max_len + 1, size=n)]
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum([len(t) for t in texts], out=offsets[1:])
    data = np.frombuffer(b"".join(texts) + b"\0", dtype=np.uint8)[:-1]
    return texts, data, offsets


@pytest.mark.parametrize("pat", PATS)
def test_c_oracle_equals_python_oracle(pat):
    rng = np.random.default_rng(zlib.crc32(pat))
    try:
        cd = CDfa(pat)
    except UnsupportedByOracle:
        pytest.skip("not a plain DFAEngine route")
    for alphabet in (b"abcxyz0189 -.fobar\n", bytes(range(256)), b"0123456789ab "):
        texts, data, offsets = _batch(rng, 120, 80, alphabet)
        counts, spans, total = cd.findall_batch(data, offsets)
        fs, fe = cd.span_batch("match_first", data, offsets)
        ss, se = cd.span_batch("search", data, offsets)
        k = 0
        for i, t in enumerate(texts):
            want = O.findall(pat, t)
            assert counts[i] == len(want)
            assert [tuple(int(x) for x in r) for r in spans[k:k + len(want)]] == want, (pat, t)
            k += len(want)
            w = O.match_first(pat, t)
            assert (int(fs[i]), int(fe[i])) == (w if w else (-1, -1)), (pat, t)
            w = O.search(pat, t)
            assert (int(ss[i]), int(se[i])) == (w if w else (-1, -1)), (pat, t)
        assert k == total


def test_c_oracle_on_reference_vector_texts():
    done = 0
    for v in load_vectors():
        if v["op"] not in ("findall", "match_first", "search") or v.get("pattern") is None:
            continue
        pat, text = v["pattern"].encode(), v["text"].encode()
        try:
            cd = CDfa(pat)
        except UnsupportedByOracle:
            continue
        data = np.frombuffer(text + b"\0", dtype=np.uint8)[:-1]
        offsets = np.array([0, len(text)], np.int64)
        counts, spans, total = cd.findall_batch(data, offsets)
        assert [tuple(int(x) for x in r) for r in spans] == O.findall(pat, text)
        done += 1
    assert done > 150


# ---- the backtracking matcher's C twin (oracle/c/mrx_backtrack.c) against backtrack.py ----------------------------
def _both(pat):
    from mrx_ref import hybrid as H
    from mrx_ref.cbacktrack import CBacktrack
    py = H.nfa_engine(pat)
    assert py.__class__.__name__ == "BacktrackNFA"
    return py, CBacktrack(py)


def _same(py, c, text, starts=(0,)):
    assert c.match_all(text) == py.match_all(text), ("match_all", py.pattern, text)
    for st in starts:
        assert c.match_first(text, st) == py.match_first(text, st), ("match_first", py.pattern, text, st)
        assert c.match_next(text, st) == py.match_next(text, st), ("match_next", py.pattern, text, st)
        assert c.match_next_with_groups(text, st) == py.match_next_with_groups(text, st), ("groups", py.pattern, text, st)


def test_c_backtracker_on_the_reference_nfa_vectors_and_quirks():
    import json
    import os
    done = 0
    for v in load_vectors():
        if not v["op"].startswith("nfa_"):
            continue
        py, c = _both(v["pattern"].encode())
        _same(py, c, v["text"].encode())
        done += 1
    assert done >= 104
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "backtrack_quirk_vectors.json")))
    for v in doc["vectors"]:
        py, c = _both(v["pattern"].encode("latin-1"))
        _same(py, c, v["text"].encode("latin-1"))


def test_c_backtracker_equals_python_on_generated_patterns():
    import random
    from pattern_gen import patterns, patterns2
    from mrx_ref import RegexSyntaxError
    rng = np.random.default_rng(77)
    alphabets = [np.frombuffer(a, dtype=np.uint8) for a in (b"abcxyz0189 -.@\n", b"abfoo bar 12hello", bytes(range(256)))]
    r = random.Random(5)
    done = 0
    for pat in patterns(31, 220) + patterns2(32, 220):
        try:
            py, c = _both(pat.encode())
        except RegexSyntaxError:
            continue
        if py.regex is None:
            continue
        for al in alphabets:
            for _ in range(4):
                t = bytes(rng.choice(al, size=int(rng.integers(0, 70))).tolist())
                _same(py, c, t, starts=(0, r.randrange(0, len(t) + 2)))
        # one long text: runs past the 50 / 100-byte cut-offs of match_first mode and the > 8 SIMD switch
        t = bytes(rng.choice(alphabets[0], size=400).tolist())
        _same(py, c, t)
        done += 1
    assert done > 350
