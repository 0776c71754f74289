"""Shared evaluator for tests/golden/reference_vectors.json.

A *backend* is an object with the methods used below (the oracle and the
product's Python host layer both provide one), so the same reference
known-answer vectors pin the oracle (CPU) and the HIP path (GPU).
"""
from __future__ import annotations

import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
VECTORS = os.path.join(HERE, "golden", "reference_vectors.json")


def load_vectors():
    with open(VECTORS) as fh:
        return json.load(fh)["vectors"]


def b(s):
    return s.encode("utf-8") if isinstance(s, str) else s


class Unsupported(Exception):
    """The backend declares this call out of scope (not a failure)."""


def evaluate(backend, v):
    """Run vector v on backend; return the list of failed checks (strings)."""
    op = v["op"]
    text = b(v["text"])
    pat = b(v["pattern"]) if v.get("pattern") is not None else None
    start = v.get("start", 0)
    kind_result = None
    if op in ("match_first", "search", "ct_match_first", "ct_search", "nfa_match_first",
              "obj_match_first", "obj_match_next", "dfa_match_first", "dfa_match_next"):
        kind_result = "opt"
        if op == "match_first":
            r = backend.match_first(pat, text)
        elif op == "search":
            r = backend.search(pat, text)
        elif op == "nfa_match_first":
            r = backend.nfa_match_first(pat, text)
        elif op == "ct_match_first":
            r = backend.ct_match_first(pat, text)
        elif op == "ct_search":
            r = backend.ct_search(pat, text)
        elif op == "obj_match_first":
            r = backend.obj_match_first(pat, text, start)
        elif op == "obj_match_next":
            r = backend.obj_match_next(pat, text, start)
        elif op == "dfa_match_first":
            r = backend.dfa_match_first(v["engine"], text, start)
        else:
            r = backend.dfa_match_next(v["engine"], text, start)
    elif op in ("findall", "ct_findall", "obj_match_all", "dfa_match_all", "nfa_findall"):
        kind_result = "list"
        if op == "findall":
            r = backend.findall(pat, text)
        elif op == "nfa_findall":
            r = backend.nfa_findall(pat, text)
        elif op == "ct_findall":
            r = backend.ct_findall(pat, text)
        elif op == "obj_match_all":
            r = backend.findall(pat, text)
        else:
            r = backend.dfa_match_all(v["engine"], text)
    elif op == "sub":
        kind_result = "str"
        r = backend.sub(pat, b(v["repl"]), text, v.get("count", 0))
    elif op == "split":
        kind_result = "strlist"
        r = backend.split(pat, text, v.get("maxsplit", 0))
    elif op == "obj_test":
        kind_result = "bool"
        r = backend.obj_test(pat, text)
    elif op == "obj_is_match":
        kind_result = "bool"
        r = backend.obj_is_match(pat, text, start)
    elif op == "obj_engine_type":
        kind_result = "scalar"
        r = backend.engine_type(pat)
    elif op == "obj_stats":
        kind_result = "scalar"
        r = backend.stats(pat)
    else:
        raise Unsupported("unknown op " + op)

    fails = []
    for c in v["checks"]:
        k = c["kind"]
        want = c["value"]
        idx = c.get("index")
        got = None
        try:
            if k in ("matched", "bool"):
                got = (r is not None) if kind_result == "opt" else bool(r)
            elif k == "count":
                got = len(r)
            elif k in ("start", "end", "text", "length"):
                m = r if kind_result == "opt" else r[idx]
                if m is None:
                    got = None
                elif k == "start":
                    got = m[0]
                elif k == "end":
                    got = m[1]
                elif k == "length":
                    got = m[1] - m[0]
                else:
                    got = text[m[0]:m[1]]
                    want = b(want)
            elif k == "result":
                got, want = r, b(want)
            elif k == "part":
                got, want = r[idx], b(want)
            elif k in ("engine_type", "stats"):
                got = r
            elif k == "group_id":
                got = 0
            else:
                raise Unsupported("unknown check " + k)
        except IndexError:
            got = "<index %s out of range, %d results>" % (idx, len(r))
        if c.get("how") == "contains":
            ok = isinstance(got, (str, bytes)) and (want in got)
        else:
            ok = got == want
        if c.get("negate"):
            ok = not ok
        if not ok:
            fails.append("%s:%d %s(%r, %r): %s[%s] want %r got %r" % (
                v["file"], v["line"], op, v.get("pattern") or v.get("engine"),
                v["text"], k, idx, want, got))
    return fails
