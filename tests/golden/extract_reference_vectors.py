#!/usr/bin/env python3
"""Transcribe the reference's own known-answer tests into a data fixture.

Reads the reference's ``tests/test_*.mojo`` files AS TEXT (the reference cannot
be executed here: no Mojo toolchain) and records, for every call of the regex
API inside a test, the inputs (pattern, text, arguments) and every value the
test asserts about the result (match / no match, start, end, matched text,
number of matches, replaced string ...).  Output: ``reference_vectors.json`` --
data only (inputs + expected outputs + file:line), no reference source text.

How: the test bodies are straight-line Python-like code.  Each ``def test_*``
body is rewritten to Python syntax, parsed with ``ast`` and run in an empty
sandbox namespace in which the regex API names are *recorders*: they return
symbolic result objects, and ``assert_equal/assert_true/assert_false`` record
what is asserted about them.  Nothing from the reference is imported or run.

Usage:  python tests/golden/extract_reference_vectors.py [/root/reference]
"""
from __future__ import annotations

import ast
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")

# files whose tests exercise the hot path through the hybrid/DFA/comptime API, plus
# tests/test_nfa.mojo: its `match_first` / `findall` are regex.nfa's (NFAEngine driven
# directly, nfa.mojo:1733-1769), recorded under the distinct ops `nfa_match_first` /
# `nfa_findall` -- they pin the backtracking matcher (oracle/mrx_ref/backtrack.py and
# the product's flat program), whatever the hybrid router would do with the pattern.
FILES = [
    "tests/test_matcher.mojo",
    "tests/test_dfa.mojo",
    "tests/test_comptime_regex.mojo",
    "tests/test_predefined_classes.mojo",
    "tests/test_split.mojo",
    "tests/test_simd.mojo",
    "tests/test_nfa.mojo",
]


# ----------------------------------------------------------------------------
# symbolic values
# ----------------------------------------------------------------------------
class Vec:
    """One recorded API call plus the checks a test makes on its result."""

    def __init__(self, rec, op, pattern, text, **extra):
        self.d = {"file": rec.file, "line": rec.cur_line, "test": rec.test, "op": op,
                  "pattern": pattern, "text": text}
        self.d.update(extra)
        self.d["checks"] = []
        rec.vectors.append(self.d)

    def check(self, **c):
        if c not in self.d["checks"]:
            self.d["checks"].append(c)


class Expr:
    """A symbolic scalar derived from a result (start, end, text, len ...)."""

    def __init__(self, vec, kind, index=None):
        self.vec, self.kind, self.index = vec, kind, index

    def __sub__(self, other):
        if isinstance(other, Expr) and self.kind == "end" and other.kind == "start" \
                and self.vec is other.vec and self.index == other.index:
            return Expr(self.vec, "length", self.index)
        raise Skip("unsupported arithmetic")

    def __bool__(self):
        raise Skip("symbolic value used as bool")


class BoolE:
    def __init__(self, vec, neg=False, index=None):
        self.vec, self.neg, self.index = vec, neg, index

    def __bool__(self):
        raise Skip("symbolic bool used in control flow")


class MatchP:
    def __init__(self, vec, index=None):
        self._vec, self._index = vec, index
        self.start_idx = Expr(vec, "start", index)
        self.end_idx = Expr(vec, "end", index)
        self.group_id = Expr(vec, "group_id", index)

    def get_match_text(self):
        return Expr(self._vec, "text", self._index)


class OptMatch:
    def __init__(self, vec):
        self._vec = vec

    def value(self):
        return MatchP(self._vec)

    def __bool__(self):
        raise Skip("symbolic Optional used in control flow")


class MatchListP:
    def __init__(self, vec):
        self._vec = vec

    def __getitem__(self, i):
        if not isinstance(i, int):
            raise Skip("non-constant index")
        return MatchP(self._vec, i)


class StrListP:
    def __init__(self, vec):
        self._vec = vec

    def __getitem__(self, i):
        return Expr(self._vec, "part", i)


class Skip(Exception):
    pass


def _s(x):
    if isinstance(x, str):
        return x
    raise Skip("non-literal string argument: %r" % (x,))


# ----------------------------------------------------------------------------
# recorder environment
# ----------------------------------------------------------------------------
class Recorder:
    def __init__(self):
        self.vectors = []
        self.file = ""
        self.test = ""
        self.cur_line = 0

    # ---- module-level API (regex.matcher / regex) ---------------------------
    def api(self, prefix=""):
        rec = self

        def match_first(p, t):
            return OptMatch(Vec(rec, prefix + "match_first", _s(p), _s(t)))

        def search(p, t):
            return OptMatch(Vec(rec, prefix + "search", _s(p), _s(t)))

        def findall(p, t):
            return MatchListP(Vec(rec, prefix + "findall", _s(p), _s(t)))

        def split(p, t, maxsplit=0):
            return StrListP(Vec(rec, "split", _s(p), _s(t), maxsplit=int(maxsplit)))

        def sub(p, r, t, count=0):
            return Expr(Vec(rec, "sub", _s(p), _s(t), repl=_s(r), count=int(count)), "result")

        return dict(match_first=match_first, search=search, findall=findall,
                    split=split, sub=sub)


class Comptime:
    """``search["pat"]("text")`` of regex.comptime_regex."""

    def __init__(self, rec, op):
        self.rec, self.op = rec, op

    def __getitem__(self, pat):
        rec, op = self.rec, self.op

        def call(t):
            v = Vec(rec, "ct_" + op, _s(pat), _s(t))
            return MatchListP(v) if op == "findall" else OptMatch(v)
        return call


class CompiledP:
    """compile_regex(p) / CompiledRegex(p) / HybridMatcher(p) objects."""

    def __init__(self, rec, pattern):
        self.rec, self.pattern = rec, _s(pattern)

    def match_first(self, t, start=0):
        return OptMatch(Vec(self.rec, "obj_match_first", self.pattern, _s(t), start=int(start)))

    def match_next(self, t, start=0):
        return OptMatch(Vec(self.rec, "obj_match_next", self.pattern, _s(t), start=int(start)))

    def match_all(self, t):
        return MatchListP(Vec(self.rec, "obj_match_all", self.pattern, _s(t)))

    def test(self, t):
        return BoolE(Vec(self.rec, "obj_test", self.pattern, _s(t)))

    def is_match(self, t, start=0):
        return BoolE(Vec(self.rec, "obj_is_match", self.pattern, _s(t), start=int(start)))

    def sub(self, r, t, count=0):
        return Expr(Vec(self.rec, "sub", self.pattern, _s(t), repl=_s(r), count=int(count)), "result")

    def get_stats(self):
        return Expr(Vec(self.rec, "obj_stats", self.pattern, ""), "stats")

    def get_engine_type(self):
        return Expr(Vec(self.rec, "obj_engine_type", self.pattern, ""), "engine_type")

    def get_complexity(self):
        raise Skip("complexity object")


class DFAEngineP:
    """regex.dfa.DFAEngine driven directly (tests/test_dfa.mojo)."""

    def __init__(self, rec, build=None):
        self.rec, self.build = rec, build

    def compile_pattern(self, lit, hs, he):
        self.build = {"kind": "literal", "literal": _s(lit), "start_anchor": bool(hs),
                      "end_anchor": bool(he)}

    def compile_character_class(self, cc, mn, mx):
        self.build = {"kind": "char_class", "char_class": _s(cc), "min": int(mn), "max": int(mx)}

    def _v(self, op, t, **kw):
        if self.build is None:
            raise Skip("engine not built")
        return Vec(self.rec, op, None, _s(t), engine=self.build, **kw)

    def match_first(self, t, start=0):
        return OptMatch(self._v("dfa_match_first", t, start=int(start)))

    def match_next(self, t, start=0):
        return OptMatch(self._v("dfa_match_next", t, start=int(start)))

    def match_all(self, t):
        return MatchListP(self._v("dfa_match_all", t))


class AstP:
    def __init__(self, pattern):
        self.pattern = _s(pattern)


def make_env(rec: Recorder):
    env = {}
    env.update(rec.api())
    env["compile_regex"] = lambda p: CompiledP(rec, p)
    env["CompiledRegex"] = lambda p: CompiledP(rec, p)
    env["HybridMatcher"] = lambda p: CompiledP(rec, p)
    env["DFAEngine"] = lambda: DFAEngineP(rec)
    env["parse"] = lambda p: AstP(p)
    env["compile_dfa_pattern"] = lambda a: DFAEngineP(rec, {"kind": "pattern", "pattern": a.pattern})
    def String(x=""):
        if isinstance(x, bool) or not isinstance(x, (str, int)):
            raise Skip("String() of unsupported value")
        return str(x)
    env["String"] = String
    env["chr"] = chr
    env["ord"] = ord
    env["range"] = range
    env["True"] = True
    env["False"] = False

    def assert_true(x, *a, **k):
        _bool_check(x, True)

    def assert_false(x, *a, **k):
        _bool_check(x, False)

    def _bool_check(x, want):
        if isinstance(x, OptMatch):
            x._vec.check(kind="matched", value=want)
        elif isinstance(x, BoolE):
            x.vec.check(kind="matched" if x.vec.d["op"] not in ("obj_test", "obj_is_match")
                        else "bool", value=(want != x.neg), **({} if x.index is None else {}))
        elif isinstance(x, CmpE):
            x.record(want)
        else:
            raise Skip("assert on non-symbolic value")

    def assert_equal(a, b, *rest, **k):
        if isinstance(b, Expr) and not isinstance(a, Expr):
            a, b = b, a
        if not isinstance(a, Expr):
            raise Skip("assert_equal on non-symbolic value")
        if isinstance(b, (Expr, BoolE, OptMatch, MatchListP)):
            raise Skip("assert_equal between two symbolic values")
        if isinstance(b, bool) or not isinstance(b, (int, str)):
            raise Skip("assert_equal with unsupported literal")
        c = {"kind": a.kind, "value": b}
        if a.index is not None:
            c["index"] = a.index
        a.vec.check(**c)

    env["assert_true"] = assert_true
    env["assert_false"] = assert_false
    env["assert_equal"] = assert_equal

    # helpers the AST rewrite introduces
    def NOT(x):
        if isinstance(x, OptMatch):
            return BoolE(x._vec, neg=True)
        if isinstance(x, BoolE):
            return BoolE(x.vec, neg=not x.neg)
        if isinstance(x, bool):
            return not x
        raise Skip("not on unsupported value")

    def LEN(x):
        if isinstance(x, MatchListP):
            return Expr(x._vec, "count")
        if isinstance(x, StrListP):
            return Expr(x._vec, "count")
        if isinstance(x, str):
            return len(x)
        raise Skip("len of unsupported value")

    def BOOL(x):  # x.__bool__()
        if isinstance(x, OptMatch):
            return BoolE(x._vec)
        if isinstance(x, BoolE):
            return x
        raise Skip("__bool__ of unsupported value")

    def IN(needle, hay):  # "DFA" in stats
        if isinstance(hay, Expr) and isinstance(needle, str):
            return CmpE(hay, "contains", needle)
        raise Skip("unsupported 'in'")

    def EQ(a, b):
        if isinstance(a, Expr) and isinstance(b, (int, str)) and not isinstance(b, bool):
            return CmpE(a, "eq", b)
        if isinstance(a, (int, str)) and isinstance(b, (int, str)):
            return a == b
        raise Skip("unsupported ==")

    env.update(NOT=NOT, LEN=LEN, BOOL=BOOL, IN=IN, EQ=EQ)
    return env


class CmpE:
    def __init__(self, expr, how, value):
        self.expr, self.how, self.value = expr, how, value

    def record(self, want):
        c = {"kind": self.expr.kind, "value": self.value}
        if self.how == "contains":
            c["how"] = "contains"
        if not want:
            c["negate"] = True
        if self.expr.index is not None:
            c["index"] = self.expr.index
        self.expr.vec.check(**c)

    def __bool__(self):
        raise Skip("symbolic comparison in control flow")


# ----------------------------------------------------------------------------
# Mojo test body -> Python AST
# ----------------------------------------------------------------------------
class Rewrite(ast.NodeTransformer):
    def visit_UnaryOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Not):
            return ast.copy_location(
                ast.Call(ast.Name("NOT", ast.Load()), [node.operand], []), node)
        return node

    def visit_Call(self, node):
        self.generic_visit(node)
        if isinstance(node.func, ast.Name) and node.func.id == "len":
            node.func.id = "LEN"
        if (isinstance(node.func, ast.Attribute) and node.func.attr == "__bool__"
                and not node.args):
            return ast.copy_location(
                ast.Call(ast.Name("BOOL", ast.Load()), [node.func.value], []), node)
        return node

    def visit_Compare(self, node):
        self.generic_visit(node)
        if len(node.ops) == 1 and isinstance(node.ops[0], ast.In):
            return ast.copy_location(
                ast.Call(ast.Name("IN", ast.Load()), [node.left, node.comparators[0]], []), node)
        if len(node.ops) == 1 and isinstance(node.ops[0], ast.Eq):
            return ast.copy_location(
                ast.Call(ast.Name("EQ", ast.Load()), [node.left, node.comparators[0]], []), node)
        return node


def mojo_body_to_python(lines):
    out = []
    for ln in lines:
        ln = re.sub(r"^(\s*)(var|ref)\s+", r"\1", ln)
        ln = re.sub(r"^(\s*)comptime\s+", r"\1", ln)
        # drop simple type annotations on assignments:  x: Type = ...
        ln = re.sub(r"^(\s*)(\w+)\s*:\s*[\w\[\], .]+=\s", r"\1\2 = ", ln)
        out.append(ln)
    return "\n".join(out)


def run_file(rec: Recorder, relpath: str, stats: dict):
    path = os.path.join(REF, relpath)
    src = open(path, encoding="utf-8").read().split("\n")
    rec.file = relpath
    # locate test functions
    starts = [i for i, l in enumerate(src) if re.match(r"^def test_\w+\(", l)]
    starts.append(len(src))
    for k in range(len(starts) - 1):
        a, b = starts[k], starts[k + 1]
        name = re.match(r"^def (test_\w+)\(", src[a]).group(1)
        body = []
        for i in range(a + 1, b):
            l = src[i]
            if l and not l.startswith((" ", "\t")) and l.strip():
                break  # dedent: end of function
            body.append(l)
        py = "def _t():\n" + mojo_body_to_python(body) + "\n"
        stats["tests"] += 1
        try:
            tree = ast.parse(py)
        except SyntaxError:
            stats["unparsed"] += 1
            continue
        tree = Rewrite().visit(tree)
        ast.fix_missing_locations(tree)
        rec.test = name
        fn = tree.body[0]
        env = make_env(rec)
        if relpath.endswith("test_nfa.mojo"):
            env.update(rec.api("nfa_"))   # `from regex.nfa import match_first, findall`
        if relpath.endswith("test_comptime_regex.mojo"):
            for op in ("search", "match_first", "findall"):
                env[op] = Comptime(rec, op)
        env["__builtins__"] = {}
        n_before = len(rec.vectors)
        # execute statement by statement so one unsupported statement does not
        # lose the rest of the test
        for stmt in fn.body:
            rec.cur_line = a + stmt.lineno  # 1-based line in the .mojo file
            mod = ast.Module([stmt], [])
            try:
                exec(compile(mod, relpath, "exec"), env)
            except Exception:  # Skip, NameError, ...: statement not understood
                stats["skipped_stmts"] += 1
                # any name this statement (re)binds is now unreliable: drop it so
                # later uses are skipped too instead of recording stale inputs
                for n in ast.walk(stmt):
                    if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
                        env.pop(n.id, None)
        if len(rec.vectors) > n_before:
            stats["tests_with_vectors"] += 1


def main():
    rec = Recorder()
    stats = {"tests": 0, "unparsed": 0, "skipped_stmts": 0, "tests_with_vectors": 0}
    for f in FILES:
        run_file(rec, f, stats)
    # keep only calls about which the test asserts something
    vecs = [v for v in rec.vectors if v["checks"]]
    doc = {
        "source": "msaelices/mojo-regex v0.21.0 tests/*.mojo (transcribed as data)",
        "note": "file/line point at the reference test statement that makes the call",
        "stats": dict(stats, vectors=len(vecs)),
        "vectors": vecs,
    }
    with open(OUT, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print(json.dumps(doc["stats"]))


if __name__ == "__main__":
    main()
