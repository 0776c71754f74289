#!/usr/bin/env python3
"""Transcribe the AST-shape assertions of the reference's parser tests into data.

Reads /root/reference/tests/test_parser.mojo AS TEXT, rewrites every test body into Python
syntax (same trick as extract_reference_vectors.py) and executes it statement by statement in
a sandbox in which `parse(pattern)` returns a symbolic node: `.get_child(i)` extends a path,
`.type / .min / .max / .positive_logic / .group_id / .get_children_len() / .get_value().value()`
yield symbolic terms, and `assert_equal / assert_true / assert_false` on such a term record
  {pattern, path, attr, want, file, line}.
No reference code runs; assertions this model does not understand (is_match, raises, ...) are
skipped.  Output: tests/golden/ast_vectors.json.
"""
import ast
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from extract_reference_vectors import REF, Rewrite, mojo_body_to_python  # noqa: E402

OUT = os.path.join(HERE, "ast_vectors.json")
FILE = "tests/test_parser.mojo"
TYPES = ["RE", "ELEMENT", "WILDCARD", "SPACE", "DIGIT", "WORD", "RANGE", "START", "END", "OR", "NOT", "GROUP"]


class Skip(Exception):
    pass


class Term:
    def __init__(self, node, attr):
        self.node, self.attr = node, attr


class ValueP:
    def __init__(self, node):
        self.node = node

    def value(self):
        return Term(self.node, "value")

    def __bool__(self):
        raise Skip("truth of an optional")


class NodeP:
    def __init__(self, pattern, path):
        self.pattern, self.path = pattern, path

    def get_child(self, i):
        if not isinstance(i, int):
            raise Skip("symbolic child index")
        return NodeP(self.pattern, self.path + [i])

    def get_children_len(self):
        return Term(self, "children_len")

    def get_value(self):
        return ValueP(self)

    def __getattr__(self, name):
        if name in ("type", "min", "max", "positive_logic", "group_id"):
            return Term(self, name)
        raise Skip("attribute %s" % name)

    def __bool__(self):
        raise Skip("truth of a node")


def main():
    src = open(os.path.join(REF, FILE), encoding="utf-8").read().split("\n")
    starts = [i for i, l in enumerate(src) if re.match(r"^def test_\w+\(", l)]
    starts.append(len(src))
    vectors = []
    stats = {"tests": 0, "skipped_stmts": 0}
    for k in range(len(starts) - 1):
        a, b = starts[k], starts[k + 1]
        name = re.match(r"^def (test_\w+)\(", src[a]).group(1)
        body = []
        for i in range(a + 1, b):
            l = src[i]
            if l and not l.startswith((" ", "\t")) and l.strip():
                break
            body.append(l)
        try:
            tree = ast.parse("def _t():\n" + mojo_body_to_python(body) + "\n")
        except SyntaxError:
            continue
        tree = Rewrite().visit(tree)
        ast.fix_missing_locations(tree)
        stats["tests"] += 1
        cur = {"line": 0}

        def record(term, want):
            if isinstance(want, Term) or not isinstance(term, Term):
                raise Skip("not (term, constant)")
            vectors.append({"file": FILE, "line": cur["line"], "test": name, "pattern": term.node.pattern,
                            "path": term.node.path, "attr": term.attr, "want": want})

        def assert_equal(x, y, *r, **kw):
            if isinstance(y, Term) and not isinstance(x, Term):
                x, y = y, x
            record(x, y)

        def assert_true(x, *r, **kw):
            record(x, True)

        def assert_false(x, *r, **kw):
            record(x, False)

        def parse(p):
            if not isinstance(p, str):
                raise Skip("non-literal pattern")
            return NodeP(p, [])

        env = {"parse": parse, "assert_equal": assert_equal, "assert_true": assert_true,
               "assert_false": assert_false, "True": True, "False": False, "__builtins__": {}}
        for t in TYPES:
            env[t] = t
        # helper names the Rewrite pass may introduce: unsupported here
        for n in ("NOT", "LEN", "BOOL", "IN", "EQ"):
            env.setdefault(n, lambda *a, **k: (_ for _ in ()).throw(Skip("rewritten helper")))
        env["NOT"] = "NOT"
        for stmt in tree.body[0].body:
            cur["line"] = a + stmt.lineno
            try:
                exec(compile(ast.Module([stmt], []), FILE, "exec"), env)
            except Exception:
                stats["skipped_stmts"] += 1
                for nn in ast.walk(stmt):
                    if isinstance(nn, ast.Name) and isinstance(nn.ctx, ast.Store):
                        env.pop(nn.id, None)
    doc = {"source": "msaelices/mojo-regex v0.21.0 tests/test_parser.mojo (transcribed as data)",
           "stats": dict(stats, vectors=len(vectors)), "vectors": vectors}
    json.dump(doc, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps(doc["stats"]))


if __name__ == "__main__":
    main()
