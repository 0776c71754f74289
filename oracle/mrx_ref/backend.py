"""ORACLE (test infrastructure) -- adapter exposing the oracle through the
backend interface tests/vector_eval.py drives (also mirrors the comptime API
of src/regex/comptime_regex.mojo:158-233 and direct DFAEngine use as in the
reference's tests/test_dfa.mojo)."""
from __future__ import annotations

from . import hybrid
from .frontend import parse
from .dfa_engine import DFAEngine, compile_dfa_pattern, DFACompileError


class OracleBackend:
    name = "oracle"

    def match_first(self, p, t):
        return hybrid.match_first(p, t)

    def search(self, p, t):
        return hybrid.search(p, t)

    def findall(self, p, t):
        return hybrid.findall(p, t)

    def sub(self, p, r, t, count=0):
        return hybrid.sub(p, r, t, count)

    def split(self, p, t, maxsplit=0):
        return hybrid.split(p, t, maxsplit)

    # regex.nfa's module functions: NFAEngine driven directly (tests/test_nfa.mojo)
    def nfa_match_first(self, p, t):
        return hybrid.nfa_match_first(p, t)

    def nfa_findall(self, p, t):
        return hybrid.nfa_findall(p, t)

    def obj_match_first(self, p, t, start=0):
        return hybrid.compile_regex(p).match_first(t, start)

    def obj_match_next(self, p, t, start=0):
        return hybrid.compile_regex(p).match_next(t, start)

    def obj_test(self, p, t):
        return hybrid.compile_regex(p).test(t)

    def obj_is_match(self, p, t, start=0):
        return hybrid.compile_regex(p).is_match(t, start)

    def engine_type(self, p):
        return hybrid.compile_regex(p).matcher.get_engine_type()

    def stats(self, p):
        return hybrid.compile_regex(p).get_stats()

    # comptime API: DFAEngine straight from compile_dfa_pattern, runtime API
    # when that raises (comptime_regex.mojo:59-87, 176-233)
    def _ct_engine(self, p):
        try:
            return compile_dfa_pattern(parse(p))
        except DFACompileError:
            return None

    def ct_search(self, p, t):
        e = self._ct_engine(p)
        return e.match_next(t) if e is not None else hybrid.search(p, t)

    def ct_match_first(self, p, t):
        e = self._ct_engine(p)
        if e is None:
            return hybrid.match_first(p, t)
        r = e.match_first(t, 0)
        return r if (r is not None and r[0] == 0) else None

    def ct_findall(self, p, t):
        e = self._ct_engine(p)
        return e.match_all(t) if e is not None else hybrid.findall(p, t)

    # direct DFAEngine use
    def _dfa(self, build):
        if build["kind"] == "literal":
            e = DFAEngine()
            e.compile_pattern(build["literal"].encode(), build["start_anchor"],
                              build["end_anchor"])
            return e
        if build["kind"] == "char_class":
            e = DFAEngine()
            e.compile_character_class_with_logic(build["char_class"].encode(),
                                                 build["min"], build["max"], True)
            return e
        return compile_dfa_pattern(parse(build["pattern"].encode()))

    def dfa_match_first(self, build, t, start=0):
        return self._dfa(build).match_first(t, start)

    def dfa_match_next(self, build, t, start=0):
        return self._dfa(build).match_next(t, start)

    def dfa_match_all(self, build, t):
        return self._dfa(build).match_all(t)
