"""ORACLE (test infrastructure) -- ctypes bridge to oracle/c/mrx_backtrack.c.

The C file restates NFAEngine's control flow (nfa.mojo:169-1443); the byte predicates come from
backtrack.py's own functions as three 256-entry tables per leaf, so membership cannot drift between the
two halves.  CBacktrack has BacktrackNFA's interface; backtrack.py stays the pinned restatement (it is what
the reference's vectors run against), this one exists to make kilobyte texts affordable in the fuzz.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

from . import frontend as F
from .backtrack import (BacktrackNFA, is_match_char, range_first_test, simd_predicate, _is_digit, _is_word,
                        _is_space5)

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.dirname(_HERE)
_SO = os.path.join(_ORACLE_DIR, "_build", "libmrx_backtrack.so")
_lib = None

_TYPES = {F.RE: 0, F.ELEMENT: 1, F.WILDCARD: 2, F.SPACE: 3, F.DIGIT: 4, F.WORD: 5, F.RANGE: 6, F.START: 7, F.END: 8,
          F.OR: 9, F.GROUP: 10}
_LEAVES = (F.ELEMENT, F.WILDCARD, F.SPACE, F.DIGIT, F.WORD, F.RANGE)


class _Node(C.Structure):
    _fields_ = [("type", C.c_int32), ("min", C.c_int32), ("max", C.c_int32), ("capturing", C.c_int32),
                ("group_id", C.c_int32), ("value_len", C.c_int32), ("nchildren", C.c_int32), ("child0", C.c_int32),
                ("tbl", C.c_int32)]


class _Prog(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("kids", C.c_void_p), ("tables", C.c_void_p), ("root", C.c_int32),
                ("pattern", C.c_char_p), ("pattern_len", C.c_int32), ("literal", C.c_char_p), ("literal_len", C.c_int32),
                ("has_literal_optimization", C.c_int32), ("ends_with_dotstar", C.c_int32),
                ("starts_with_dotstar", C.c_int32), ("is_prefix_literal", C.c_int32)]


class _Group(C.Structure):
    _fields_ = [("gid", C.c_int32), ("start", C.c_int32), ("end", C.c_int32)]


class _Groups(C.Structure):
    _fields_ = [("v", C.POINTER(_Group)), ("n", C.c_int64), ("cap", C.c_int64)]


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_ORACLE_DIR, "c", "mrx_backtrack.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)
        lib = C.CDLL(_SO)
        lib.mrx_bt_match_first.restype = C.c_int
        lib.mrx_bt_match_first.argtypes = [C.POINTER(_Prog), C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        lib.mrx_bt_match_next.restype = C.c_int
        lib.mrx_bt_match_next.argtypes = [C.POINTER(_Prog), C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64), C.POINTER(_Groups)]
        lib.mrx_bt_match_all.restype = C.c_int64
        lib.mrx_bt_match_all.argtypes = [C.POINTER(_Prog), C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
        lib.mrx_bt_groups_free.argtypes = [C.POINTER(_Groups)]
        assert lib.mrx_bt_node_size() == C.sizeof(_Node) and lib.mrx_bt_prog_size() == C.sizeof(_Prog)
        _lib = lib
    return _lib


def _first_test(node: F.Node, ch: int) -> bool:
    """The test each leaf matcher makes on the byte at str_i (nfa.mojo:757-995)."""
    t = node.type
    if t == F.ELEMENT:
        v = node.get_value()
        return bool(v) and v[0] == ch
    if t == F.WILDCARD:
        return ch != 0x0A
    if t == F.SPACE:
        return _is_space5(ch)
    if t == F.DIGIT:
        return _is_digit(ch)
    if t == F.WORD:
        return _is_word(ch)
    return range_first_test(node, ch)


class CBacktrack:
    """BacktrackNFA's interface on the C matcher."""

    def __init__(self, py: BacktrackNFA):
        self.py = py
        self._lib = load()
        nodes: List[_Node] = []
        kids: List[int] = []
        tables = bytearray()
        ntbl = 0

        def add(node: F.Node) -> int:
            nonlocal ntbl
            idx = len(nodes)
            nd = _Node()
            nodes.append(nd)
            nd.type = _TYPES[node.type]
            nd.min, nd.max = node.min, node.max
            nd.capturing = 1 if node.capturing_group else 0
            nd.group_id = node.group_id
            v = node.get_value()
            nd.value_len = len(v) if v else 0
            nd.tbl = -1
            if node.type in _LEAVES:
                nd.tbl = ntbl
                ntbl += 1
                sp = simd_predicate(node)
                tables.extend(bytes(1 if _first_test(node, c) else 0 for c in range(256)))
                tables.extend(bytes(1 if is_match_char(node, c) else 0 for c in range(256)))
                tables.extend(bytes(1 if (sp is not None and sp(c)) else 0 for c in range(256)))
            ch = [add(node.get_child(i)) for i in range(node.get_children_len())]
            nd.nchildren = len(ch)
            nd.child0 = len(kids)
            kids.extend(ch)
            return idx

        root = add(py.regex) if py.regex is not None else -1
        self._nodes = (_Node * max(1, len(nodes)))(*nodes)
        self._kids = (C.c_int32 * max(1, len(kids)))(*kids)
        self._tables = (C.c_uint8 * max(1, len(tables))).from_buffer_copy(bytes(tables) or b"\0")
        self._pattern = bytes(py.pattern)
        self._literal = bytes(py.literal_prefix)
        p = _Prog()
        p.nodes = C.addressof(self._nodes)
        p.kids = C.addressof(self._kids)
        p.tables = C.addressof(self._tables)
        p.root = root
        p.pattern, p.pattern_len = self._pattern, len(self._pattern)
        p.literal, p.literal_len = self._literal, len(self._literal)
        p.has_literal_optimization = 1 if py.has_literal_optimization else 0
        p.ends_with_dotstar = 1 if py.ends_with_dotstar else 0
        p.starts_with_dotstar = 1 if py.starts_with_dotstar else 0
        p.is_prefix_literal = 1 if py.is_prefix_literal else 0
        self._prog = p

    # BacktrackNFA's attributes that callers read
    def __getattr__(self, name):
        return getattr(self.py, name)

    def match_first(self, text: bytes, start: int = 0) -> Optional[Tuple[int, int]]:
        end = C.c_int64(0)
        ok = self._lib.mrx_bt_match_first(C.byref(self._prog), text, len(text), start, C.byref(end))
        return (start, end.value) if ok else None

    def match_next(self, text: bytes, start: int = 0) -> Optional[Tuple[int, int]]:
        ms, me = C.c_int64(0), C.c_int64(0)
        ok = self._lib.mrx_bt_match_next(C.byref(self._prog), text, len(text), start, C.byref(ms), C.byref(me), None)
        return (ms.value, me.value) if ok else None

    def match_next_with_groups(self, text: bytes, start: int = 0):
        ms, me = C.c_int64(0), C.c_int64(0)
        g = _Groups()
        ok = self._lib.mrx_bt_match_next(C.byref(self._prog), text, len(text), start, C.byref(ms), C.byref(me), C.byref(g))
        groups = [(g.v[k].gid, g.v[k].start, g.v[k].end) for k in range(g.n)] if ok else []
        self._lib.mrx_bt_groups_free(C.byref(g))
        return ((ms.value, me.value), groups) if ok else (None, [])

    def match_all(self, text: bytes) -> List[Tuple[int, int]]:
        cap = len(text) + 2
        buf = (C.c_int32 * (2 * cap))()
        n = self._lib.mrx_bt_match_all(C.byref(self._prog), text, len(text), buf, cap)
        assert n <= cap
        return [(buf[2 * k], buf[2 * k + 1]) for k in range(n)]
