"""ORACLE (test infrastructure) -- OnePass NFA.

Restates src/regex/onepass.mojo: _epsilon_close (:64-110), compile_onepass (:180-355,
subset construction with the one-pass ambiguity check), the per-state `$` fixup flag
(:122-140, 303-307) and OnePassNFA.match_first / match_next / match_all (:440-553).
The reference consults it only from NFAMatcher.match_first for programs that contain
OP_END_ANCHOR (matcher.mojo:310-313, 378-379).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

from .pikevm import (Program, OP_BYTE, OP_RANGE, OP_CLASS, OP_ANY, OP_SPLIT, OP_JUMP, OP_MATCH,
                     OP_START_ANCHOR, OP_END_ANCHOR, MAX_STATES)

ONEPASS_DEAD = -1
ONEPASS_MAX_STATES = 512  # onepass.mojo:56


def _epsilon_close(program: Program, start_pcs: List[int], at_start: bool, at_end: bool = False):
    """onepass.mojo:64-110.  Every visited pc is marked (SPLIT / JUMP / anchors included);
    byte-consuming ops and MATCH are retained and not advanced past."""
    n = len(program)
    result = [0] * n
    stack = list(start_pcs)
    while stack:
        pc = stack.pop()
        if pc < 0 or pc >= n or result[pc]:
            continue
        result[pc] = 1
        op, a0, a1 = program.instructions[pc]
        if op == OP_SPLIT:
            stack.append(a0)
            stack.append(a1)
        elif op == OP_JUMP:
            stack.append(a0)
        elif op == OP_START_ANCHOR:
            if at_start:
                stack.append(pc + 1)
        elif op == OP_END_ANCHOR:
            if at_end:
                stack.append(pc + 1)
    return tuple(result)


def _set_contains_match(program: Program, nfa_set) -> bool:
    return any(nfa_set[pc] and program.instructions[pc][0] == OP_MATCH for pc in range(len(program)))


def _closure_reaches_match_with_end_anchor(program: Program, nfa_set) -> bool:
    """onepass.mojo:122-140."""
    pcs = [pc for pc in range(len(program)) if nfa_set[pc]]
    end_set = _epsilon_close(program, pcs, at_start=False, at_end=True)
    return _set_contains_match(program, end_set)


def _fires(program: Program, pc: int, byte: int) -> bool:
    op, a0, a1 = program.instructions[pc]
    if op == OP_BYTE:
        return byte == a0
    if op == OP_CLASS:
        return program.class_tables[a0][byte] != 0
    if op == OP_ANY:
        return byte != 10
    if op == OP_RANGE:
        return a0 <= byte <= a1
    return False


class OnePassNFA:
    """onepass.mojo:398-553."""

    def __init__(self, transitions, is_match, is_end_match, has_start_anchor, has_end_anchor):
        self.transitions = transitions
        self.is_match_flags = is_match
        self.is_end_match_flags = is_end_match
        self.has_start_anchor = has_start_anchor
        self.has_end_anchor = has_end_anchor

    def match_first(self, text: bytes, start: int = 0) -> Optional[Tuple[int, int]]:
        """onepass.mojo:440-488."""
        if self.has_start_anchor and start > 0:
            return None
        n = len(text)
        state = 0
        match_end = -1
        if self.is_match_flags[0]:
            match_end = start
        pos = start
        while pos < n:
            nxt = self.transitions[state][text[pos]]
            if nxt == ONEPASS_DEAD:
                break
            state = nxt
            pos += 1
            if self.is_match_flags[state]:
                match_end = pos
        if self.has_end_anchor and pos == n:
            if self.is_end_match_flags[state]:
                match_end = pos
        if match_end >= 0:
            return (start, match_end)
        return None

    def match_next(self, text: bytes, start: int = 0):
        """onepass.mojo:490-507."""
        if self.has_start_anchor:
            if start > 0:
                return None
            return self.match_first(text, 0)
        for p in range(start, len(text) + 1):
            m = self.match_first(text, p)
            if m is not None:
                return m
        return None

    def match_all(self, text: bytes):
        """onepass.mojo:509-553."""
        out = []
        if self.has_start_anchor:
            m = self.match_first(text, 0)
            if m is not None:
                out.append(m)
            return out
        pos = 0
        n = len(text)
        while pos <= n:
            m = self.match_first(text, pos)
            if m is not None:
                out.append(m)
                pos = pos + 1 if m[1] == m[0] else m[1]
            else:
                pos += 1
        return out


def compile_onepass(program: Program) -> Optional[OnePassNFA]:
    """onepass.mojo:180-355.  None when the program is not one-pass."""
    n = len(program)
    if n == 0 or n > MAX_STATES:
        return None
    has_start = any(i[0] == OP_START_ANCHOR for i in program.instructions)
    has_end = any(i[0] == OP_END_ANCHOR for i in program.instructions)
    start_set = _epsilon_close(program, [0], at_start=True)
    sets = [start_set]
    index: Dict[tuple, int] = {start_set: 0}
    trans: List[List[int]] = [[ONEPASS_DEAD] * 256]
    work = [0]

    def find_or_add(s) -> Tuple[int, bool]:
        sid = index.get(s)
        if sid is not None:
            return sid, False
        if len(sets) >= ONEPASS_MAX_STATES:
            return -1, False
        sets.append(s)
        trans.append([ONEPASS_DEAD] * 256)
        index[s] = len(sets) - 1
        return len(sets) - 1, True

    while work:
        sid = work.pop()
        cur = sets[sid]
        active = [pc for pc in range(n) if cur[pc] and program.instructions[pc][0] in
                  (OP_BYTE, OP_CLASS, OP_ANY, OP_RANGE)]
        row = [ONEPASS_DEAD] * 256
        for byte in range(256):
            nxt = [pc + 1 for pc in active if _fires(program, pc, byte)]
            if not nxt:
                continue
            first = _epsilon_close(program, [nxt[0]], at_start=False)
            for other in nxt[1:]:
                if _epsilon_close(program, [other], at_start=False) != first:
                    return None  # ambiguous successors: not one-pass
            idx, is_new = find_or_add(first)
            if idx < 0:
                return None
            if is_new:
                work.append(idx)
            row[byte] = idx
        trans[sid] = row
    is_match = [_set_contains_match(program, s) for s in sets]
    is_end = [False] * len(sets)
    if has_end:
        is_end = [_closure_reaches_match_with_end_anchor(program, s) for s in sets]
    return OnePassNFA(trans, is_match, is_end, has_start, has_end)
