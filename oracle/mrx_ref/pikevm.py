"""ORACLE (test infrastructure) -- PikeVM bytecode, Thompson simulation and the
LazyDFA built on it.

Restates src/regex/pikevm.mojo: opcodes (:39-56), Program (:80-116),
compile_ast (:124-333), PikeVMEngine (:345-648) and LazyDFA (:664-987).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

from .frontend import (Node, RE, ELEMENT, WILDCARD, SPACE, DIGIT, WORD, RANGE,
                       START, END, OR, GROUP)
from .dfa_engine import (expand_character_range, build_nibble_tables,
                         find_first_in_nibble_tables)

OP_BYTE, OP_RANGE, OP_CLASS, OP_ANY, OP_SPLIT, OP_JUMP, OP_MATCH, \
    OP_START_ANCHOR, OP_END_ANCHOR = range(9)
MAX_STATES = 512  # pikevm.mojo:342


class Program:
    def __init__(self):
        self.instructions: List[List[int]] = []  # [opcode, arg0, arg1]
        self.class_tables: List[List[int]] = []

    def __len__(self):
        return len(self.instructions)

    def emit(self, op: int, a0: int = 0, a1: int = 0) -> int:
        self.instructions.append([op, a0, a1])
        return len(self.instructions) - 1

    def add_class_table(self, table: List[int]) -> int:
        for i, t in enumerate(self.class_tables):
            if t == table:
                return i
        self.class_tables.append(list(table))
        return len(self.class_tables) - 1

    def patch(self, idx: int, arg0: int = -1, arg1: int = -1):
        if arg0 >= 0:
            self.instructions[idx][1] = arg0
        if arg1 >= 0:
            self.instructions[idx][2] = arg1


def compile_ast(ast: Node) -> Program:
    """pikevm.mojo:124-139."""
    p = Program()
    if ast.type == RE and ast.get_children_len() > 0:
        _compile_node(ast.get_child(0), p)
    p.emit(OP_MATCH)
    return p


def _compile_node(node: Node, p: Program):
    """pikevm.mojo:142-164."""
    t = node.type
    if t == GROUP:
        for i in range(node.get_children_len()):
            _compile_quantified(node.get_child(i), p)
    elif t == OR:
        _compile_or(node, p)
    elif t == ELEMENT:
        v = node.get_value()
        if v and len(v) > 0:
            p.emit(OP_BYTE, v[0])
    elif t in (DIGIT, WORD, SPACE, RANGE):
        _compile_char_class_node(node, p)
    elif t == WILDCARD:
        p.emit(OP_ANY)
    elif t == START:
        p.emit(OP_START_ANCHOR)
    elif t == END:
        p.emit(OP_END_ANCHOR)


def _compile_quantified(node: Node, p: Program):
    """pikevm.mojo:174-229."""
    mn, mx = node.min, node.max
    if mn == 1 and mx == 1:
        _compile_node(node, p)
        return
    if mn == 0 and mx == 1:
        sp = p.emit(OP_SPLIT, 0, 0)
        body = len(p)
        _compile_node(node, p)
        p.patch(sp, arg0=body, arg1=len(p))
        return
    if mn == mx and mn > 1:
        for _ in range(mn):
            _compile_node(node, p)
        return
    if mx > 0:
        for _ in range(mn):
            _compile_node(node, p)
        splits = []
        for _ in range(mx - mn):
            splits.append(p.emit(OP_SPLIT, 0, 0))
            _compile_node(node, p)
        after = len(p)
        for s in splits:
            p.patch(s, arg0=s + 1, arg1=after)
        return
    if mx == -1:
        for _ in range(mn):
            _compile_node(node, p)
        sp = p.emit(OP_SPLIT, 0, 0)
        body = len(p)
        _compile_node(node, p)
        p.emit(OP_JUMP, sp)
        p.patch(sp, arg0=body, arg1=len(p))
        return
    # {0,0} and friends fall through: nothing emitted (pikevm.mojo:229)


def _compile_or(node: Node, p: Program):
    """pikevm.mojo:232-254."""
    n = node.get_children_len()
    if n == 0:
        return
    if n == 1:
        _compile_node(node.get_child(0), p)
        return
    sp = p.emit(OP_SPLIT, 0, 0)
    left_start = len(p)
    _compile_node(node.get_child(0), p)
    jp = p.emit(OP_JUMP, 0)
    right_start = len(p)
    _compile_node(node.get_child(1), p)
    after = len(p)
    p.patch(sp, arg0=left_start, arg1=right_start)
    p.patch(jp, arg0=after)


def _compile_char_class_node(node: Node, p: Program):
    """pikevm.mojo:271-333 (escapes inside [...] ARE interpreted here)."""
    table = [0] * 256
    if node.type in (DIGIT, WORD, SPACE):
        for b in expand_character_range(node.type, b""):
            table[b] = 1
    elif node.type == RANGE and node.get_value():
        raw = node.get_value()
        inner = raw
        if raw.startswith(b"[") and raw.endswith(b"]"):
            inner = raw[1:-1]
        j = 0
        n = len(inner)
        while j < n:
            if j + 1 < n and inner[j] == ord("\\"):
                nc = inner[j + 1]
                if nc == ord("s"):
                    for c in b" \t\n\r\x0c":
                        table[c] = 1
                elif nc == ord("d"):
                    for c in range(ord("0"), ord("9") + 1):
                        table[c] = 1
                elif nc == ord("w"):
                    for c in range(ord("a"), ord("z") + 1):
                        table[c] = 1
                    for c in range(ord("A"), ord("Z") + 1):
                        table[c] = 1
                    for c in range(ord("0"), ord("9") + 1):
                        table[c] = 1
                    table[ord("_")] = 1
                else:
                    table[nc] = 1
                j += 2
            elif j + 2 < n and inner[j + 1] == ord("-"):
                for c in range(inner[j], inner[j + 2] + 1):
                    table[c] = 1
                j += 3
            else:
                table[inner[j]] = 1
                j += 1
    if not node.positive_logic:
        table = [1 - x for x in table]
    p.emit(OP_CLASS, p.add_class_table(table))


class PikeVMEngine:
    """pikevm.mojo:345-648."""

    def __init__(self, program: Program):
        self.program = program
        self.first_byte_filter = [0] * 256
        self.has_filter = False
        self._build_first_byte_filter()

    def is_supported(self) -> bool:
        return len(self.program) <= MAX_STATES

    def _build_first_byte_filter(self):
        """pikevm.mojo:367-416."""
        n = len(self.program)
        if n == 0 or n > MAX_STATES:
            return
        seen = [0] * MAX_STATES
        stack = [0]
        matching = 0
        while stack:
            pc = stack.pop()
            if pc >= n or seen[pc] != 0:
                continue
            seen[pc] = 1
            op, a0, a1 = self.program.instructions[pc]
            if op == OP_BYTE:
                self.first_byte_filter[a0] = 1
                matching += 1
            elif op == OP_CLASS:
                tbl = self.program.class_tables[a0]
                for c in range(256):
                    if tbl[c] != 0:
                        self.first_byte_filter[c] = 1
                        matching += 1
            elif op == OP_RANGE:
                for c in range(a0, a1 + 1):
                    self.first_byte_filter[c] = 1
                    matching += 1
            elif op in (OP_ANY, OP_MATCH):
                self.has_filter = False
                return
            elif op == OP_SPLIT:
                stack.append(a0)
                stack.append(a1)
            elif op == OP_JUMP:
                stack.append(a0)
            elif op in (OP_START_ANCHOR, OP_END_ANCHOR):
                stack.append(pc + 1)
        self.has_filter = matching < 128

    def add_state(self, pcs: List[int], seen: List[int], pc: int, pos: int,
                  text_len: int):
        """_add_state, pikevm.mojo:604-648 (recursive epsilon closure)."""
        if pc >= len(self.program) or seen[pc] != 0:
            return
        op, a0, a1 = self.program.instructions[pc]
        if op == OP_SPLIT:
            seen[pc] = 1
            self.add_state(pcs, seen, a0, pos, text_len)
            self.add_state(pcs, seen, a1, pos, text_len)
        elif op == OP_JUMP:
            seen[pc] = 1
            self.add_state(pcs, seen, a0, pos, text_len)
        elif op == OP_START_ANCHOR:
            if pos == 0:
                seen[pc] = 1
                self.add_state(pcs, seen, pc + 1, pos, text_len)
        elif op == OP_END_ANCHOR:
            if pos == text_len:
                seen[pc] = 1
                self.add_state(pcs, seen, pc + 1, pos, text_len)
        else:
            seen[pc] = 1
            pcs.append(pc)

    def step_ok(self, pc: int, ch: int) -> bool:
        op, a0, a1 = self.program.instructions[pc]
        if op == OP_BYTE:
            return ch == a0
        if op == OP_CLASS:
            return self.program.class_tables[a0][ch] != 0
        if op == OP_ANY:
            return ch != 10
        if op == OP_RANGE:
            return a0 <= ch <= a1
        return False

    def run(self, text: bytes, start: int) -> Optional[Tuple[int, int]]:
        """_run, pikevm.mojo:497-602: leftmost-start (fixed), longest-end."""
        text_len = len(text)
        n = len(self.program)
        if n > MAX_STATES:
            return None
        cur: List[int] = []
        seen = [0] * n
        match_end = -1
        self.add_state(cur, seen, 0, start, text_len)
        pos = start
        while pos <= text_len:
            if not cur:
                break
            for pc in cur:
                if self.program.instructions[pc][0] == OP_MATCH:
                    match_end = pos
            if pos == text_len:
                break
            ch = text[pos]
            nxt: List[int] = []
            nseen = [0] * n
            for pc in cur:
                if self.step_ok(pc, ch):
                    self.add_state(nxt, nseen, pc + 1, pos + 1, text_len)
            cur = nxt
            pos += 1
        if match_end >= 0:
            return (start, match_end)
        return None


LAZY_DEAD = -1
LAZY_UNKNOWN = -2


class LazyDFA:
    """pikevm.mojo:682-987, including the transition cache (so that cache
    state carried between calls behaves as upstream)."""

    def __init__(self, vm: PikeVMEngine):
        self.has_end_anchor = any(ins[0] == OP_END_ANCHOR
                                  for ins in vm.program.instructions)
        self.pikevm = vm
        self.states: List[dict] = []
        self.start_state_id = self._get_or_create_state_for_pos(0, 0)
        self.lo_tbl = [0] * 16
        self.hi_tbl = [0] * 16
        if vm.has_filter:
            self.lo_tbl, self.hi_tbl = build_nibble_tables(vm.first_byte_filter)

    def reset(self):
        """The cache as a freshly constructed LazyDFA has it (pikevm.mojo:702-717): no states but the start state."""
        self.states = []
        self.start_state_id = self._get_or_create_state_for_pos(0, 0)

    def _has_match_in_set(self, nfa_set) -> bool:
        prog = self.pikevm.program
        return any(nfa_set[pc] != 0 and prog.instructions[pc][0] == OP_MATCH
                   for pc in range(len(prog)))

    def _find_or_create_state(self, nfa_set) -> int:
        key = tuple(nfa_set)
        for i, st in enumerate(self.states):
            if st["set"] == key:
                return i
        self.states.append({"set": key, "is_match": self._has_match_in_set(nfa_set),
                            "trans": [LAZY_UNKNOWN] * 256})
        return len(self.states) - 1

    def _get_or_create_state_for_pos(self, pc: int, pos: int) -> int:
        """pikevm.mojo:944-959: start closure built with pos=0, text_len=0."""
        pcs: List[int] = []
        seen = [0] * len(self.pikevm.program)
        self.pikevm.add_state(pcs, seen, pc, pos, 0)
        if len(pcs) == 0 and not self._has_match_in_set(seen):
            return LAZY_DEAD
        return self._find_or_create_state(seen)

    def _compute_transition(self, state_id: int, ch: int, pos: int, text_len: int) -> int:
        """pikevm.mojo:869-942."""
        st = self.states[state_id]
        prog = self.pikevm.program
        nxt: List[int] = []
        nseen = [0] * len(prog)
        for pc in range(len(prog)):
            if st["set"][pc] == 0:
                continue
            if self.pikevm.step_ok(pc, ch):
                self.pikevm.add_state(nxt, nseen, pc + 1, pos + 1, text_len)
        if len(nxt) == 0:
            return LAZY_DEAD
        return self._find_or_create_state(nseen)

    def run_lazy(self, text: bytes, start: int) -> Optional[Tuple[int, int]]:
        """_run_lazy, pikevm.mojo:819-867."""
        text_len = len(text)
        sid = self.start_state_id
        match_end = -1
        if sid == LAZY_DEAD:
            return None
        pos = start
        while pos < text_len:
            if self.states[sid]["is_match"]:
                match_end = pos
            ch = text[pos]
            nid = self.states[sid]["trans"][ch]
            if nid == LAZY_UNKNOWN:
                nid = self._compute_transition(sid, ch, pos, text_len)
                self.states[sid]["trans"][ch] = nid
            if nid == LAZY_DEAD:
                break
            sid = nid
            pos += 1
        if self.states[sid]["is_match"]:
            match_end = pos
        if match_end >= 0:
            return (start, match_end)
        return None

    def match_first(self, text: bytes, start: int = 0):
        return self.run_lazy(text, start)

    def _find_first_candidate(self, text: bytes, start: int, text_len: int) -> int:
        return find_first_in_nibble_tables(self.lo_tbl, self.hi_tbl,
                                           self.pikevm.first_byte_filter,
                                           text, start, text_len)

    def match_next(self, text: bytes, start: int = 0):
        """pikevm.mojo:754-780."""
        text_len = len(text)
        if self.pikevm.has_filter:
            pos = start
            while pos < text_len:
                cand = self._find_first_candidate(text, pos, text_len)
                if cand == -1:
                    break
                pos = cand
                r = self.run_lazy(text, pos)
                if r is not None:
                    return r
                pos += 1
            return self.run_lazy(text, text_len)
        for p in range(start, text_len + 1):
            r = self.run_lazy(text, p)
            if r is not None:
                return r
        return None

    def match_all(self, text: bytes):
        """pikevm.mojo:782-817."""
        text_len = len(text)
        out = []
        pos = 0
        if self.pikevm.has_filter:
            while pos < text_len:
                cand = self._find_first_candidate(text, pos, text_len)
                if cand == -1:
                    break
                pos = cand
                r = self.run_lazy(text, pos)
                if r is not None:
                    out.append(r)
                    pos = max(pos + 1, r[1])
                else:
                    pos += 1
            return out
        while pos <= text_len:
            r = self.run_lazy(text, pos)
            if r is not None:
                out.append(r)
                pos = max(pos + 1, r[1])
            else:
                pos += 1
        return out
