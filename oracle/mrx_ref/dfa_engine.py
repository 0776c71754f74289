"""ORACLE (test infrastructure) -- DFAEngine: table construction + matching.

Restates src/regex/dfa.mojo: _expand_character_range (:71-168), DFAState
(:215-254), the shape compilers (:308-1686), transitions (:1748-1803), matching
(:1815-2253), compile_dfa_pattern and its recognisers (:2385-3589); and the
byte-class matcher of src/regex/simd_ops.mojo:261-841 (CharacterClassSIMD:
lookup table, range detection, nibble tables, find_first_nibble_match,
count_consecutive_matches) plus simd_search / verify_match (:937-1024).

The matching methods here are pure-Python loops (small cases).  The same
loops exist in C (oracle/c/mrx_oracle.c) for large batches; tests check the
two agree.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

from . import SIMD_WIDTH
from .frontend import (Node, RE, ELEMENT, WILDCARD, SPACE, DIGIT, WORD, RANGE,
                       START, END, OR, GROUP)
from .analysis import (is_literal_pattern, get_literal_string, pattern_has_anchors,
                       common_prefix)

DIGITS = b"0123456789"
LOWER = b"abcdefghijklmnopqrstuvwxyz"
UPPER = b"ABCDEFGHIJKLMNOPQRSTUVWXYZ"
ALL_LETTERS = LOWER + UPPER
ALPHANUMERIC = LOWER + UPPER + DIGITS
WORD_CHARS = LOWER + UPPER + DIGITS + b"_"          # aliases.mojo:7-9
SPACE_CHARS = b" \t\n\r\f"
ALL_EXCEPT_NEWLINE = bytes(range(32, 127))           # aliases.mojo:1-3, 45


class DFACompileError(Exception):
    """The reference's compile_dfa_pattern raised (caught by HybridMatcher)."""


def expand_character_range(node_type: int, range_str: bytes) -> bytes:
    """dfa.mojo:71-168."""
    if node_type == DIGIT:
        return DIGITS
    if node_type == WORD:
        return WORD_CHARS
    if node_type == SPACE:
        return SPACE_CHARS
    if not range_str.startswith(b"[") or not range_str.endswith(b"]"):
        return range_str
    if range_str == b"[a-z]":
        return LOWER
    if range_str == b"[A-Z]":
        return UPPER
    if range_str == b"[0-9]":
        return DIGITS
    if range_str == b"[a-zA-Z0-9]":
        return ALPHANUMERIC
    if range_str == b"[a-zA-Z]":
        return ALL_LETTERS
    inner = range_str[1:-1]
    if inner.startswith(b"^"):
        inner = inner[1:]
    if len(inner) == 3 and inner[1] == ord("-"):
        s, e = inner[0], inner[2]
        # dfa.mojo:127-142 (slicing clamps like Mojo/Python slices)
        if s >= ord("a") and e <= ord("z"):
            return LOWER[max(s - ord("a"), 0):max(e - ord("a") + 1, 0)]
        elif s >= ord("a") and e <= ord("Z"):
            return UPPER[max(s - ord("A"), 0):max(e - ord("A") + 1, 0)]
        elif s >= ord("0") and e <= ord("9"):
            return DIGITS[max(s - ord("0"), 0):max(e - ord("0") + 1, 0)]
    out = bytearray()
    i = 0
    while i < len(inner):
        if i + 2 < len(inner) and inner[i + 1] == ord("-"):
            for c in range(inner[i], inner[i + 2] + 1):
                out.append(c & 0xFF) if c < 256 else None
            i += 3
        else:
            out.append(inner[i])
            i += 1
    return bytes(out)


class ClassMatcher:
    """CharacterClassSIMD (simd_ops.mojo:261-841), semantics only."""

    def __init__(self, char_class: bytes = b"", table: Optional[List[int]] = None):
        self.lookup = [0] * 256
        if table is not None:
            self.lookup = list(table)
        else:
            for b in char_class:
                self.lookup[b] = 1
        self._detect_ranges()
        self._build_nibble_tables()

    @staticmethod
    def for_class(char_class: bytes) -> "ClassMatcher":
        """get_character_class_matcher (simd_ops.mojo:1177-1216): the cached
        matchers are equal to a fresh one except whitespace, which adds \\v."""
        if char_class in (b" \t\n\r\f", b" \t\n\r\f\v"):
            return ClassMatcher(b" \t\n\r\f\v")
        return ClassMatcher(char_class)

    def _detect_ranges(self):
        # simd_ops.mojo:364-401
        starts, ends = [-1] * 4, [-1] * 4
        count = 0
        in_range = False
        for c in range(256):
            if self.lookup[c] != 0:
                if not in_range:
                    if count < 4:
                        starts[count] = c
                    in_range = True
            else:
                if in_range:
                    if count < 4:
                        ends[count] = c - 1
                    count += 1
                    in_range = False
        if in_range:
            if count < 4:
                ends[count] = 255
            count += 1
        if count > 3:
            count = 0
        self.num_ranges = count
        self.ranges = [(starts[k], ends[k]) for k in range(count)]

    def _build_nibble_tables(self):
        # simd_ops.mojo:63-86
        self.lo_tbl, self.hi_tbl = build_nibble_tables(self.lookup)

    def contains(self, c: int) -> bool:
        return 0 <= c < 256 and self.lookup[c] == 1

    def nibble_hit(self, c: int) -> bool:
        return (self.lo_tbl[c & 0xF] & self.hi_tbl[(c >> 4) & 0xF]) != 0

    def find_first_nibble_match(self, text: bytes, start: int, text_len: int) -> int:
        """simd_ops.mojo:566-648."""
        if self.num_ranges in (1, 2, 3):
            pos = start
            while pos < text_len:
                if self.lookup[text[pos]] != 0:
                    return pos
                pos += 1
            return -1
        return find_first_in_nibble_tables(self.lo_tbl, self.hi_tbl, self.lookup,
                                           text, start, text_len)

    def count_consecutive_matches(self, text: bytes, start: int, text_len: int) -> int:
        """simd_ops.mojo:651-786 (exact in every branch)."""
        pos = start
        while pos < text_len and self.lookup[text[pos]] != 0:
            pos += 1
        return pos - start


def build_nibble_tables(filt: List[int]) -> Tuple[List[int], List[int]]:
    """simd_ops.mojo:63-86."""
    lo_tbl, hi_tbl = [0] * 16, [0] * 16
    bucket = 0
    for c in range(256):
        if filt[c] != 0:
            bit = 1 << (bucket & 7)
            lo_tbl[c & 0xF] |= bit
            hi_tbl[(c >> 4) & 0xF] |= bit
            bucket += 1
    return lo_tbl, hi_tbl


def find_first_in_nibble_tables(lo_tbl, hi_tbl, filt, text: bytes, start: int,
                                text_len: int) -> int:
    """simd_ops.mojo:90-134.  SIMD chunks use the (inexact) nibble test, the
    scalar tail the exact filter."""
    pos = start
    W = SIMD_WIDTH
    while pos + W <= text_len:
        for i in range(W):
            c = text[pos + i]
            if (lo_tbl[c & 0xF] & hi_tbl[(c >> 4) & 0xF]) != 0:
                return pos + i
        pos += W
    while pos < text_len:
        if filt[text[pos]] != 0:
            return pos
        pos += 1
    return -1


def verify_match(pattern: bytes, text: bytes, pos: int) -> bool:
    """simd_ops.mojo:937-960."""
    if pos + len(pattern) > len(text):
        return False
    return text[pos:pos + len(pattern)] == pattern


def simd_search(pattern: bytes, text: bytes, start: int = 0) -> int:
    """simd_ops.mojo:963-1024: leftmost occurrence at or after start."""
    if len(pattern) == 0:
        return start
    if len(pattern) == 1:
        # simd_find_byte (:217-241): scans [start, text_len)
        if start >= len(text):
            return -1
        return text.find(pattern, start)
    if start < 0:
        start = 0
    if start + len(pattern) > len(text):
        return -1
    return text.find(pattern, start)


class DFAState:
    """dfa.mojo:215-254."""
    __slots__ = ("transitions", "is_accepting", "match_length")

    def __init__(self, is_accepting: bool = False, match_length: int = 0):
        self.transitions = [-1] * 256
        self.is_accepting = is_accepting
        self.match_length = match_length

    def add_transition(self, c: int, target: int):
        if 0 <= c < 256:
            self.transitions[c] = target


class SeqElement:
    """dfa.mojo:171-196."""

    def __init__(self, char_class: bytes, mn: int, mx: int, positive: bool = True):
        self.char_class = char_class
        self.min_matches = mn
        self.max_matches = mx
        self.positive_logic = positive
        self.alternation_branches: List[bytes] = []


class SeqInfo:
    def __init__(self):
        self.elements: List[SeqElement] = []
        self.has_start_anchor = False
        self.has_end_anchor = False


class DFAEngine:
    """dfa.mojo:257-2294."""

    def __init__(self):
        self.states: List[DFAState] = []
        self.start_state = 0
        self.has_start_anchor = False
        self.has_end_anchor = False
        self.is_pure_literal = False
        self.matcher = ClassMatcher(b"")
        self.has_simd_matcher = False
        self.simd_scan_eligible = False
        self.literal_pattern = b""
        self.shape = ""          # which dispatcher branch fired (introspection)

    # ---- compilers ---------------------------------------------------------
    def _create_accepting_state(self):
        self.states.append(DFAState(True, 0))
        self.start_state = 0

    def compile_pattern(self, pattern: bytes, has_start: bool, has_end: bool):
        """dfa.mojo:308-352."""
        self.has_start_anchor = has_start
        self.has_end_anchor = has_end
        self.literal_pattern = pattern
        if len(pattern) == 0:
            self._create_accepting_state()
            return
        if not has_start and not has_end:
            self.is_pure_literal = True
        for i, c in enumerate(pattern):
            st = DFAState()
            st.add_transition(c, i + 1)
            self.states.append(st)
        self.states.append(DFAState(True, len(pattern)))
        self.start_state = 0

    def _cc(self, frm: int, to: int, char_class: bytes, positive: bool):
        """_add_character_class_transitions_with_logic, dfa.mojo:1748-1803."""
        if frm >= len(self.states):
            return
        st = self.states[frm]
        if positive:
            for c in char_class:
                st.add_transition(c, to)
        else:
            st.transitions = [to] * 256
            for c in char_class:
                st.transitions[c] = -1

    def compile_character_class_with_logic(self, char_class: bytes, mn: int, mx: int,
                                           positive: bool):
        """dfa.mojo:375-496."""
        if mn >= 0 and positive:
            self.matcher = ClassMatcher.for_class(char_class)
            self.has_simd_matcher = True
            self.simd_scan_eligible = (mx == -1)
        S = self.states
        if mn == 0:
            S.append(DFAState(True, 0))
            S.append(DFAState(True, 1))
            self._cc(0, 1, char_class, positive)
            if mx == -1 or mx > 1:
                self._cc(1, 1, char_class, positive)
        elif mn == 1:
            S.append(DFAState())
            S.append(DFAState(True, 1))
            self._cc(0, 1, char_class, positive)
            if mx == -1:
                self._cc(1, 1, char_class, positive)
            elif mx > 1:
                for k in range(2, mx + 1):
                    S.append(DFAState(True, k))
                    self._cc(k - 1, k, char_class, positive)
        else:
            for k in range(mn + 1):
                S.append(DFAState(k >= mn, k))
                if k > 0:
                    self._cc(k - 1, k, char_class, positive)
            if mx == -1:
                last = len(S) - 1
                self._cc(last, last, char_class, positive)
            elif mx > mn:
                for k in range(mn + 1, mx + 1):
                    S.append(DFAState(True, k))
                    self._cc(k - 1, k, char_class, positive)
        self.start_state = 0

    def compile_sequential_pattern(self, info: SeqInfo):
        """dfa.mojo:498-605."""
        self.has_start_anchor = info.has_start_anchor
        self.has_end_anchor = info.has_end_anchor
        if not info.elements:
            self._create_accepting_state()
            return
        S = self.states
        cur = 0
        n = len(info.elements)
        for idx, el in enumerate(info.elements):
            is_last = idx == n - 1
            if el.min_matches == 0:
                if idx == 0:
                    S.append(DFAState(not is_last))
                    cur = 0
                S.append(DFAState(True))
                m = len(S) - 1
                self._cc(cur, m, el.char_class, el.positive_logic)
                if el.max_matches == -1:
                    self._cc(m, m, el.char_class, el.positive_logic)
                cur = m
            else:
                for k in range(el.min_matches):
                    acc = (k >= el.min_matches - 1) and is_last
                    S.append(DFAState(acc))
                    si = len(S) - 1
                    if k == 0:
                        self._cc(cur, si, el.char_class, el.positive_logic)
                    else:
                        self._cc(si - 1, si, el.char_class, el.positive_logic)
                    cur = si
                if el.max_matches == -1:
                    self._cc(cur, cur, el.char_class, el.positive_logic)
                elif el.max_matches > el.min_matches:
                    for _ in range(el.max_matches - el.min_matches):
                        S.append(DFAState(is_last))
                        si = len(S) - 1
                        self._cc(cur, si, el.char_class, el.positive_logic)
                        cur = si
        self.start_state = 0

    def compile_multi_character_class_sequence(self, info: SeqInfo):
        """dfa.mojo:607-871."""
        self.has_start_anchor = info.has_start_anchor
        self.has_end_anchor = info.has_end_anchor
        if not info.elements:
            self._create_accepting_state()
            return
        e0 = info.elements[0]
        if len(e0.alternation_branches) == 0 and len(e0.char_class) > 0:
            self.matcher = ClassMatcher.for_class(e0.char_class)
            self.has_simd_matcher = True
        S = self.states
        cur = 0
        n = len(info.elements)
        for idx, el in enumerate(info.elements):
            is_last = idx == n - 1
            all_rest_optional = all(info.elements[j].min_matches <= 0
                                    for j in range(idx + 1, n))
            if len(el.alternation_branches) > 0:
                # dfa.mojo:660-697
                if idx == 0:
                    S.append(DFAState())
                    cur = 0
                S.append(DFAState(is_last or all_rest_optional))
                end_idx = len(S) - 1
                for br in el.alternation_branches:
                    prev = cur
                    for ci, c in enumerate(br):
                        if ci == len(br) - 1:
                            S[prev].add_transition(c, end_idx)
                        else:
                            S.append(DFAState())
                            mid = len(S) - 1
                            S[prev].add_transition(c, mid)
                            prev = mid
                cur = end_idx
                continue
            if el.min_matches == 0:
                # dfa.mojo:699-744
                if idx == 0:
                    S.append(DFAState(all_rest_optional))
                    cur = 0
                S.append(DFAState(is_last or all_rest_optional))
                m = len(S) - 1
                self._cc(cur, m, el.char_class, el.positive_logic)
                if el.max_matches == -1:
                    self._cc(m, m, el.char_class, el.positive_logic)
                if idx == 0:
                    cur = 0
                else:
                    cur = m
            elif el.min_matches == 1:
                # dfa.mojo:746-807
                if idx == 0:
                    S.append(DFAState())
                    cur = 0
                S.append(DFAState(is_last or all_rest_optional))
                m = len(S) - 1
                self._cc(cur, m, el.char_class, el.positive_logic)
                if idx == 1 and cur == 0:
                    prev_el = info.elements[0]
                    if prev_el.min_matches == 0:
                        self._cc(1, m, el.char_class, el.positive_logic)
                if el.max_matches == -1:
                    self._cc(m, m, el.char_class, el.positive_logic)
                elif el.max_matches > 1:
                    for k in range(2, el.max_matches + 1):
                        S.append(DFAState(is_last))
                        ai = len(S) - 1
                        self._cc(m + k - 2, ai, el.char_class, el.positive_logic)
                cur = m
            else:
                # dfa.mojo:809-869
                if idx == 0:
                    S.append(DFAState())
                    cur = 0
                for k in range(el.min_matches):
                    acc = (k >= el.min_matches - 1) and is_last
                    S.append(DFAState(acc))
                    si = len(S) - 1
                    if k > 0:
                        self._cc(si - 1, si, el.char_class, el.positive_logic)
                    else:
                        self._cc(cur, si, el.char_class, el.positive_logic)
                    cur = si
                if el.max_matches == -1:
                    self._cc(cur, cur, el.char_class, el.positive_logic)
                elif el.max_matches > el.min_matches:
                    for _ in range(el.min_matches + 1, el.max_matches + 1):
                        S.append(DFAState(is_last))
                        oi = len(S) - 1
                        self._cc(cur, oi, el.char_class, el.positive_logic)
                        cur = oi
        self.start_state = 0

    def _find_or_create_state(self, frm: int, c: int) -> int:
        """dfa.mojo:1245-1268."""
        t = self.states[frm].transitions[c]
        if t != -1:
            return t
        self.states.append(DFAState())
        ni = len(self.states) - 1
        self.states[frm].add_transition(c, ni)
        return ni

    def compile_alternation(self, ast: Node):
        """dfa.mojo:873-928."""
        self.states = [DFAState()]
        self.start_state = 0
        or_node = _find_or_node(ast)
        if or_node is None:
            raise DFACompileError("No OR node found in alternation pattern")
        self.states.append(DFAState(True, 0))
        acc = len(self.states) - 1
        for br in _collect_all_alternation_branches(or_node):
            if len(br) == 0:
                continue
            cur = 0
            for j, c in enumerate(br):
                if j == len(br) - 1:
                    self.states[cur].add_transition(c, acc)
                else:
                    cur = self._find_or_create_state(cur, c)

    def _chain_to(self, text: bytes, final_target_fn):
        """Shared body of the (pattern)?/(pattern)* chain builders: every byte
        but the last opens a new state; the last goes where the caller says."""
        cur = 0
        for i, c in enumerate(text):
            if i == len(text) - 1:
                final_target_fn(cur, c)
            else:
                self.states.append(DFAState())
                ni = len(self.states) - 1
                self.states[cur].add_transition(c, ni)
                cur = ni

    def _one_or_more_chain(self, text: bytes):
        """dfa.mojo:1041-1074 and :1209-1243 (identical bodies)."""
        def last(cur, c):
            self.states.append(DFAState(True))
            li = len(self.states) - 1
            self.states[cur].add_transition(c, li)
            self.states[li].add_transition(text[0], 1 if len(text) > 1 else li)
        self._chain_to(text, last)

    def compile_quantified_group(self, ast: Node):
        """dfa.mojo:930-980."""
        self.states = [DFAState()]
        self.start_state = 0
        inner = ast.get_child(0).get_child(0)
        mn, mx = inner.min, inner.max
        text = _extract_group_text(inner)
        if len(text) == 0:
            raise DFACompileError("Empty quantified group")
        self.states.append(DFAState(True, 0))
        acc = len(self.states) - 1
        if mn == 0 and mx == 1:
            # dfa.mojo:982-1010 ("epsilon" is a transition on byte 0)
            self.states[0].add_transition(0, acc)
            self._chain_to(text, lambda cur, c: self.states[cur].add_transition(c, acc))
        elif mn == 0 and mx == -1:
            # dfa.mojo:1012-1039
            self.states[0].is_accepting = True
            self._chain_to(text, lambda cur, c: self.states[cur].add_transition(c, 0))
        elif mn == 1 and mx == -1:
            self._one_or_more_chain(text)
        else:
            raise DFACompileError("Unsupported quantifier range for group")

    def compile_simple_quantifier(self, ast: Node):
        """dfa.mojo:1076-1145 (the whole element sequence is quantified)."""
        self.states = [DFAState()]
        self.start_state = 0
        group = ast.get_child(0)
        qmin, qmax = 1, 1
        text = b""
        for i in range(group.get_children_len()):
            el = group.get_child(i)
            if el.min == 0 and el.max == -1:
                qmin, qmax = 0, -1
            elif el.min == 1 and el.max == -1:
                qmin, qmax = 1, -1
            elif el.min == 0 and el.max == 1:
                qmin, qmax = 0, 1
            text += el.get_value()
        if len(text) == 0:
            raise DFACompileError("Empty quantifier pattern")
        self.states.append(DFAState(True, 0))
        acc = len(self.states) - 1
        if qmin == 0 and qmax == 1:
            # dfa.mojo:1147-1177
            self.states[0].is_accepting = True
            self._chain_to(text, lambda cur, c: self.states[cur].add_transition(c, acc))
        elif qmin == 0 and qmax == -1:
            # dfa.mojo:1179-1207
            self.states[0].is_accepting = True
            self._chain_to(text, lambda cur, c: self.states[cur].add_transition(c, 0))
        elif qmin == 1 and qmax == -1:
            self._one_or_more_chain(text)
        else:
            raise DFACompileError("Unsupported quantifier type for simple quantifier")

    def compile_wildcard_quantifier(self, ast: Node):
        """dfa.mojo:1270-1361."""
        self.states = [DFAState()]
        self.start_state = 0
        wc = ast.get_child(0).get_child(0)
        qmin, qmax = wc.min, wc.max
        self.states.append(DFAState(True, 0))
        acc = len(self.states) - 1
        not_nl = [c for c in range(256) if c != 10]
        if qmin == 0 and qmax == 1:
            self.states[0].is_accepting = True
            for c in not_nl:
                self.states[0].add_transition(c, acc)
        elif qmin == 0 and qmax == -1:
            self.states[0].is_accepting = True
            for c in not_nl:
                self.states[0].add_transition(c, 0)
        elif qmin == 1 and qmax == -1:
            self.states.append(DFAState(True))
            li = len(self.states) - 1
            for c in not_nl:
                self.states[0].add_transition(c, li)
            for c in not_nl:
                self.states[li].add_transition(c, li)
        elif qmin == 1 and qmax == 1:
            for c in not_nl:
                self.states[0].add_transition(c, acc)
        else:
            raise DFACompileError("Unsupported quantifier type for wildcard quantifier")

    def compile_common_prefix_alternation(self, ast: Node):
        """dfa.mojo:1363-1463."""
        self.states = [DFAState()]
        self.start_state = 0
        or_node = ast.get_child(0).get_child(0).get_child(0)
        branches: List[bytes] = []
        _extract_branches_lenient(or_node, branches)
        if not branches:
            return
        prefix = common_prefix(branches)
        cur = 0
        for c in prefix:
            cur = self._find_or_create_state(cur, c)
        for br in branches:
            if len(br) == len(prefix):
                self.states[cur].is_accepting = True
            else:
                suffix = br[len(prefix):]
                sc = cur
                for j, c in enumerate(suffix):
                    if j == len(suffix) - 1:
                        t = self._find_or_create_state(sc, c)
                        self.states[t].is_accepting = True
                    else:
                        sc = self._find_or_create_state(sc, c)

    def compile_quantified_alternation_group(self, ast: Node):
        """dfa.mojo:1507-1686."""
        self.states = [DFAState()]
        self.start_state = 0
        qg = ast.get_child(0).get_child(0)
        or_node = qg.get_child(0)
        qmin, qmax = qg.min, qg.max
        branches: List[bytes] = []
        _extract_branches_lenient(or_node, branches)

        def paths(origin: int, final: int):
            for br in branches:
                cur = origin
                for j, c in enumerate(br):
                    if j == len(br) - 1:
                        self.states[cur].add_transition(c, final)
                    else:
                        cur = self._find_or_create_state(cur, c)

        if qmin == 0 and qmax == 1:
            self.states[0].is_accepting = True
            self.states.append(DFAState(True))
            paths(0, len(self.states) - 1)
        elif qmin == 0 and qmax == -1:
            self.states[0].is_accepting = True
            paths(0, 0)
        elif qmin == 1 and qmax == -1:
            self.states.append(DFAState(True))
            li = len(self.states) - 1
            paths(0, li)
            paths(li, li)
        else:
            raise DFACompileError(
                "Unsupported quantifier type for quantified alternation group")

    # ---- matching ----------------------------------------------------------
    def is_match(self, text: bytes, start: int = 0) -> bool:
        """dfa.mojo:1815-1849."""
        if self.has_start_anchor and start > 0:
            return False
        if self.has_simd_matcher and len(self.states) > 0:
            if start >= len(text):
                return self.states[self.start_state].is_accepting
            if self.matcher.contains(text[start]):
                return True
            return self.states[self.start_state].is_accepting
        return self._try_match_at_position(text, start, True) is not None

    def match_first(self, text: bytes, start: int = 0):
        """dfa.mojo:1852-1872."""
        if self.has_start_anchor and start > 0:
            return None
        return self._try_match_at_position(text, start, True)

    def match_next(self, text: bytes, start: int = 0):
        """dfa.mojo:1875-1903."""
        if self.has_start_anchor:
            if start == 0:
                return self._try_match_at_position(text, 0)
            return None
        if self.has_simd_matcher and not self.has_end_anchor:
            return self._optimized_simd_search(text, start)
        for p in range(start, len(text) + 1):
            r = self._try_match_at_position(text, p)
            if r is not None:
                return r
        return None

    def _try_match_at_position(self, text: bytes, start_pos: int,
                               require_exact_position: bool = False):
        """dfa.mojo:1906-2026.  Returns (start, end) or None."""
        text_len = len(text)
        if start_pos > text_len:
            return None
        if self.is_pure_literal:
            plen = len(self.literal_pattern)
            if require_exact_position:
                if verify_match(self.literal_pattern, text, start_pos):
                    return (start_pos, start_pos + plen)
                return None
            pos = simd_search(self.literal_pattern, text, start_pos)
            if pos != -1:
                return (pos, pos + plen)
            return None
        S = self.states
        if (self.has_simd_matcher and len(S) > 0
                and (self.simd_scan_eligible or S[self.start_state].is_accepting)):
            r = self._try_match_simd(text, start_pos)
            if r is not None:
                return r
        if start_pos == text_len:
            if len(S) > 0 and S[self.start_state].is_accepting:
                return (start_pos, start_pos)
            return None
        cur = self.start_state
        pos = start_pos
        last_acc = -1
        if cur < len(S) and S[cur].is_accepting:
            last_acc = pos
        while pos < text_len:
            nxt = S[cur].transitions[text[pos]]
            if nxt == -1:
                break
            cur = nxt
            pos += 1
            if S[cur].is_accepting:
                last_acc = pos
        if pos == text_len and cur < len(S) and S[cur].is_accepting:
            last_acc = pos
        if last_acc != -1:
            if self.has_end_anchor and last_acc != text_len:
                return None
            return (start_pos, last_acc)
        return None

    def _try_match_simd(self, text: bytes, start_pos: int):
        """dfa.mojo:2133-2197."""
        if not self.has_simd_matcher or len(self.states) == 0:
            return None
        text_len = len(text)
        start_acc = self.states[self.start_state].is_accepting
        if not start_acc and not self.simd_scan_eligible:
            return None
        n = self.matcher.count_consecutive_matches(text, start_pos, text_len)
        valid = False
        end = start_pos + n
        if n == 0:
            if start_acc:
                valid = True
                end = start_pos
        else:
            valid = True
        if valid:
            if self.has_end_anchor and end != text_len:
                return None
            return (start_pos, end)
        return None

    def _optimized_simd_search(self, text: bytes, start: int):
        """dfa.mojo:2200-2253."""
        if not self.has_simd_matcher:
            return None
        m = self.matcher
        text_len = len(text)
        pos = start
        if self.simd_scan_eligible:
            while pos < text_len:
                mp = m.find_first_nibble_match(text, pos, text_len)
                if mp == -1:
                    return None
                ml = m.count_consecutive_matches(text, mp, text_len)
                if ml > 0:
                    me = mp + ml
                    if self.has_end_anchor and me != text_len:
                        pos = me
                        continue
                    return (mp, me)
                pos = mp + 1
            return None
        while pos < text_len:
            fp = m.find_first_nibble_match(text, pos, text_len)
            if fp == -1:
                return None
            r = self._try_match_at_position(text, fp)
            if r is not None:
                return r
            pos = fp + 1
        return None

    def match_all(self, text: bytes) -> List[Tuple[int, int]]:
        """dfa.mojo:2028-2130."""
        text_len = len(text)
        out: List[Tuple[int, int]] = []
        if self.has_start_anchor or self.has_end_anchor:
            r = self.match_next(text, 0)
            if r is not None:
                out.append(r)
            return out
        pos = 0
        if self.is_pure_literal:
            plen = len(self.literal_pattern)
            while pos <= text_len - plen:
                hit = simd_search(self.literal_pattern, text, pos)
                if hit == -1:
                    break
                out.append((hit, hit + plen))
                pos = hit + plen
            return out
        if self.has_simd_matcher and len(self.states) > 0:
            m = self.matcher
            if self.simd_scan_eligible:
                while pos < text_len:
                    mp = m.find_first_nibble_match(text, pos, text_len)
                    if mp == -1:
                        break
                    ml = m.count_consecutive_matches(text, mp, text_len)
                    if ml > 0:
                        out.append((mp, mp + ml))
                        pos = mp + ml
                    else:
                        pos = mp + 1
                return out
            while pos < text_len:
                np_ = m.find_first_nibble_match(text, pos, text_len)
                if np_ == -1:
                    break
                pos = np_
                r = self._try_match_at_position(text, pos)
                if r is not None:
                    out.append(r)
                    if r[1] == r[0]:
                        pos += 1
                    else:
                        pos = r[1]
                else:
                    pos += 1
            return out
        while pos <= text_len:
            r = self._try_match_at_position(text, pos)
            if r is not None:
                out.append(r)
                if r[1] == r[0]:
                    pos += 1
                else:
                    pos = r[1]
            else:
                pos += 1
        return out


# ---------------------------------------------------------------------------
# compile_dfa_pattern + recognisers (dfa.mojo:2385-3589)
# ---------------------------------------------------------------------------
def compile_dfa_pattern(ast: Node) -> DFAEngine:
    """dfa.mojo:2385-2496: first matching shape wins."""
    dfa = DFAEngine()
    if is_literal_pattern(ast):
        hs, he = pattern_has_anchors(ast)
        dfa.compile_pattern(get_literal_string(ast), hs, he)
        dfa.shape = "literal"
    elif _is_pure_anchor_pattern(ast):
        hs, he = pattern_has_anchors(ast)
        dfa.compile_pattern(b"", hs, he)
        dfa.shape = "pure_anchor"
    elif _is_simple_character_class_pattern(ast):
        cc, mn, mx, hs, he, positive = _extract_character_class_info(ast)
        if cc is None:
            raise DFACompileError("character class without value")
        expanded = expand_character_range(ast.type, cc)
        dfa.compile_character_class_with_logic(expanded, mn, mx, positive)
        dfa.has_start_anchor = hs
        dfa.has_end_anchor = he
        dfa.shape = "single_class"
    elif _is_multi_character_class_sequence(ast):
        dfa.compile_multi_character_class_sequence(_extract_multi_class_sequence_info(ast))
        dfa.shape = "multi_class_sequence"
    elif _is_sequential_character_class_pattern(ast):
        dfa.compile_sequential_pattern(_extract_sequential_pattern_info(ast))
        dfa.shape = "sequential"
    elif _is_mixed_sequential_pattern(ast):
        dfa.compile_multi_character_class_sequence(_extract_multi_class_sequence_info(ast))
        dfa.shape = "mixed_sequential"
    elif _is_alternation_pattern(ast):
        dfa.compile_alternation(ast)
        dfa.has_start_anchor, dfa.has_end_anchor = pattern_has_anchors(ast)
        dfa.shape = "alternation"
    elif _is_quantified_group(ast):
        dfa.compile_quantified_group(ast)
        dfa.has_start_anchor, dfa.has_end_anchor = pattern_has_anchors(ast)
        dfa.shape = "quantified_group"
    elif _is_simple_quantifier_pattern(ast):
        dfa.compile_simple_quantifier(ast)
        dfa.has_start_anchor, dfa.has_end_anchor = pattern_has_anchors(ast)
        dfa.shape = "simple_quantifier"
    elif _is_wildcard_quantifier_pattern(ast):
        dfa.compile_wildcard_quantifier(ast)
        dfa.has_start_anchor, dfa.has_end_anchor = pattern_has_anchors(ast)
        dfa.shape = "wildcard_quantifier"
    elif _is_common_prefix_alternation_pattern(ast):
        dfa.compile_common_prefix_alternation(ast)
        dfa.has_start_anchor, dfa.has_end_anchor = pattern_has_anchors(ast)
        dfa.shape = "common_prefix_alternation"
    elif _is_quantified_alternation_group(ast):
        dfa.compile_quantified_alternation_group(ast)
        dfa.has_start_anchor, dfa.has_end_anchor = pattern_has_anchors(ast)
        dfa.shape = "quantified_alternation_group"
    else:
        raise DFACompileError("Pattern too complex for current DFA implementation")
    return dfa


def _is_simple_character_class_pattern(ast: Node) -> bool:
    """dfa.mojo:2499-2529."""
    if _is_multi_character_class_sequence(ast):
        return False
    if ast.type == RE and ast.get_children_len() == 1:
        ch = ast.get_child(0)
        if ch.type in (DIGIT, WORD, RANGE):
            return True
        if ch.type == GROUP and ch.get_children_len() == 1:
            return ch.get_child(0).type in (DIGIT, WORD, RANGE)
    elif ast.type in (DIGIT, WORD, RANGE):
        return True
    return False


def _extract_character_class_info(ast: Node):
    """dfa.mojo:2532-2602."""
    cc = None
    mn = mx = 1
    hs = he = False
    positive = True
    if ast.type in (DIGIT, WORD, RANGE):
        node = ast
    elif ast.type == RE and ast.get_children_len() == 1:
        c = ast.get_child(0)
        if c.type in (DIGIT, WORD, RANGE):
            node = c
        elif c.type == GROUP and c.get_children_len() == 1:
            node = c.get_child(0)
        else:
            node = c
        hs, he = pattern_has_anchors(ast)
    else:
        node = ast
    if node.type == DIGIT:
        mn, mx, positive, cc = node.min, node.max, node.positive_logic, DIGITS
    elif node.type == WORD:
        mn, mx, positive, cc = node.min, node.max, node.positive_logic, WORD_CHARS
    elif node.type == RANGE:
        mn, mx, positive = node.min, node.max, node.positive_logic
        v = node.get_value()
        if v:
            cc = v
    return cc, mn, mx, hs, he, positive


def _is_pure_anchor_pattern(ast: Node) -> bool:
    """dfa.mojo:2605-2629."""
    if ast.type in (START, END):
        return True
    if ast.type == RE:
        if not ast.has_children():
            return False
        return _is_pure_anchor_pattern(ast.get_child(0))
    if ast.type == GROUP:
        for i in range(ast.get_children_len()):
            if not _is_pure_anchor_pattern(ast.get_child(i)):
                return False
        return True
    return False


def _is_sequential_character_class_pattern(ast: Node) -> bool:
    """dfa.mojo:2632-2663."""
    if ast.type != RE or ast.get_children_len() != 1:
        return False
    ch = ast.get_child(0)
    if ch.type != GROUP:
        return False
    for i in range(ch.get_children_len()):
        if ch.get_child(i).type not in (RANGE, DIGIT, WORD):
            return False
    return ch.get_children_len() >= 2


def _extract_sequential_pattern_info(ast: Node) -> SeqInfo:
    """dfa.mojo:2666-2708."""
    info = SeqInfo()
    info.has_start_anchor, info.has_end_anchor = pattern_has_anchors(ast)
    if ast.type == RE and ast.get_children_len() == 1:
        ch = ast.get_child(0)
        if ch.type == GROUP:
            for i in range(ch.get_children_len()):
                e = ch.get_child(i)
                if e.type == DIGIT:
                    cc = DIGITS
                elif e.type == WORD:
                    cc = WORD_CHARS
                elif e.type == RANGE:
                    cc = expand_character_range(e.type, e.get_value())
                else:
                    continue
                info.elements.append(SeqElement(cc, e.min, e.max, e.positive_logic))
    return info


def _element_to_char_class(e: Node) -> bytes:
    """dfa.mojo:2885-2908."""
    if e.type == DIGIT:
        return DIGITS
    if e.type == WORD:
        return WORD_CHARS
    if e.type == RANGE:
        return expand_character_range(e.type, e.get_value())
    if e.type == SPACE:
        return SPACE_CHARS
    if e.type == WILDCARD:
        return ALL_EXCEPT_NEWLINE
    if e.type == ELEMENT:
        if e.get_value():
            return e.get_value()
    return b""


def _is_char_class_group(node: Node) -> bool:
    """dfa.mojo:2711-2733."""
    if node.type != GROUP:
        return False
    has_cc = False
    for i in range(node.get_children_len()):
        ch = node.get_child(i)
        if not _element_to_char_class(ch):
            return False
        if ch.type != ELEMENT:
            has_cc = True
    return has_cc


def _is_literal_alternation_group(node: Node) -> bool:
    """dfa.mojo:2736-2771."""
    if node.type != GROUP or node.get_children_len() != 1:
        return False
    child = node.get_child(0)
    if child.type != OR:
        return False
    has_branch = False
    stack = [child]
    while stack:
        cur = stack.pop()
        if cur.type == GROUP:
            for i in range(cur.get_children_len()):
                if cur.get_child(i).type != ELEMENT:
                    return False
            has_branch = True
        elif cur.type == OR:
            if cur.get_children_len() != 2:
                return False
            stack.append(cur.get_child(1))
            stack.append(cur.get_child(0))
        else:
            return False
    return has_branch


def _collect_alternation_branches(node: Node) -> List[bytes]:
    """dfa.mojo:2774-2812 (left-to-right branch order)."""
    branches: List[bytes] = []
    stack: List[Node] = []
    if node.get_children_len() > 0:
        stack.append(node.get_child(0))
    while stack:
        cur = stack.pop()
        if cur.type == GROUP:
            br = b""
            for i in range(cur.get_children_len()):
                ch = cur.get_child(i)
                if ch.type == ELEMENT and ch.get_value():
                    br += ch.get_value()
            branches.append(br)
        elif cur.type == OR:
            if cur.get_children_len() >= 2:
                stack.append(cur.get_child(1))
            if cur.get_children_len() >= 1:
                stack.append(cur.get_child(0))
    return branches


def _is_multi_character_class_sequence(ast: Node) -> bool:
    """dfa.mojo:2815-2882."""
    if ast.type != RE or ast.get_children_len() != 1:
        return False
    ch = ast.get_child(0)
    if ch.type != GROUP:
        return False
    if ch.get_children_len() < 2:
        return False
    cc = 0
    for i in range(ch.get_children_len()):
        e = ch.get_child(i)
        if e.type in (RANGE, DIGIT, WORD, SPACE):
            cc += 1
        elif e.type == WILDCARD:
            cc += 1
        elif e.type == ELEMENT and e.min == 1 and e.max == 1:
            pass
        elif e.type == GROUP and _is_literal_alternation_group(e):
            pass
        elif e.type == GROUP and _is_char_class_group(e):
            cc += 1
        else:
            return False
    return cc >= 2


def _extract_multi_class_sequence_info(ast: Node) -> SeqInfo:
    """dfa.mojo:2911-2970."""
    info = SeqInfo()
    info.has_start_anchor, info.has_end_anchor = pattern_has_anchors(ast)
    if ast.type == RE and ast.get_children_len() == 1:
        ch = ast.get_child(0)
        if ch.type == GROUP:
            for i in range(ch.get_children_len()):
                e = ch.get_child(i)
                if e.type == GROUP and _is_literal_alternation_group(e):
                    pe = SeqElement(b"", 1, 1, True)
                    pe.alternation_branches = _collect_alternation_branches(e)
                    info.elements.append(pe)
                elif e.type == GROUP and _is_char_class_group(e):
                    for j in range(e.get_children_len()):
                        sub = e.get_child(j)
                        sc = _element_to_char_class(sub)
                        if sc:
                            info.elements.append(
                                SeqElement(sc, sub.min, sub.max, sub.positive_logic))
                else:
                    cc = _element_to_char_class(e)
                    if cc:
                        info.elements.append(
                            SeqElement(cc, e.min, e.max, e.positive_logic))
    return info


def _is_mixed_sequential_pattern(ast: Node) -> bool:
    """dfa.mojo:2973-3015."""
    if ast.type != RE or ast.get_children_len() != 1:
        return False
    ch = ast.get_child(0)
    if ch.type != GROUP:
        return False
    if ch.get_children_len() < 3:
        return False
    has_cc = has_opt = False
    for i in range(ch.get_children_len()):
        e = ch.get_child(i)
        if e.type in (RANGE, DIGIT, WORD):
            has_cc = True
        elif e.type == ELEMENT:
            if e.min == 0 and e.max == 1:
                has_opt = True
    return has_cc and has_opt


def _is_alternation_pattern(ast: Node) -> bool:
    """dfa.mojo:3034-3051."""
    if ast.type == OR:
        return _is_simple_alternation_branches(ast)
    return _is_pure_alternation_pattern(ast)


def _group_contains_only_literals(g: Node) -> bool:
    """dfa.mojo:3100-3120."""
    if g.type != GROUP:
        return False
    for i in range(g.get_children_len()):
        if g.get_child(i).type != ELEMENT:
            return False
    return True


def _is_simple_alternation_branches(ast: Node) -> bool:
    """dfa.mojo:3054-3097."""
    if ast.type != OR:
        return False
    for i in range(ast.get_children_len()):
        br = ast.get_child(i)
        if br.type == GROUP:
            if not _group_contains_only_literals(br):
                inner = br
                while inner.type == GROUP and inner.get_children_len() == 1:
                    inner = inner.get_child(0)
                if inner.type == OR:
                    if not _is_simple_alternation_branches(inner):
                        return False
                else:
                    return False
        elif br.type == ELEMENT:
            continue
        elif br.type == OR:
            if not _is_simple_alternation_branches(br):
                return False
        else:
            return False
    return True


def _find_or_node(ast: Node) -> Optional[Node]:
    """dfa.mojo:3146-3169."""
    if ast.type == OR:
        return ast
    for i in range(ast.get_children_len()):
        f = _find_or_node(ast.get_child(i))
        if f is not None:
            return f
    return None


def _extract_branch_text(br: Node) -> bytes:
    """dfa.mojo:3172-3194."""
    if br.type == ELEMENT and br.get_value():
        return br.get_value()
    if br.type == GROUP:
        out = b""
        for i in range(br.get_children_len()):
            ch = br.get_child(i)
            if ch.type == ELEMENT and ch.get_value():
                out += ch.get_value()
        return out
    return b""


def _is_quantified_group(ast: Node) -> bool:
    """dfa.mojo:3197-3242."""
    if ast.type == RE and ast.get_children_len() == 1:
        ch = ast.get_child(0)
        if ch.type == GROUP and ch.get_children_len() == 1:
            g = ch.get_child(0)
            if g.type == GROUP:
                if g.min != 1 or g.max != 1:
                    return _group_contains_only_literals(g)
    return False


def _extract_group_text(g: Node) -> bytes:
    """dfa.mojo:3245-3262."""
    out = b""
    for i in range(g.get_children_len()):
        ch = g.get_child(i)
        if ch.type == ELEMENT:
            out += ch.get_value()
    return out


def _collect_all_alternation_branches(or_node: Node) -> List[bytes]:
    """dfa.mojo:3265-3306."""
    branches: List[bytes] = []
    for i in range(or_node.get_children_len()):
        br = or_node.get_child(i)
        if br.type == OR:
            branches.extend(_collect_all_alternation_branches(br))
        elif br.type == GROUP:
            inner = br
            while inner.type == GROUP and inner.get_children_len() == 1:
                inner = inner.get_child(0)
            if inner.type == OR:
                branches.extend(_collect_all_alternation_branches(inner))
            else:
                t = _extract_branch_text(br)
                if len(t) > 0:
                    branches.append(t)
        else:
            t = _extract_branch_text(br)
            if len(t) > 0:
                branches.append(t)
    return branches


def _is_pure_alternation_pattern(ast: Node) -> bool:
    """dfa.mojo:3309-3342 (unwraps single-child GROUPs without looking at
    their quantifier -- SURVEY.md A.6 #1)."""
    if ast.type != RE or ast.get_children_len() != 1:
        return False
    child = ast.get_child(0)
    if child.type == OR:
        return _is_simple_alternation_branches(child)
    node = child
    while node.type == GROUP and node.get_children_len() == 1:
        node = node.get_child(0)
    if node.type == OR:
        return _is_simple_alternation_branches(node)
    return False


def _is_simple_quantifier_pattern(ast: Node) -> bool:
    """dfa.mojo:3345-3383."""
    if ast.type != RE or ast.get_children_len() != 1:
        return False
    g = ast.get_child(0)
    if g.type != GROUP or g.get_children_len() == 0:
        return False
    has_q = False
    for i in range(g.get_children_len()):
        ch = g.get_child(i)
        if ch.type != ELEMENT:
            return False
        if ((ch.min == 0 and ch.max == -1) or (ch.min == 1 and ch.max == -1)
                or (ch.min == 0 and ch.max == 1)):
            has_q = True
    return has_q


def _is_wildcard_quantifier_pattern(ast: Node) -> bool:
    """dfa.mojo:3386-3415."""
    if ast.type != RE or ast.get_children_len() != 1:
        return False
    g = ast.get_child(0)
    if g.type != GROUP or g.get_children_len() != 1:
        return False
    w = g.get_child(0)
    if w.type != WILDCARD:
        return False
    return ((w.min == 0 and w.max == -1) or (w.min == 1 and w.max == -1)
            or (w.min == 0 and w.max == 1) or (w.min == 1 and w.max == 1))


def _extract_literal_branches_strict(node: Node, branches: List[bytes]) -> bool:
    """dfa.mojo:3462-3486 and :3565-3589 (identical bodies)."""
    if node.type == OR:
        return (_extract_literal_branches_strict(node.get_child(0), branches)
                and _extract_literal_branches_strict(node.get_child(1), branches))
    if node.type == GROUP:
        t = b""
        for i in range(node.get_children_len()):
            e = node.get_child(i)
            if e.type != ELEMENT:
                return False
            t += e.get_value()
        branches.append(t)
        return True
    return False


def _extract_branches_lenient(node: Node, branches: List[bytes]):
    """dfa.mojo:1396-1418 and :1556-1581 (non-ELEMENT children skipped)."""
    if node.type == OR:
        _extract_branches_lenient(node.get_child(0), branches)
        _extract_branches_lenient(node.get_child(1), branches)
    elif node.type == GROUP:
        t = b""
        for i in range(node.get_children_len()):
            e = node.get_child(i)
            if e.type == ELEMENT:
                t += e.get_value()
        branches.append(t)


def _is_common_prefix_alternation_pattern(ast: Node) -> bool:
    """dfa.mojo:3418-3459."""
    if ast.type != RE or ast.get_children_len() != 1:
        return False
    outer = ast.get_child(0)
    if outer.type != GROUP or outer.get_children_len() != 1:
        return False
    inner = outer.get_child(0)
    if inner.type != GROUP or inner.get_children_len() != 1:
        return False
    or_node = inner.get_child(0)
    if or_node.type != OR:
        return False
    branches: List[bytes] = []
    if not _extract_literal_branches_strict(or_node, branches):
        return False
    if len(branches) < 2:
        return False
    return len(common_prefix(branches)) >= 2


def _is_quantified_alternation_group(ast: Node) -> bool:
    """dfa.mojo:3525-3562."""
    if ast.type != RE or ast.get_children_len() != 1:
        return False
    outer = ast.get_child(0)
    if outer.type != GROUP or outer.get_children_len() != 1:
        return False
    qg = outer.get_child(0)
    if qg.type != GROUP or qg.get_children_len() != 1:
        return False
    if qg.min == 1 and qg.max == 1:
        return False
    or_node = qg.get_child(0)
    if or_node.type != OR:
        return False
    return _extract_literal_branches_strict(or_node, [])
