"""ORACLE (test infrastructure) -- the reference's recursive backtracking matcher.

Restates NFAEngine of src/regex/nfa.mojo function by function:
  match_all                :169-340      match_first  :342-389     match_next :391-498
  match_next_with_groups   :500-574      (what regex.sub uses for \\1..\\9 on patterns
                                          outside the fixed-width group form, matcher.mojo:1781-1822)
  _match_node              :657-755      leaf matchers :757-1017   _match_or  :1019-1055
  _match_group             :1057-1103    _match_group_with_quantifier :1105-1156
  _match_sequence          :1158-1224    _match_with_backtracking     :1231-1311
  _try_match_count         :1313-1349    _apply_quantifier(+_simd)    :1375-1731
and the ASTNode helpers they call (src/regex/ast.mojo:372-513, 695-725) and the cached
"SIMD" matchers whose membership differs from the scalar tests (simd_matchers.mojo:129-147,
285-342: the whitespace nibble tables also accept 0x00 and ')' '*' '+' ',' '-').

Everything that looks odd is the reference's behaviour and is kept: groups are appended when a
group's sequence succeeds and never rolled back; a quantified leaf that is the LAST child of its
sequence fails on a non-matching byte even when its minimum is 0 (except \\d and \\w); the path a
quantifier takes depends on how much text is left (is_simd_optimizable is asked with max already
replaced by the remaining length); match_first_mode stops greedy runs 50 / 100 bytes past the start.
Pinned by the reference's own tests through tests/golden/reference_vectors.json (the vectors the
reference routes to this engine, and the regex.sub vectors with group references).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

from .frontend import (Node, RE, ELEMENT, WILDCARD, SPACE, DIGIT, WORD, RANGE, START, END, OR, GROUP)

Span = Tuple[int, int]
GroupMatch = Tuple[int, int, int]   # (group_id, start, end)

# ast.mojo:41-52
RK_LOWER, RK_UPPER, RK_DIGITS, RK_ALNUM, RK_ALPHA, RK_COMPLEX_ALNUM, RK_OTHER = 1, 2, 3, 4, 5, 6, 7
COMPLEX_CHAR_CLASS_THRESHOLD = 10


def _has_range_seq(pattern: bytes, lo: int, hi: int) -> bool:
    """ast.mojo:672-692."""
    for i in range(len(pattern) - 2):
        if pattern[i] == lo and pattern[i + 1] == 0x2D and pattern[i + 2] == hi:
            return True
    return False


def classify_range_kind(pattern: bytes) -> int:
    """ast.mojo:695-725."""
    if pattern == b"[a-z]":
        return RK_LOWER
    if pattern == b"[A-Z]":
        return RK_UPPER
    if pattern == b"[0-9]":
        return RK_DIGITS
    if pattern in (b"[a-zA-Z0-9]", b"[0-9a-zA-Z]"):
        return RK_ALNUM
    if pattern == b"[a-zA-Z]":
        return RK_ALPHA
    if pattern.startswith(b"[") and pattern.endswith(b"]"):
        if (len(pattern) - 2 > COMPLEX_CHAR_CLASS_THRESHOLD and _has_range_seq(pattern, 0x61, 0x7A)
                and _has_range_seq(pattern, 0x41, 0x5A) and _has_range_seq(pattern, 0x30, 0x39)):
            return RK_COMPLEX_ALNUM
    return RK_OTHER


def range_kind(node: Node) -> int:
    """RangeElement, ast.mojo:728-755: classified once from the raw [...] slice."""
    v = node.get_value()
    return classify_range_kind(v) if v else RK_OTHER


def _char_code_matches_range(ch: int, syntax: bytes) -> bool:
    """ast.mojo:480-508."""
    i = 1 if (len(syntax) > 0 and syntax[0] == 0x5E) else 0
    n = len(syntax)
    while i < n:
        if i + 2 < n and syntax[i + 1] == 0x2D:
            if syntax[i] <= ch <= syntax[i + 2]:
                return True
            i += 3
        else:
            if syntax[i] == ch:
                return True
            i += 1
    return False


def _is_char_in_range_by_code(ch: int, range_pattern: bytes) -> bool:
    """ast.mojo:464-478."""
    if range_pattern.startswith(b"["):
        return _char_code_matches_range(ch, range_pattern[1:len(range_pattern) - 1])
    return ch in range_pattern


def _is_lower(c): return 0x61 <= c <= 0x7A
def _is_upper(c): return 0x41 <= c <= 0x5A
def _is_digit(c): return 0x30 <= c <= 0x39
def _is_word(c): return _is_lower(c) or _is_upper(c) or _is_digit(c) or c == 0x5F
def _is_space5(c): return c in (0x20, 0x09, 0x0A, 0x0D, 0x0C)   # _match_space / is_match_char: no \v


def whitespace_matcher_contains(c: int) -> bool:
    """NibbleBasedMatcher.contains with _create_whitespace_matcher's tables
    (simd_matchers.mojo:129-147, 285-342): low nibble in {0, 9..D} AND high nibble in {0, 2}."""
    if c < 0 or c > 255:
        return False
    return ((c & 15) == 0 or 9 <= (c & 15) <= 13) and (c >> 4) in (0, 2)


def is_match_char(node: Node, ch: int, str_i: int = 0, str_len: int = 0) -> bool:
    """ASTNode.is_match_char, ast.mojo:415-462."""
    t = node.type
    if t == ELEMENT:
        v = node.get_value()
        return bool(v) and len(v) == 1 and v[0] == ch
    if t == WILDCARD:
        return ch != 0x0A
    if t == SPACE:
        return _is_space5(ch)
    if t == DIGIT:
        return _is_digit(ch)
    if t == WORD:
        return _is_word(ch)
    if t == RANGE:
        found = False
        v = node.get_value()
        if v:
            found = _is_char_in_range_by_code(ch, v)
        return not (found ^ node.positive_logic)
    if t == START:
        return str_i == 0
    if t == END:
        return str_i == str_len
    return False


def is_simd_optimizable(node: Node, min_matches: int, max_matches: int) -> bool:
    """ASTNode.is_simd_optimizable, ast.mojo:372-400."""
    if node.type not in (SPACE, DIGIT, WORD, RANGE):
        return False
    if min_matches == 1 and max_matches == 1:
        return False
    if max_matches == -1:
        if node.type in (DIGIT, WORD, SPACE):
            return min_matches >= 1
        return min_matches > 3
    if max_matches > 8:
        return True
    if node.type == RANGE and node.get_value():
        return len(node.get_value()) > 8
    return False


def _run(pred, text: bytes, str_i: int, min_matches: int, max_matches: int) -> Tuple[bool, int]:
    """apply_quantifier_simd_generic (simd_ops.mojo:1308-1354) and the _quantifier_*_loop helpers
    (nfa.mojo:1672-1731): count consecutive bytes that satisfy pred, at most max."""
    pos = str_i
    count = 0
    n = len(text)
    actual_max = max_matches if max_matches != -1 else n - str_i
    while pos < n and count < actual_max:
        if pred(text[pos]):
            count += 1
            pos += 1
        else:
            break
    if count >= min_matches:
        return (True, pos)
    return (False, str_i)


def _match_char_in_range(range_pattern: bytes, ch: int) -> bool:
    """NFAEngine._match_char_in_range, nfa.mojo:1650-1670."""
    if range_pattern.startswith(b"[") and range_pattern.endswith(b"]"):
        inner = range_pattern[1:len(range_pattern) - 1]
        if len(inner) == 3 and inner[1] == 0x2D:
            return inner[0] <= ch <= inner[2]
        return ch in inner
    return ch in range_pattern


def range_first_test(ast: Node, ch: int) -> bool:
    """The membership test _match_range makes on the byte at str_i (nfa.mojo:930-995): by range kind."""
    kind = range_kind(ast)
    found = False
    if kind == RK_ALNUM:
        found = _is_lower(ch) or _is_upper(ch) or _is_digit(ch)
    elif kind == RK_LOWER:
        found = _is_lower(ch)
    elif kind == RK_UPPER:
        found = _is_upper(ch)
    elif kind == RK_DIGITS:
        found = _is_digit(ch)
    elif kind == RK_ALPHA:
        found = _is_lower(ch) or _is_upper(ch)
    elif kind == RK_COMPLEX_ALNUM:
        if _is_lower(ch) or _is_upper(ch) or _is_digit(ch):
            found = True
        else:
            v = ast.get_value()
            if v:
                found = ch in v[1:len(v) - 1]
    else:
        v = ast.get_value()
        if v:
            found = _is_char_in_range_by_code(ch, v)
    return found == ast.positive_logic


def simd_predicate(ast: Node):
    """The byte predicate of _apply_quantifier_simd's loop for this leaf (nfa.mojo:1446-1647), None when that
    function falls off its end (`return (False, str_i)`)."""
    t = ast.type
    if t == DIGIT:
        return _is_digit
    if t == SPACE:
        return whitespace_matcher_contains
    if t == WORD:
        return _is_word
    v = ast.get_value()
    if t == RANGE and v:
        kind = range_kind(ast)
        if kind == RK_ALNUM:
            pred = lambda c: _is_lower(c) or _is_upper(c) or _is_digit(c)   # noqa: E731
        elif kind == RK_LOWER:
            pred = _is_lower
        elif kind == RK_UPPER:
            pred = _is_upper
        elif kind == RK_DIGITS:
            pred = _is_digit
        elif kind == RK_ALPHA:
            pred = lambda c: _is_lower(c) or _is_upper(c)   # noqa: E731
        elif kind == RK_COMPLEX_ALNUM:
            inner = v[1:len(v) - 1]
            pred = lambda c: _is_lower(c) or _is_upper(c) or _is_digit(c) or c in inner   # noqa: E731
        else:
            # RK_OTHER: _create_range_matcher returns None for every bracket pattern
            # (nfa.mojo:587-640), so the scalar fallback runs (:1620-1645)
            pred = lambda c: _match_char_in_range(v, c)   # noqa: E731
        return pred if ast.positive_logic else (lambda c: not pred(c))
    return None


class BacktrackNFA:
    """NFAEngine (nfa.mojo:66-1731).  `flags` carries the constructor's routing facts
    (literal_prefix, has_literal_optimization, starts/ends_with_dotstar; nfa.mojo:86-143)."""

    def __init__(self, pattern: bytes, ast: Optional[Node], flags):
        self.pattern = pattern
        self.regex = ast
        self.literal_prefix: bytes = flags.literal_prefix
        self.has_literal_optimization: bool = flags.has_literal_optimization
        self.pattern_len = len(pattern)
        self.ends_with_dotstar: bool = flags.ends_with_dotstar
        self.starts_with_dotstar: bool = flags.starts_with_dotstar
        self.is_prefix_literal = len(self.literal_prefix) > 0 and pattern.startswith(self.literal_prefix)

    # -- helpers ---------------------------------------------------------------------
    def _search_literal(self) -> bytes:
        """_get_search_literal_bytes, nfa.mojo:157-167."""
        return self.literal_prefix if self.has_literal_optimization else self.pattern

    def _find_last_literal(self, text: bytes, start: int) -> int:
        """nfa.mojo:577-585."""
        pos = text.rfind(self.literal_prefix)
        return pos if pos >= start else -1

    def _match_contains_literal(self, text: bytes, start: int, end: int) -> bool:
        """nfa.mojo:642-655."""
        if not self.has_literal_optimization or len(self.literal_prefix) == 0:
            return True
        pos = text.find(self.literal_prefix, start)
        return pos != -1 and pos + len(self.literal_prefix) <= end

    # -- public operations -----------------------------------------------------------
    def match_all(self, text: bytes) -> List[Span]:
        """nfa.mojo:169-340."""
        out: List[Span] = []
        if self.regex is None:
            return out
        ast = self.regex
        n = len(text)
        current_pos = 0
        if self.starts_with_dotstar and self.has_literal_optimization and text.find(b"\n") == -1:
            last_pos = self._find_last_literal(text, current_pos)
            if last_pos >= 0:
                out.append((current_pos, last_pos + len(self.literal_prefix)))
            return out
        if (self.ends_with_dotstar and self.has_literal_optimization and self.is_prefix_literal
                and text.find(b"\n") == -1):
            search = current_pos
            while search < n:
                pos = text.find(self.literal_prefix, search)
                if pos == -1:
                    break
                out.append((pos, n))
                break
            return out
        if self.has_literal_optimization:
            while current_pos <= n:
                literal_pos = text.find(self._search_literal(), current_pos)
                if literal_pos == -1:
                    break
                if literal_pos < current_pos:
                    current_pos = literal_pos + 1
                    continue
                try_pos = literal_pos
                search_window = 10
                if self.literal_prefix and not self.is_prefix_literal:
                    try_pos = max(current_pos, literal_pos - search_window)
                found = False
                max_search_positions = min(5, literal_pos - try_pos + 1)
                search_count = 0
                while try_pos <= literal_pos and try_pos <= n and search_count < max_search_positions:
                    tmp: List[GroupMatch] = []
                    ok, match_end = self._match_node(ast, text, try_pos, tmp, False, -1)
                    if ok and self._match_contains_literal(text, try_pos, match_end):
                        out.append((try_pos, match_end))
                        current_pos = try_pos + 1 if match_end == try_pos else match_end
                        found = True
                        break
                    try_pos += 1
                    search_count += 1
                if not found:
                    current_pos = literal_pos + 1
        else:
            while current_pos <= n:
                tmp = []
                ok, match_end = self._match_node(ast, text, current_pos, tmp, False, -1)
                if ok:
                    out.append((current_pos, match_end))
                    current_pos = current_pos + 1 if match_end == current_pos else match_end
                else:
                    current_pos += 1
        return out

    def match_first(self, text: bytes, start: int = 0) -> Optional[Span]:
        """nfa.mojo:342-389."""
        if self.regex is None:
            return None
        ok, end = self._match_node(self.regex, text, start, [], True, start)
        return (start, end) if ok else None

    def match_next(self, text: bytes, start: int = 0) -> Optional[Span]:
        """nfa.mojo:391-498."""
        if self.regex is None:
            return None
        ast = self.regex
        n = len(text)
        search_pos = start
        if self.starts_with_dotstar and self.has_literal_optimization and text.find(b"\n") == -1:
            last_pos = self._find_last_literal(text, start)
            if last_pos >= 0:
                return (start, last_pos + len(self.literal_prefix))
            return None
        if (self.ends_with_dotstar and self.has_literal_optimization and self.is_prefix_literal
                and text.find(b"\n") == -1):
            pos = text.find(self.literal_prefix, start)
            return (pos, n) if pos >= 0 else None
        if self.has_literal_optimization:
            while search_pos <= n:
                literal_pos = text.find(self._search_literal(), search_pos)
                if literal_pos == -1:
                    return None
                try_pos = literal_pos
                if self.literal_prefix and not self.is_prefix_literal:
                    try_pos = max(0, literal_pos - self.pattern_len)
                while try_pos <= literal_pos:
                    ok, match_end = self._match_node(ast, text, try_pos, [], False, -1)
                    if ok and self._match_contains_literal(text, try_pos, match_end):
                        return (try_pos, match_end)
                    try_pos += 1
                search_pos = literal_pos + 1
        else:
            while search_pos <= n:
                ok, end_idx = self._match_node(ast, text, search_pos, [], False, -1)
                if ok:
                    return (search_pos, end_idx)
                search_pos += 1
        return None

    def match_next_with_groups(self, text: bytes, start: int = 0) -> Tuple[Optional[Span], List[GroupMatch]]:
        """nfa.mojo:500-574."""
        if self.regex is None:
            return (None, [])
        ast = self.regex
        n = len(text)
        search_pos = start
        matches: List[GroupMatch] = []
        if self.has_literal_optimization:
            while search_pos <= n:
                literal_pos = text.find(self._search_literal(), search_pos)
                if literal_pos == -1:
                    return (None, [])
                try_pos = literal_pos
                if self.literal_prefix and not self.is_prefix_literal:
                    try_pos = max(0, literal_pos - self.pattern_len)
                while try_pos <= literal_pos:
                    matches.clear()
                    ok, match_end = self._match_node(ast, text, try_pos, matches, False, -1)
                    if ok and self._match_contains_literal(text, try_pos, match_end):
                        return ((try_pos, match_end), list(matches))
                    try_pos += 1
                search_pos = literal_pos + 1
        else:
            while search_pos <= n:
                matches.clear()
                ok, end = self._match_node(ast, text, search_pos, matches, False, -1)
                if ok:
                    return ((search_pos, end), list(matches))
                search_pos += 1
        return (None, [])

    # -- the recursive matcher ---------------------------------------------------------
    def _match_node(self, ast: Node, s: bytes, i: int, matches: List[GroupMatch], mfm: bool, req: int):
        """nfa.mojo:657-755."""
        t = ast.type
        if t == ELEMENT:
            return self._match_element(ast, s, i, mfm, req)
        if t == WILDCARD:
            return self._match_wildcard(ast, s, i, mfm, req)
        if t == SPACE:
            return self._match_space(ast, s, i, mfm, req)
        if t == DIGIT:
            return self._match_digit_or_word(ast, s, i, mfm, req, _is_digit)
        if t == WORD:
            return self._match_digit_or_word(ast, s, i, mfm, req, _is_word)
        if t == RANGE:
            return self._match_range(ast, s, i, mfm, req)
        if t == START:
            return (i == 0, i)
        if t == END:
            return (i == len(s), i)
        if t == OR:
            return self._match_or(ast, s, i, matches, mfm, req)
        if t == GROUP:
            return self._match_group(ast, s, i, matches, mfm, req)
        if t == RE:
            if not ast.has_children():           # _match_re, nfa.mojo:1351-1373
                return (True, i)
            return self._match_node(ast.get_child(0), s, i, matches, mfm, req)
        return (False, i)

    def _match_element(self, ast, s, i, mfm, req):
        """nfa.mojo:757-782."""
        if i >= len(s):
            return (False, i)
        v = ast.get_value()
        if v and v[0] == s[i]:
            return self._apply_quantifier(ast, s, i, 1, mfm, req)
        return (False, i)

    def _match_wildcard(self, ast, s, i, mfm, req):
        """nfa.mojo:785-806."""
        if i >= len(s):
            return (False, i)
        if s[i] != 0x0A:
            return self._apply_quantifier(ast, s, i, 1, mfm, req)
        return (False, i)

    def _match_space(self, ast, s, i, mfm, req):
        """nfa.mojo:809-838."""
        if i >= len(s):
            return (False, i)
        if _is_space5(s[i]):
            return self._apply_quantifier(ast, s, i, 1, mfm, req)
        return (False, i)

    def _match_digit_or_word(self, ast, s, i, mfm, req, pred):
        """_match_digit / _match_word, nfa.mojo:841-927: the two leaves that accept zero
        repetitions (min == 0) at the end of the text or on a non-matching byte."""
        if i >= len(s):
            if ast.min == 0:
                return self._apply_quantifier(ast, s, i, 0, mfm, req)
            return (False, i)
        if pred(s[i]):
            return self._apply_quantifier(ast, s, i, 1, mfm, req)
        if ast.min == 0:
            return self._apply_quantifier(ast, s, i, 0, mfm, req)
        return (False, i)

    def _match_range(self, ast, s, i, mfm, req):
        """nfa.mojo:930-995."""
        if i >= len(s):
            return (False, i)
        if range_first_test(ast, s[i]):
            return self._apply_quantifier(ast, s, i, 1, mfm, req)
        return (False, i)

    def _match_or(self, ast, s, i, matches, mfm, req):
        """nfa.mojo:1019-1055."""
        if ast.get_children_len() < 2:
            return (False, i)
        left = self._match_node(ast.get_child(0), s, i, matches, mfm, req)
        if left[0]:
            return left
        return self._match_node(ast.get_child(1), s, i, matches, mfm, req)

    @staticmethod
    def _has_quantifier(ast: Node) -> bool:
        return ast.min != 1 or ast.max != 1

    def _match_group(self, ast, s, i, matches, mfm, req):
        """nfa.mojo:1057-1103."""
        start_pos = i
        if self._has_quantifier(ast):
            return self._match_group_with_quantifier(ast, s, i, matches, mfm, req)
        result = self._match_sequence(ast, 0, s, i, matches, mfm, req)
        if not result[0]:
            return (False, i)
        if ast.capturing_group:
            gid = ast.group_id if ast.group_id >= 0 else 0
            matches.append((gid, start_pos, result[1]))
        return result

    def _match_group_with_quantifier(self, ast, s, i, matches, mfm, req):
        """nfa.mojo:1105-1156."""
        min_matches, max_matches = ast.min, ast.max
        current_pos = i
        group_matches = 0
        n = len(s)
        if max_matches == -1:
            max_matches = n - i
        while group_matches < max_matches and current_pos <= n:
            ok, pos = self._match_sequence(ast, 0, s, current_pos, matches, mfm, req)
            if ok:
                group_matches += 1
                current_pos = pos
                if mfm and req >= 0 and current_pos > req + 100:
                    break
                if ast.capturing_group:
                    gid = ast.group_id if ast.group_id >= 0 else 0
                    matches.append((gid, i, current_pos))
            else:
                break
        if group_matches >= min_matches:
            return (True, current_pos)
        return (False, i)

    def _match_sequence(self, parent, child_index, s, i, matches, mfm, req):
        """nfa.mojo:1158-1224."""
        children_len = parent.get_children_len()
        if child_index >= children_len:
            return (True, i)
        if child_index == children_len - 1:
            return self._match_node(parent.get_child(child_index), s, i, matches, mfm, req)
        first = parent.get_child(child_index)
        if self._has_quantifier(first):
            return self._match_with_backtracking(first, parent, child_index + 1, s, i, matches, mfm, req)
        result = self._match_node(first, s, i, matches, mfm, req)
        if not result[0]:
            return (False, i)
        return self._match_sequence(parent, child_index + 1, s, result[1], matches, mfm, req)

    def _match_with_backtracking(self, qnode, parent, remaining_index, s, i, matches, mfm, req):
        """nfa.mojo:1231-1311."""
        min_matches, max_matches = qnode.min, qnode.max
        if max_matches == -1:
            max_matches = len(s) - i
        if min_matches == max_matches:
            consumed = self._try_match_count(qnode, s, i, min_matches, mfm, req)
            if consumed >= 0:
                result = self._match_sequence(parent, remaining_index, s, i + consumed, matches, mfm, req)
                if result[0]:
                    return (True, result[1])
            return (False, i)
        match_count = max_matches
        while match_count >= min_matches:
            consumed = self._try_match_count(qnode, s, i, match_count, mfm, req)
            if consumed >= 0:
                new_pos = i + consumed
                if mfm and req >= 0 and new_pos > req + 100:
                    return (False, i)
                result = self._match_sequence(parent, remaining_index, s, new_pos, matches, mfm, req)
                if result[0]:
                    return (True, result[1])
            match_count -= 1
        return (False, i)

    @staticmethod
    def _try_match_count(ast, s, i, count, mfm, req) -> int:
        """nfa.mojo:1313-1349."""
        pos = i
        matched = 0
        n = len(s)
        while matched < count and pos < n:
            if mfm and req >= 0 and pos > req + 100:
                return -1
            if is_match_char(ast, s[pos], pos, n):
                matched += 1
                pos += 1
            else:
                return -1
        return pos - i if matched == count else -1

    def _apply_quantifier(self, ast, s, i, char_consumed, mfm, req):
        """nfa.mojo:1375-1443."""
        min_matches, max_matches = ast.min, ast.max
        n = len(s)
        if max_matches == -1:
            max_matches = n - i
        if min_matches == 1 and max_matches == 1:
            return (True, i + char_consumed)
        if is_simd_optimizable(ast, min_matches, max_matches):
            return self._apply_quantifier_simd(ast, s, i, min_matches, max_matches)
        count = 0
        pos = i
        while count < max_matches and pos < n:
            if mfm and req >= 0 and pos > req + 50:
                break
            if is_match_char(ast, s[pos], pos, n):
                count += 1
                pos += 1
            else:
                break
        if count >= min_matches:
            return (True, pos)
        return (False, i)

    def _apply_quantifier_simd(self, ast, s, i, min_matches, max_matches):
        """nfa.mojo:1446-1647."""
        pred = simd_predicate(ast)
        if pred is None:
            return (False, i)
        return _run(pred, s, i, min_matches, max_matches)
