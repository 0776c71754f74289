"""ORACLE (test infrastructure) -- pattern front end: bytes -> tokens -> AST.

Restates src/regex/lexer.mojo:61-195 (scan), src/regex/parser.mojo:56-114
(check_for_quantifiers), parser.mojo:130-464 (parse_token_list),
parser.mojo:467-510 (parse) and the node model of src/regex/ast.mojo:168-558.

Patterns and texts are ``bytes``; all offsets are byte offsets.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import List, Optional


class RegexSyntaxError(Exception):
    """A pattern the reference's lexer/parser raises on."""


# --- token kinds (src/regex/tokens.mojo:44-103) ------------------------------
T_ELEMENT = 0
T_WILDCARD = 1
T_SPACE = 2
T_DIGIT = 3
T_WORD = 4
T_START = 5
T_END = 6
T_COMMA = 8
T_LPAREN = 10
T_RPAREN = 11
T_LCURLY = 13
T_RCURLY = 14
T_LBRACKET = 16
T_RBRACKET = 17
T_ASTERISK = 22
T_PLUS = 23
T_QMARK = 24
T_VBAR = 26
T_NOTTOKEN = 27
T_CIRCUMFLEX = 28
T_DASH = 29


@dataclass
class Token:
    type: int
    char: int
    start_pos: int


_SIMPLE = {
    ord("."): T_WILDCARD,
    ord("("): T_LPAREN,
    ord(")"): T_RPAREN,
    ord("["): T_LBRACKET,
    ord("-"): T_DASH,
    ord("]"): T_RBRACKET,
    ord("$"): T_END,
    ord("?"): T_QMARK,
    ord("*"): T_ASTERISK,
    ord("+"): T_PLUS,
    ord("|"): T_VBAR,
    ord("}"): T_RCURLY,
}


def scan(regex: bytes) -> List[Token]:
    """lexer.mojo:61-195."""
    tokens: List[Token] = []
    i = 0
    escape_found = False
    n = len(regex)
    while i < n:
        ch = regex[i]
        if escape_found:
            # lexer.mojo:78-100
            if ch == ord("t"):
                tokens.append(Token(T_ELEMENT, 9, i - 1))
            elif ch == ord("s"):
                tokens.append(Token(T_SPACE, ch, i - 1))
            elif ch == ord("d"):
                tokens.append(Token(T_DIGIT, ch, i - 1))
            elif ch == ord("w"):
                tokens.append(Token(T_WORD, ch, i - 1))
            else:
                tokens.append(Token(T_ELEMENT, ch, i))
        elif ch == ord("\\"):
            escape_found = True
            i += 1
            continue
        elif ch == ord("{"):
            # lexer.mojo:129-153: quantifier sub-scan
            tokens.append(Token(T_LCURLY, ch, i))
            i += 1
            while i < n:
                c2 = regex[i]
                if c2 == ord(","):
                    tokens.append(Token(T_COMMA, c2, i))
                elif ord("0") <= c2 <= ord("9"):
                    tokens.append(Token(T_ELEMENT, c2, i))
                elif c2 == ord("}"):
                    tokens.append(Token(T_RCURLY, c2, i))
                    break
                else:
                    raise RegexSyntaxError("Bad token at index %d.{" % i)
                i += 1
        elif ch == ord("^"):
            # lexer.mojo:154-162
            if i == 0:
                tokens.append(Token(T_START, ch, i))
            else:
                tokens.append(Token(T_CIRCUMFLEX, ch, i))
        elif ch in _SIMPLE:
            tokens.append(Token(_SIMPLE[ch], ch, i))
        else:
            tokens.append(Token(T_ELEMENT, ch, i))
        escape_found = False
        i += 1
    return tokens


# --- AST (src/regex/ast.mojo:25-36) ------------------------------------------
RE = 0
ELEMENT = 1
WILDCARD = 2
SPACE = 3
DIGIT = 4
WORD = 5
RANGE = 6
START = 7
END = 8
OR = 9
NOT = 10
GROUP = 11

TYPE_NAMES = {
    RE: "RE", ELEMENT: "ELEMENT", WILDCARD: "WILDCARD", SPACE: "SPACE",
    DIGIT: "DIGIT", WORD: "WORD", RANGE: "RANGE", START: "START", END: "END",
    OR: "OR", NOT: "NOT", GROUP: "GROUP",
}


class Regex:
    """ast.mojo:83-165: the pattern plus the node arena (1-based child ids)."""

    def __init__(self, pattern: bytes):
        self.pattern = pattern
        self.children: List["Node"] = []

    def append_child(self, node: "Node") -> int:
        # value semantics: the arena holds a copy (ast.mojo:156-165)
        self.children.append(copy.copy(node))
        return len(self.children)  # 1-based index of the appended node


@dataclass
class Node:
    """ast.mojo:168-558."""

    type: int
    regex: Regex
    start_idx: int
    end_idx: int
    capturing_group: bool = False
    children_indexes: List[int] = field(default_factory=list)
    min: int = 0
    max: int = 0
    positive_logic: bool = True
    group_id: int = -1

    def __copy__(self):
        return Node(self.type, self.regex, self.start_idx, self.end_idx,
                    self.capturing_group, list(self.children_indexes),
                    self.min, self.max, self.positive_logic, self.group_id)

    def get_children_len(self) -> int:
        return len(self.children_indexes)

    def has_children(self) -> bool:
        return len(self.children_indexes) > 0

    def get_child(self, i: int) -> "Node":
        # ast.mojo:541-543
        return self.regex.children[self.children_indexes[i] - 1]

    def get_value(self) -> Optional[bytes]:
        # ast.mojo:546-558: the raw pattern slice; None when empty
        if self.start_idx == self.end_idx:
            return None
        return self.regex.pattern[self.start_idx:self.end_idx]

    def dump(self, depth: int = 0) -> str:
        v = self.get_value()
        s = "%s%s[%d:%d]{%d,%d}%s%s%s\n" % (
            "  " * depth, TYPE_NAMES[self.type], self.start_idx, self.end_idx,
            self.min, self.max, "" if self.positive_logic else "^",
            " cap=%d" % self.group_id if self.type == GROUP else "",
            " %r" % v if self.type in (ELEMENT, RANGE) else "")
        for i in range(self.get_children_len()):
            s += self.get_child(i).dump(depth + 1)
        return s


def _leaf(type_: int, regex: Regex, start: int, end: int, positive: bool = True) -> Node:
    return Node(type_, regex, start, end, min=1, max=1, positive_logic=positive)


def _group(regex: Regex, children: List[int], start: int, end: int,
           capturing: bool, group_id: int) -> Node:
    # ast.mojo:851-877
    return Node(GROUP, regex, start, end, capturing_group=capturing,
                children_indexes=list(children), min=1, max=1, group_id=group_id)


def _check_for_quantifiers(i: int, elem: Node, tokens: List[Token]) -> int:
    """parser.mojo:56-114.  Returns the updated token index ``i``."""
    nxt = tokens[i + 1]
    if nxt.type == T_ASTERISK:
        elem.min, elem.max = 0, -1
        i += 1
    elif nxt.type == T_PLUS:
        elem.min, elem.max = 1, -1
        i += 1
    elif nxt.type == T_QMARK:
        elem.min, elem.max = 0, 1
        i += 1
    elif nxt.type == T_LCURLY:
        i += 2
        min_val = max_val = 0
        has_min = has_max = False
        while i < len(tokens) and tokens[i].type == T_ELEMENT:
            d = tokens[i].char
            if ord("0") <= d <= ord("9"):
                min_val = min_val * 10 + (d - ord("0"))
                has_min = True
            else:
                raise RegexSyntaxError("Invalid digit in quantifier")
            i += 1
        elem.min = min_val if has_min else 0
        if i < len(tokens) and tokens[i].type == T_COMMA:
            i += 1
            while i < len(tokens) and tokens[i].type == T_ELEMENT:
                d = tokens[i].char
                if ord("0") <= d <= ord("9"):
                    max_val = max_val * 10 + (d - ord("0"))
                    has_max = True
                else:
                    raise RegexSyntaxError("Invalid digit in quantifier")
                i += 1
            elem.max = max_val if has_max else -1
        else:
            elem.max = elem.min
        if i < len(tokens) and tokens[i].type == T_RCURLY:
            i += 1
        i -= 1
    return i


class _Counter:
    def __init__(self):
        self.v = 0


def _parse_token_list(regex: Regex, tokens: List[Token], gc: _Counter) -> Node:
    """parser.mojo:130-464."""
    plen = len(regex.pattern)
    if len(tokens) == 0:
        return _group(regex, [], 0, 0, True, 0)

    # parser.mojo:150-214: split on the first top-level '|'
    paren_depth = 0
    for k, tk in enumerate(tokens):
        if tk.type == T_LPAREN:
            paren_depth += 1
        elif tk.type == T_RPAREN:
            paren_depth -= 1
        elif tk.type == T_VBAR and paren_depth == 0:
            left_tokens = tokens[:k]
            right_tokens = tokens[k + 1:]
            if left_tokens:
                left_ast = _parse_token_list(regex, left_tokens, gc)
            else:
                left_ast = _group(regex, [], 0, 0, True, 0)
            if right_tokens:
                right_ast = _parse_token_list(regex, right_tokens, gc)
            else:
                right_ast = _group(regex, [], 0, 0, True, 0)
            left_index = regex.append_child(left_ast)
            right_index = regex.append_child(right_ast)
            return Node(OR, regex, 0, plen, children_indexes=[left_index, right_index],
                        min=1, max=1)

    # parser.mojo:216-237: validation
    bracket_depth = 0
    paren_val = 0
    for tk in tokens:
        if tk.type == T_LBRACKET:
            bracket_depth += 1
        elif tk.type == T_RBRACKET:
            bracket_depth -= 1
            if bracket_depth < 0:
                raise RegexSyntaxError(
                    "Unescaped closing bracket ']' at position %d" % tk.start_pos)
        elif tk.type == T_LPAREN:
            paren_val += 1
        elif tk.type == T_RPAREN:
            paren_val -= 1
            if paren_val < 0:
                raise RegexSyntaxError(
                    "Unescaped closing parenthesis ')' at position %d" % tk.start_pos)

    elements: List[Node] = []
    i = 0
    nt = len(tokens)
    while i < nt:
        tk = tokens[i]
        if tk.type in (T_ELEMENT, T_DASH):
            # parser.mojo:246-255, 355-364
            elem = _leaf(ELEMENT, regex, tk.start_pos, tk.start_pos + 1)
            if i + 1 < nt:
                i = _check_for_quantifiers(i, elem, tokens)
            elements.append(elem)
        elif tk.type == T_WILDCARD:
            elem = _leaf(WILDCARD, regex, tk.start_pos, tk.start_pos + 1)
            if i + 1 < nt:
                i = _check_for_quantifiers(i, elem, tokens)
            elements.append(elem)
        elif tk.type in (T_SPACE, T_DIGIT, T_WORD):
            kind = {T_SPACE: SPACE, T_DIGIT: DIGIT, T_WORD: WORD}[tk.type]
            elem = _leaf(kind, regex, tk.start_pos, tk.start_pos + 2)
            if i + 1 < nt:
                i = _check_for_quantifiers(i, elem, tokens)
            elements.append(elem)
        elif tk.type == T_START:
            elements.append(_leaf(START, regex, tk.start_pos, tk.start_pos + 1))
        elif tk.type == T_END:
            elements.append(_leaf(END, regex, tk.start_pos, tk.start_pos + 1))
        elif tk.type == T_LBRACKET:
            # parser.mojo:313-354
            bracket_start = tk.start_pos
            i += 1
            positive = True
            if i < nt and tokens[i].type in (T_NOTTOKEN, T_CIRCUMFLEX):
                positive = False
                i += 1
            while i < nt and tokens[i].type != T_RBRACKET:
                if (i + 2 < nt and tokens[i + 1].type == T_DASH
                        and tokens[i + 2].type == T_ELEMENT):
                    i += 3
                else:
                    i += 1
            if i >= nt:
                raise RegexSyntaxError("Missing closing ']'.")
            bracket_end = tokens[i].start_pos + 1
            relem = _leaf(RANGE, regex, bracket_start, bracket_end, positive)
            if i + 1 < nt:
                i = _check_for_quantifiers(i, relem, tokens)
            elements.append(relem)
        elif tk.type == T_LPAREN:
            # parser.mojo:365-444
            paren_start = tk.start_pos
            i += 1
            is_capturing = True
            content_start = paren_start + 1
            if (i + 1 < nt and tokens[i].type == T_QMARK
                    and tokens[i + 1].type == T_ELEMENT
                    and tokens[i + 1].char == ord(":")):
                is_capturing = False
                i += 2
                content_start = paren_start + 3
            group_tokens: List[Token] = []
            paren_count = 1
            while i < nt and paren_count > 0:
                if tokens[i].type == T_LPAREN:
                    paren_count += 1
                elif tokens[i].type == T_RPAREN:
                    paren_count -= 1
                    if paren_count == 0:
                        break
                group_tokens.append(tokens[i])
                i += 1
            if paren_count > 0:
                raise RegexSyntaxError("Missing closing parenthesis ')'.")
            paren_end = tokens[i].start_pos
            gid = -1
            if is_capturing:
                gc.v += 1
                gid = gc.v
            group_ast = _parse_token_list(regex, group_tokens, gc)
            if group_ast.type == GROUP:
                group = group_ast
                group.capturing_group = is_capturing
                group.group_id = gid
                group.start_idx = content_start
                group.end_idx = paren_end
            else:
                child_index = regex.append_child(group_ast)
                group = _group(regex, [child_index], content_start, paren_end,
                               is_capturing, gid)
            if i + 1 < nt:
                i = _check_for_quantifiers(i, group, tokens)
            elements.append(group)
        # every other token kind is silently skipped (parser.mojo:446)
        i += 1

    children = []
    for e in elements:
        children.append(regex.append_child(e))
    return _group(regex, children, 0, plen, True, 0)


def parse(pattern: bytes) -> Node:
    """parser.mojo:467-510."""
    if isinstance(pattern, str):
        pattern = pattern.encode("latin-1")
    regex = Regex(pattern)
    tokens = scan(pattern)
    gc = _Counter()
    parsed = _parse_token_list(regex, tokens, gc)
    root_child = regex.append_child(parsed)
    return Node(RE, regex, 0, len(pattern), children_indexes=[root_child])
