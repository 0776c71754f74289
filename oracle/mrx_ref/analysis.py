"""ORACLE (test infrastructure) -- pattern analysis: complexity classifier,
literal helpers and literal extraction.

Restates src/regex/optimizer.mojo:103-999 (PatternAnalyzer, is_literal_pattern,
get_literal_string, pattern_has_anchors) and
src/regex/literal_optimizer.mojo:24-517 (LiteralSet, extract_literals,
has_literal_prefix).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

from .frontend import (Node, RE, ELEMENT, WILDCARD, SPACE, DIGIT, WORD, RANGE,
                       START, END, OR, GROUP)

SIMPLE, MEDIUM, COMPLEX = 0, 1, 2
COMPLEXITY_NAMES = {SIMPLE: "SIMPLE", MEDIUM: "MEDIUM", COMPLEX: "COMPLEX"}
MAX_LITERAL_QUANT_REPETITIONS = 10  # optimizer.mojo:33


# ---------------------------------------------------------------------------
# PatternAnalyzer (optimizer.mojo:103-845)
# ---------------------------------------------------------------------------
def classify(ast: Node) -> int:
    """optimizer.mojo:110-119."""
    return _analyze_node(ast, 0)


def count_simd_nodes(ast: Node) -> int:
    """optimizer.mojo:234-258."""
    count = 0
    if ast.type in (RANGE, DIGIT, WORD, SPACE):
        if ast.min > 1 or ast.max == -1:
            count += 2
        else:
            count += 1
    elif ast.type in (GROUP, RE, OR):
        for i in range(ast.get_children_len()):
            count += count_simd_nodes(ast.get_child(i))
    return count


def should_use_pure_dfa(ast: Node) -> bool:
    """optimizer.mojo:174-201 (the second test is unreachable, as upstream)."""
    if classify(ast) != SIMPLE:
        return False
    return count_simd_nodes(ast) <= 1


def _classify_quantifier(ast: Node) -> int:
    """optimizer.mojo:320-349."""
    if ast.min == 1 and ast.max == 1:
        return SIMPLE
    if (ast.min == 0 and ast.max == -1) or (ast.min == 1 and ast.max == -1):
        return SIMPLE
    if ast.min == 0 and ast.max == 1:
        return SIMPLE
    if ast.max != -1 and ast.max - ast.min <= 10:
        return SIMPLE
    if ast.max != -1 and ast.max - ast.min <= 100:
        return MEDIUM
    return COMPLEX


def _analyze_node(ast: Node, depth: int) -> int:
    """optimizer.mojo:260-318."""
    t = ast.type
    if t == RE:
        if ast.get_children_len() == 0:
            return SIMPLE
        return _analyze_node(ast.get_child(0), depth)
    if t in (ELEMENT, WILDCARD, SPACE, DIGIT, WORD, RANGE):
        return _classify_quantifier(ast)
    if t in (START, END):
        return SIMPLE
    if t == OR:
        return _analyze_alternation(ast, depth)
    if t == GROUP:
        if _is_multi_char_class_sequence(ast):
            return SIMPLE
        return _analyze_group(ast, depth)
    return COMPLEX


def _analyze_alternation(ast: Node, depth: int) -> int:
    """optimizer.mojo:351-401."""
    if depth > 2:
        if _is_literal_heavy_alternation(ast):
            return MEDIUM
        if depth <= 4 and _is_common_prefix_alternation_in_tree(ast):
            return SIMPLE
        return COMPLEX
    max_c = SIMPLE
    for i in range(ast.get_children_len()):
        c = _analyze_node(ast.get_child(i), depth + 1)
        if c == COMPLEX:
            return COMPLEX
        if c == MEDIUM:
            max_c = MEDIUM
    if (max_c == SIMPLE and ast.get_children_len() <= 8
            and not _has_nested_alternation(ast)):
        return SIMPLE
    return MEDIUM


def _analyze_group(ast: Node, depth: int) -> int:
    """optimizer.mojo:403-490."""
    if depth > 4:
        return COMPLEX
    qc = _classify_quantifier(ast)
    if qc == COMPLEX:
        return COMPLEX
    if ast.min != 1 or ast.max != 1:
        if _is_simple_quantified_group(ast):
            pass
        elif _is_quantified_alternation_group_in_optimizer(ast):
            return SIMPLE
        else:
            return MEDIUM
    if ast.get_children_len() == 1:
        only = ast.get_child(0)
        if only.type in (OR, GROUP):
            if _is_all_literal_branches(only):
                return SIMPLE
    max_child = SIMPLE
    for i in range(ast.get_children_len()):
        c = _analyze_node(ast.get_child(i), depth + 1)
        if c == COMPLEX:
            return COMPLEX
        if c == MEDIUM:
            max_child = MEDIUM
    if max_child == SIMPLE and qc == SIMPLE:
        all_literal = True
        for i in range(ast.get_children_len()):
            ch = ast.get_child(i)
            is_lit = (ch.type == ELEMENT and ch.min == ch.max
                      and 1 <= ch.min <= MAX_LITERAL_QUANT_REPETITIONS)
            if not (is_lit or ch.type == START or ch.type == END):
                all_literal = False
                break
        if all_literal and ast.get_children_len() <= 20:
            return SIMPLE
        if ast.get_children_len() <= 5:
            return SIMPLE
        return MEDIUM
    return MEDIUM


def _is_all_literal_branches(ast: Node) -> bool:
    """optimizer.mojo:492-510."""
    if ast.type == GROUP:
        if ast.get_children_len() == 1:
            return _is_all_literal_branches(ast.get_child(0))
        for j in range(ast.get_children_len()):
            if ast.get_child(j).type != ELEMENT:
                return False
        return True
    if ast.type == OR:
        for i in range(ast.get_children_len()):
            if not _is_all_literal_branches(ast.get_child(i)):
                return False
        return True
    return ast.type == ELEMENT


def _is_multi_char_class_sequence(ast: Node) -> bool:
    """optimizer.mojo:512-555."""
    if ast.type != GROUP or ast.get_children_len() < 2:
        return False
    cc = 0
    for i in range(ast.get_children_len()):
        e = ast.get_child(i)
        if e.type in (RANGE, DIGIT, WORD, SPACE, WILDCARD):
            cc += 1
        elif e.type == ELEMENT and e.min == 1 and e.max == 1:
            pass
        else:
            return False
    return cc >= 2


def _is_simple_quantified_group(ast: Node) -> bool:
    """optimizer.mojo:557-586."""
    if not ((ast.min == 0 and ast.max == 1) or (ast.min == 0 and ast.max == -1)
            or (ast.min == 1 and ast.max == -1)):
        return False
    for i in range(ast.get_children_len()):
        ch = ast.get_child(i)
        if ch.type != ELEMENT or ch.min != 1 or ch.max != 1:
            return False
    return True


def _has_nested_alternation(ast: Node) -> bool:
    """optimizer.mojo:588-606."""
    for i in range(ast.get_children_len()):
        ch = ast.get_child(i)
        if ch.type == OR:
            if _has_nested_alternation(ch):
                return True
        elif ch.type == GROUP:
            if _group_contains_or(ch):
                return True
    return False


def _group_contains_or(ast: Node) -> bool:
    """optimizer.mojo:608-619."""
    for i in range(ast.get_children_len()):
        ch = ast.get_child(i)
        if ch.type == OR:
            return True
        if ch.type == GROUP and _group_contains_or(ch):
            return True
    return False


def _extract_literal_branches_tree(node: Node, branches: List[bytes]) -> bool:
    """optimizer.mojo:648-673 and :743-768 (identical bodies)."""
    if node.type == OR:
        return (_extract_literal_branches_tree(node.get_child(0), branches)
                and _extract_literal_branches_tree(node.get_child(1), branches))
    if node.type == GROUP:
        text = b""
        for i in range(node.get_children_len()):
            e = node.get_child(i)
            if e.type != ELEMENT:
                return False
            text += e.get_value()
        branches.append(text)
        return True
    return False


def common_prefix(branches: List[bytes]) -> bytes:
    """optimizer.mojo:675-710 (same routine as dfa.mojo:1465-1498, 3489-3522)."""
    if not branches:
        return b""
    if len(branches) == 1:
        return branches[0]
    first = branches[0]
    min_len = min(len(b) for b in branches)
    out = b""
    for pos in range(min_len):
        c = first[pos]
        if all(b[pos] == c for b in branches[1:]):
            out += bytes([c])
        else:
            break
    return out


def _is_common_prefix_alternation_in_tree(ast: Node) -> bool:
    """optimizer.mojo:621-646."""
    branches: List[bytes] = []
    if not _extract_literal_branches_tree(ast, branches):
        return False
    if len(branches) < 2:
        return False
    return len(common_prefix(branches)) >= 2


def _is_quantified_alternation_group_in_optimizer(ast: Node) -> bool:
    """optimizer.mojo:712-741."""
    if ast.min == 1 and ast.max == 1:
        return False
    if ast.get_children_len() != 1:
        return False
    or_node = ast.get_child(0)
    if or_node.type != OR:
        return False
    return _extract_literal_branches_tree(or_node, [])


def _is_literal_heavy_alternation(ast: Node) -> bool:
    """optimizer.mojo:770-792."""
    if ast.type != OR:
        return False
    total = ast.get_children_len()
    ok = sum(1 for i in range(total) if _is_dfa_compatible_branch(ast.get_child(i)))
    return ok * 5 >= total * 4


def _is_dfa_compatible_branch(ast: Node) -> bool:
    """optimizer.mojo:794-825."""
    if ast.type == ELEMENT:
        return True
    if ast.type in (RANGE, DIGIT, WORD, SPACE):
        return True
    if ast.type == GROUP:
        if ast.get_children_len() <= 4:
            for i in range(ast.get_children_len()):
                if not _is_simple_dfa_node(ast.get_child(i)):
                    return False
            return True
    elif ast.type == OR:
        if ast.get_children_len() <= 4:
            for i in range(ast.get_children_len()):
                if not _is_dfa_compatible_branch(ast.get_child(i)):
                    return False
            return True
    return False


def _is_simple_dfa_node(ast: Node) -> bool:
    """optimizer.mojo:827-845."""
    if ast.type in (ELEMENT, RANGE, DIGIT, WORD, SPACE, WILDCARD):
        return ast.max <= 10 or ast.max == -1
    return ast.type in (START, END)


# ---------------------------------------------------------------------------
# literal-pattern helpers (optimizer.mojo:848-999)
# ---------------------------------------------------------------------------
def is_literal_pattern(ast: Node) -> bool:
    """optimizer.mojo:848-863."""
    if ast.type != RE:
        return False
    if not ast.has_children():
        return True
    return _is_literal_sequence(ast.get_child(0))


def _is_literal_sequence(ast: Node) -> bool:
    """optimizer.mojo:866-900."""
    if ast.type == ELEMENT:
        if ast.min == 1 and ast.max == 1:
            return True
        return ast.min == ast.max and 1 <= ast.min <= MAX_LITERAL_QUANT_REPETITIONS
    if ast.type in (START, END):
        return True
    if ast.type == GROUP:
        for i in range(ast.get_children_len()):
            ch = ast.get_child(i)
            if ch.type == GROUP:
                return False
            if not _is_literal_sequence(ch):
                return False
        return True
    return False


def get_literal_string(ast: Node) -> bytes:
    """optimizer.mojo:903-918."""
    if ast.type == RE and ast.get_children_len() > 0:
        return _extract_literal_chars(ast.get_child(0))
    return b""


def _extract_literal_chars(ast: Node) -> bytes:
    """optimizer.mojo:921-953."""
    if ast.type == ELEMENT:
        v = ast.get_value()
        if not v:
            return b""
        if ast.min <= 1:
            return v
        return v * ast.min
    if ast.type == GROUP:
        return b"".join(_extract_literal_chars(ast.get_child(i))
                        for i in range(ast.get_children_len()))
    return b""


def pattern_has_anchors(ast: Node) -> Tuple[bool, bool]:
    """optimizer.mojo:956-999."""
    if ast.type == RE and ast.has_children():
        return _check_anchors_recursive(ast.get_child(0))
    return (False, False)


def _check_anchors_recursive(ast: Node) -> Tuple[bool, bool]:
    if ast.type == START:
        return (True, False)
    if ast.type == END:
        return (False, True)
    if ast.type == GROUP:
        hs = he = False
        for i in range(ast.get_children_len()):
            s, e = _check_anchors_recursive(ast.get_child(i))
            hs = hs or s
            he = he or e
        return (hs, he)
    return (False, False)


# ---------------------------------------------------------------------------
# literal extraction (literal_optimizer.mojo)
# ---------------------------------------------------------------------------
@dataclass
class LiteralInfo:
    literal: bytes
    start_offset: int
    is_prefix: bool
    is_suffix: bool
    is_required: bool


class LiteralSet:
    """literal_optimizer.mojo:119-214."""

    def __init__(self):
        self.literals: List[LiteralInfo] = []
        self.best_idx: Optional[int] = None

    def add(self, literal: bytes, start_offset=0, is_prefix=False, is_suffix=False,
            is_required=True) -> LiteralInfo:
        info = LiteralInfo(literal, start_offset, is_prefix, is_suffix, is_required)
        self.literals.append(info)
        return info

    def select_best(self):
        # literal_optimizer.mojo:166-206
        if not self.literals:
            self.best_idx = None
            return
        best_idx, best_score = 0, 0
        for i, lit in enumerate(self.literals):
            score = 0
            if lit.is_required:
                score += 1000
            score += len(lit.literal) * 10
            if lit.is_prefix:
                score += 100
            if lit.is_suffix:
                score += 100
            score += lit.start_offset
            if score > best_score:
                best_score, best_idx = score, i
        self.best_idx = best_idx

    def get_best_literal(self) -> Optional[LiteralInfo]:
        if self.best_idx is not None:
            return self.literals[self.best_idx]
        return None


def extract_literals(ast: Node) -> LiteralSet:
    """literal_optimizer.mojo:217-238."""
    result = LiteralSet()
    if ast.type == RE and ast.has_children():
        _extract_from_node(ast.get_child(0), result, 0, True, True)
    elif ast.type == GROUP:
        _extract_from_node(ast, result, 0, True, True)
    result.select_best()
    return result


def _extract_from_node(node: Node, result: LiteralSet, offset: int,
                       is_required: bool, at_start: bool):
    """literal_optimizer.mojo:241-330."""
    if node.type == ELEMENT:
        if node.min >= 1 and node.get_value():
            if node.max == 1:
                result.add(node.get_value(), offset, at_start, False, is_required)
            elif node.max == -1:
                if node.min >= 1:
                    result.add(node.get_value(), offset, at_start, False, is_required)
    elif node.type == GROUP:
        if node.min >= 1:
            if node.get_children_len() == 1:
                child = node.get_child(0)
                if child.type in (GROUP, OR):
                    _extract_from_node(child, result, offset, is_required, at_start)
                    return
            _extract_sequence(node, offset, is_required, at_start, result)
    elif node.type == OR:
        cp = _find_common_prefix_simple(node)
        if len(cp) > 0:
            result.add(cp, offset, at_start, False, True)
        for i in range(node.get_children_len()):
            _extract_from_node(node.get_child(i), result, offset, False, at_start)


def _extract_sequence(group: Node, start_offset: int, is_required: bool,
                      at_start: bool, literals: LiteralSet):
    """literal_optimizer.mojo:333-394."""
    cur = b""
    cur_off = start_offset
    seq_at_start = at_start
    for i in range(group.get_children_len()):
        ch = group.get_child(i)
        if ch.type == ELEMENT and ch.min == 1 and ch.max == 1 and ch.get_value():
            cur += ch.get_value()
        else:
            if len(cur) > 0:
                literals.add(cur, cur_off, seq_at_start, False, is_required)
                cur_off += len(cur)
                seq_at_start = False
                cur = b""
            if ch.type in (START, END):
                continue
            seq_at_start = False
            if ch.min > 0:
                cur_off += 1
    if len(cur) > 0:
        literals.add(cur, cur_off, seq_at_start, False, is_required)


def _find_common_prefix_simple(or_node: Node) -> bytes:
    """literal_optimizer.mojo:397-413."""
    prefixes: List[bytes] = []
    _collect_or_prefixes(or_node, prefixes)
    if len(prefixes) < 2:
        return b""
    common = prefixes[0]
    for p in prefixes[1:]:
        n = 0
        while n < min(len(common), len(p)) and common[n] == p[n]:
            n += 1
        common = common[:n]
        if not common:
            return b""
    return common


def _collect_or_prefixes(node: Node, prefixes: List[bytes]):
    """literal_optimizer.mojo:416-427."""
    if node.type != OR:
        p = _get_prefix_literal(node)
        if len(p) > 0:
            prefixes.append(p)
        return
    for i in range(node.get_children_len()):
        _collect_or_prefixes(node.get_child(i), prefixes)


def _get_prefix_literal(node: Node) -> bytes:
    """literal_optimizer.mojo:430-447."""
    if node.type == ELEMENT and node.min >= 1 and node.max >= 1 and node.get_value():
        return node.get_value()
    if node.type == GROUP and node.min >= 1:
        ls = LiteralSet()
        _extract_sequence(node, 0, True, True, ls)
        if ls.literals:
            return ls.literals[0].literal
    return b""


def has_literal_prefix(ast: Node) -> bool:
    """literal_optimizer.mojo:463-476."""
    if ast.type != RE or not ast.has_children():
        return False
    return _has_literal_prefix_node(ast.get_child(0))


def _has_literal_prefix_node(node: Node) -> bool:
    """literal_optimizer.mojo:479-497."""
    if node.type == START:
        return False
    if node.type == ELEMENT:
        return node.min >= 1 and node.max >= 1
    if node.type == GROUP:
        if node.min >= 1 and node.get_children_len() > 0:
            first = node.get_child(0)
            if first.type == START and node.get_children_len() > 1:
                return _has_literal_prefix_node(node.get_child(1))
            return _has_literal_prefix_node(first)
        return False
    return False
