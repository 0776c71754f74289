"""ORACLE (test infrastructure) -- engine routing and the public API.

Restates src/regex/matcher.mojo: NFAMatcher ordering (:273-431),
_is_simple_pattern_skip_prefilter (:447-532), HybridMatcher (:535-926),
CompiledRegex (:929-1163), module functions search/findall/split/match_first
(:1325-1415) and sub with its template and fixed-width-group fast path
(:1418-1917); NFAEngine's constructor flags (src/regex/nfa.mojo:86-143).

The recursive backtracking interpreter (nfa.mojo:657-1769) is restated in
backtrack.py; calls the reference routes to it are answered from there.

LazyDFA search / findall / sub of '$' programs (matcher.mojo:401-431 reach the LazyDFA before the OnePass
branch that was meant for them): upstream results depend on what the transition cache holds from EARLIER
calls -- a transition first computed while the last byte of some text was consumed carries "'$' holds" to
every later use, one first computed elsewhere never sees '$' (pikevm.mojo:697-700, 869-942).  A batch has
no call order, so the contract restated here is: every text is answered as a freshly compiled pattern
would answer it -- CompiledRegex empties the cache at the start of every public call (FRESH_LAZY_CACHE;
False = the cache lives as long as the object, as upstream).  Within one call (the walks of one findall,
the match_next calls of one sub) the cache carries over exactly as upstream.
"""
from __future__ import annotations

import os

from typing import List, Optional, Tuple

from .frontend import parse, Node, RE, GROUP, ELEMENT, START, END
from .analysis import (classify, should_use_pure_dfa, extract_literals,
                       has_literal_prefix, SIMPLE, COMPLEXITY_NAMES)
from .dfa_engine import compile_dfa_pattern, DFACompileError, DFAEngine
from .pikevm import compile_ast, PikeVMEngine, LazyDFA, OP_END_ANCHOR
from .onepass import compile_onepass
from .backtrack import BacktrackNFA

Span = Tuple[int, int]


FRESH_LAZY_CACHE = True   # module docstring: '$' programs on the LazyDFA search


class UnsupportedByOracle(Exception):
    """The reference would run an engine that is out of this repo's scope."""


class ReferenceDoesNotTerminate(Exception):
    """CompiledRegex.sub's loop (matcher.mojo:1747-1822) got a match that does not move `pos` forward:
    NFAEngine's literal prefilter backs up to literal_pos - len(pattern) without looking at `start`
    (nfa.mojo:441-447, 531-533), so the same match can come back for ever.  There is no result to restate."""


def _is_simple_pattern_skip_prefilter(pattern: bytes) -> bool:
    """matcher.mojo:447-532."""
    n = len(pattern)
    if n <= 4:
        return True
    has_q = has_alt = has_anchor = has_wild = has_cc = False
    for i, c in enumerate(pattern):
        if c in b"*+?":
            has_q = True
        elif c == ord("|"):
            has_alt = True
        elif c == ord("^") and i == 0:
            has_anchor = True
        elif c == ord("$") and i == n - 1:
            has_anchor = True
        elif c == ord("."):
            has_wild = True
        elif c == ord("["):
            has_cc = True
    if has_q and not has_alt and not has_wild:
        return True
    if has_alt and n <= 10 and not has_wild:
        if pattern.count(b"|") >= n // 3:
            return True
    if has_anchor and n <= 10:
        return True
    if has_wild and n >= 8:
        return False
    if has_alt and n >= 12:
        return False
    if n >= 15:
        return False
    return True


def check_ast_for_anchors(ast: Node) -> bool:
    """matcher.mojo:168-178."""
    if ast.type in (START, END):
        return True
    if ast.type in (GROUP, RE):
        return any(check_ast_for_anchors(ast.get_child(i))
                   for i in range(ast.get_children_len()))
    return False


def _find_rare_required_byte(ast: Node, lookup: List[int]) -> int:
    """matcher.mojo:122-165."""
    node = ast
    if node.type == RE and node.get_children_len() == 1:
        node = node.get_child(0)
    if node.type != GROUP:
        return -1
    for i in range(node.get_children_len()):
        ch = node.get_child(i)
        if ch.type != ELEMENT:
            continue
        if ch.min < 1:
            continue
        v = ch.get_value()
        if not v or len(v) != 1:
            continue
        if lookup[v[0]] == 0:
            return v[0]
    return -1


class NFAEngineFlags:
    """The parts of NFAEngine.__init__ (nfa.mojo:86-143) that steer routing."""

    def __init__(self, pattern: bytes):
        self.literal_prefix = b""
        self.has_literal_optimization = False
        try:
            ast = parse(pattern)
            ls = extract_literals(ast)
            best = ls.get_best_literal()
            if best is not None:
                if best.is_prefix and best.is_required and len(best.literal) >= 1:
                    self.literal_prefix = best.literal
                    self.has_literal_optimization = True
                elif best.is_required and len(best.literal) >= 3:
                    self.literal_prefix = best.literal
                    self.has_literal_optimization = True
        except Exception:
            pass
        self.ends_with_dotstar = pattern.endswith(b".*") and not pattern.endswith(b"\\.*")
        swd = False
        if pattern.startswith(b".*"):
            swd = True
            if len(pattern) > 2 and pattern[2:3] in (b"?", b"*", b"+"):
                swd = False
        self.starts_with_dotstar = swd


# The backtracking matcher's C twin (oracle/c/mrx_backtrack.c through cbacktrack.py): same results, affordable
# on kilobyte texts.  Off by default -- the Python restatement is what the reference's vectors pin;
# tests/big_fuzz.py switches it on for long texts (MRX_ORACLE_C_BACKTRACK=1 does the same).
USE_C_BACKTRACK = os.environ.get("MRX_ORACLE_C_BACKTRACK") == "1"


def _make_backtrack(pattern: bytes, ast, flags):
    bt = BacktrackNFA(pattern, ast, flags)
    if USE_C_BACKTRACK:
        from .cbacktrack import CBacktrack
        return CBacktrack(bt)
    return bt


def nfa_engine(pattern: bytes) -> BacktrackNFA:
    """NFAEngine(pattern) on its own (nfa.mojo:86-143), as regex.nfa's module functions and the
    reference's tests/test_nfa.mojo use it: no hybrid router, no LazyDFA, no OnePass in front."""
    try:
        own_ast = parse(pattern)
    except Exception:
        own_ast = None
    return _make_backtrack(pattern, own_ast, NFAEngineFlags(pattern))


def nfa_findall(pattern: bytes, text: bytes):
    """regex.nfa.findall, nfa.mojo:1733-1747."""
    return nfa_engine(pattern).match_all(text)


def nfa_match_first(pattern: bytes, text: bytes):
    """regex.nfa.match_first, nfa.mojo:1750-1769: NFAEngine.match_first(text, 0), kept only at 0."""
    r = nfa_engine(pattern).match_first(text, 0)
    return r if (r is not None and r[0] == 0) else None


class NFAMatcher:
    """matcher.mojo:273-431."""

    def __init__(self, ast: Node, pattern: bytes):
        self.engine = NFAEngineFlags(pattern)
        try:   # NFAEngine parses the pattern itself; a failure leaves it without an AST (nfa.mojo:98-129)
            own_ast = parse(pattern)
        except Exception:
            own_ast = None
        self.backtrack = _make_backtrack(pattern, own_ast, self.engine)   # NFAMatcher.engine's matching half
        vm = PikeVMEngine(compile_ast(ast))
        self.program = vm.program
        self.onepass = None   # OnePassNFA, only for '$' programs that compile one-pass (:310-313)
        self.lazy: Optional[LazyDFA] = None
        if vm.is_supported():
            if any(ins[0] == OP_END_ANCHOR for ins in vm.program.instructions):
                self.onepass = compile_onepass(vm.program)
            self.lazy = LazyDFA(vm)

    def _nfa_fast_paths_absent(self) -> bool:
        e = self.engine
        return (not e.has_literal_optimization and not e.starts_with_dotstar
                and not e.ends_with_dotstar)

    def _use_lazy_dfa_for_search(self) -> bool:
        return self.lazy is not None and self._nfa_fast_paths_absent()

    def match_first(self, text: bytes, start: int = 0):
        if self.lazy is not None and not self.lazy.has_end_anchor:
            return self.lazy.match_first(text, start)
        if self.onepass is not None:
            return self.onepass.match_first(text, start)
        return self.backtrack.match_first(text, start)

    def fresh_cache(self):
        if self.lazy is not None and self.lazy.has_end_anchor:
            self.lazy.reset()

    def match_next(self, text: bytes, start: int = 0):
        if self._use_lazy_dfa_for_search():
            return self.lazy.match_next(text, start)   # ('$' programs: see the module docstring)
        # (_use_onepass_for_search needs the LazyDFA branch above to have been taken: unreachable)
        return self.backtrack.match_next(text, start)

    def match_all(self, text: bytes):
        if self._use_lazy_dfa_for_search():
            return self.lazy.match_all(text)
        return self.backtrack.match_all(text)


class HybridMatcher:
    """matcher.mojo:535-926."""

    def __init__(self, pattern: bytes, force_nfa: bool = False):
        # force_nfa: behave as if DFAEngine compilation had failed (matcher.mojo:666-672);
        # the "LazyDFA semantics" switch of SURVEY.md 8(c), never the default.
        self.pattern = pattern
        self.is_wildcard_match_any = pattern == b".*"
        self.best_literal: Optional[bytes] = None
        self.literal_has_anchors = False
        self.is_exact_literal = False
        self.prefilter_literal: Optional[bytes] = None
        self.dfa: Optional[DFAEngine] = None
        self.nfa_matcher: Optional[NFAMatcher] = None
        self.use_dfa = False
        self.required_byte = -1
        self.complexity = SIMPLE
        self.use_pure_dfa = False
        if self.is_wildcard_match_any:
            return
        ast = parse(pattern)
        self.ast = ast
        self.complexity = classify(ast)
        self.use_pure_dfa = should_use_pure_dfa(ast)
        should_analyze = (not _is_simple_pattern_skip_prefilter(pattern)
                          and not self.use_pure_dfa)
        if should_analyze:
            ls = extract_literals(ast)
            has_anchors = check_ast_for_anchors(ast)
            best = ls.get_best_literal()
            is_exact = False
            if best is not None and len(best.literal) > 0:
                self.best_literal = best.literal
                has_ops = any(c in pattern for c in b"*+?.|([{")
                is_exact = (best.is_required and has_literal_prefix(ast)
                            and not has_anchors and not has_ops)
            self.literal_has_anchors = has_anchors
            self.is_exact_literal = is_exact
            if ord("|") not in pattern:
                # create_optimized_prefilter, matcher.mojo:99-108
                if self.best_literal is not None and len(self.best_literal) >= 2:
                    self.prefilter_literal = self.best_literal
        self.nfa_matcher = NFAMatcher(ast, pattern)
        if self.complexity == SIMPLE and not force_nfa:
            try:
                self.dfa = compile_dfa_pattern(ast)
                self.use_dfa = True
            except DFACompileError:
                self.dfa = None
                self.use_dfa = False
        if (self.use_dfa and not self.literal_has_anchors
                and self.dfa.has_simd_matcher):
            self.required_byte = _find_rare_required_byte(ast, self.dfa.matcher.lookup)

    def get_engine_type(self) -> str:
        """matcher.mojo:900-918."""
        base = "DFA" if self.use_dfa else "NFA"
        if self.is_exact_literal and not self.literal_has_anchors:
            return base + "+ExactLiteral"
        if self.prefilter_literal is not None and not self.literal_has_anchors:
            return base + "+Prefilter"
        return base

    def is_match(self, text: bytes, start: int = 0) -> bool:
        """matcher.mojo:721-731."""
        if self.is_wildcard_match_any:
            return start <= len(text)
        if self.use_dfa:
            return self.dfa.is_match(text, start)
        return self.nfa_matcher.match_first(text, start) is not None

    def match_first(self, text: bytes, start: int = 0) -> Optional[Span]:
        """matcher.mojo:733-753."""
        if self.is_wildcard_match_any:
            return (start, len(text)) if start <= len(text) else None
        if self.use_dfa:
            return self.dfa.match_first(text, start)
        return self.nfa_matcher.match_first(text, start)

    def match_next(self, text: bytes, start: int = 0) -> Optional[Span]:
        """matcher.mojo:755-802."""
        if self.is_wildcard_match_any:
            return (start, len(text)) if start <= len(text) else None
        if self.is_exact_literal and not self.literal_has_anchors:
            lit = self.best_literal
            if lit is not None:
                if start >= len(text):
                    return None
                pos = text.find(lit, start)
                if pos != -1:
                    end = pos + len(lit)
                    if end <= len(text):
                        return (pos, end)
                return None
        if self.prefilter_literal is not None and not self.literal_has_anchors:
            # MemchrPrefilter.find_first_candidate, prefilter.mojo:420-430
            if start >= len(text):
                return None
            cand = text.find(self.prefilter_literal, start)
            if cand == -1:
                return None
            if self.use_dfa:
                return self.dfa.match_next(text, cand)
            return self.nfa_matcher.match_next(text, cand)
        if self.use_dfa:
            return self.dfa.match_next(text, start)
        return self.nfa_matcher.match_next(text, start)

    def match_all(self, text: bytes) -> List[Span]:
        """matcher.mojo:804-862."""
        if self.is_wildcard_match_any:
            return [(0, len(text))]
        if self.is_exact_literal and not self.literal_has_anchors:
            out: List[Span] = []
            lit = self.best_literal
            if lit is not None:
                ll, tl = len(lit), len(text)
                if ll > tl:
                    return out
                start = 0
                max_start = tl - ll
                while start <= max_start:
                    pos = text.find(lit, start)
                    if pos == -1:
                        break
                    if pos + ll <= tl:
                        out.append((pos, pos + ll))
                    start = pos + 1
                    if start > max_start:
                        break
            return out
        if self.required_byte >= 0:
            return self._match_all_required_byte(text)
        if self.use_dfa:
            return self.dfa.match_all(text)
        return self.nfa_matcher.match_all(text)

    def _match_all_required_byte(self, text: bytes) -> List[Span]:
        """matcher.mojo:864-898."""
        out: List[Span] = []
        tl = len(text)
        lookup = self.dfa.matcher.lookup
        pos = 0
        while pos < tl:
            hit = text.find(bytes([self.required_byte]), pos)
            if hit == -1:
                break
            start = hit
            while start > 0 and lookup[text[start - 1]] != 0:
                start -= 1
            m = self.dfa.match_first(text, start)
            if m is not None and m[1] > hit:
                out.append(m)
                pos = m[1]
                if pos <= hit:
                    pos = hit + 1
            else:
                pos = hit + 1
        return out


# ---------------------------------------------------------------------------
# sub() machinery (matcher.mojo:1418-1917)
# ---------------------------------------------------------------------------
def _parse_repl_template(repl: bytes):
    """matcher.mojo:1436-1469: list of (group_ref, start, length)."""
    segs = []
    i = 0
    lit_start = 0
    n = len(repl)
    while i < n:
        if repl[i] == ord("\\") and i + 1 < n:
            nc = repl[i + 1]
            if ord("1") <= nc <= ord("9"):
                if i > lit_start:
                    segs.append((0, lit_start, i - lit_start))
                segs.append((nc - ord("0"), 0, 0))
                i += 2
                lit_start = i
                continue
        i += 1
    if lit_start < n:
        segs.append((0, lit_start, n - lit_start))
    return segs


def _has_group_refs(repl: bytes) -> bool:
    """matcher.mojo:1472-1482."""
    for i in range(len(repl) - 1):
        if repl[i] == ord("\\") and ord("1") <= repl[i + 1] <= ord("9"):
            return True
    return False


def detect_fixed_width_groups(p: bytes) -> Optional[List[int]]:
    """matcher.mojo:1485-1586."""
    plen = len(p)
    segs: List[int] = []
    i = 0
    lit = 0
    while i < plen:
        if p[i] == ord("("):
            if lit > 0:
                segs.append(-lit)
                lit = 0
            if i + 1 < plen and p[i + 1] == ord("?"):
                return None
            i += 1
            if i + 1 >= plen or p[i] != ord("\\") or p[i + 1] != ord("d"):
                return None
            i += 2
            if i < plen and p[i] == ord("{"):
                i += 1
                ns = i
                while i < plen and ord("0") <= p[i] <= ord("9"):
                    i += 1
                if i == ns or i >= plen or p[i] != ord("}"):
                    return None
                width = int(p[ns:i])
                i += 1
                segs.append(width)
            elif i < plen and p[i] == ord(")"):
                segs.append(1)
            else:
                return None
            if i >= plen or p[i] != ord(")"):
                return None
            i += 1
        elif p[i] in (ord("|"), ord("[")):
            return None
        else:
            if p[i] == ord("\\") and i + 1 < plen:
                lit += 1
                i += 2
            else:
                lit += 1
                i += 1
    if not any(s > 0 for s in segs):
        return None
    return segs


class CompiledRegex:
    """matcher.mojo:929-1163."""

    def __init__(self, pattern: bytes, force_nfa: bool = False):
        if isinstance(pattern, str):
            pattern = pattern.encode("latin-1")
        self.pattern = pattern
        self.matcher = HybridMatcher(pattern, force_nfa)
        self._depth = 0
        self.fixed_total_width = -1
        self.fixed_num_groups = 0
        self.fixed_offsets = [0] * 10
        self.fixed_widths = [0] * 10
        self.fixed_concat = False
        self._try_precompute_fixed_sub()

    def _try_precompute_fixed_sub(self):
        """matcher.mojo:1002-1035."""
        segs = detect_fixed_width_groups(self.pattern)
        if not segs:
            return
        ng = 0
        total = 0
        has_lit = False
        for s in segs:
            if s > 0:
                ng += 1
                if ng > 9:
                    return
                self.fixed_offsets[ng] = total
                self.fixed_widths[ng] = s
                total += s
            else:
                has_lit = True
                total += -s
        if ng == 0:
            return
        self.fixed_num_groups = ng
        self.fixed_total_width = total
        self.fixed_concat = not has_lit

    def _enter(self):
        """Start of a public call: the LazyDFA of a '$' program forgets what earlier calls cached (module docstring)."""
        if FRESH_LAZY_CACHE and self._depth == 0 and self.matcher.nfa_matcher is not None:
            self.matcher.nfa_matcher.fresh_cache()

    def match_first(self, text: bytes, start: int = 0):
        return self.matcher.match_first(text, start)

    def match_next(self, text: bytes, start: int = 0):
        self._enter()
        return self.matcher.match_next(text, start)

    def match_all(self, text: bytes):
        self._enter()
        return self.matcher.match_all(text)

    def test(self, text: bytes) -> bool:
        self._enter()
        return self.matcher.match_next(text, 0) is not None

    def is_match(self, text: bytes, start: int = 0) -> bool:
        return self.matcher.is_match(text, start)

    def get_stats(self) -> str:
        return "Pattern: '%s', Engine: %s, Complexity: %s" % (
            self.pattern.decode("latin-1"), self.matcher.get_engine_type(),
            COMPLEXITY_NAMES[self.matcher.complexity])

    def captures_fixed(self, text: bytes, start: int = 0):
        """Group spans for the fixed-width form, in the order
        NFAEngine._match_group appends them (nfa.mojo:1057-1103):
        groups 1..g first, whole match (group 0) last."""
        if self.fixed_total_width < 0:
            raise UnsupportedByOracle("general capture groups need the backtracking NFA")
        m = self.match_next(text, start)
        if m is None:
            return None
        s = m[0]
        out = [(g, s + self.fixed_offsets[g], s + self.fixed_offsets[g] + self.fixed_widths[g])
               for g in range(1, self.fixed_num_groups + 1)]
        out.append((0, m[0], m[1]))
        return out

    def sub(self, repl: bytes, text: bytes, count: int = 0) -> bytes:
        self._enter()
        self._depth += 1   # (the match_next calls of this sub share one cache)
        try:
            return _sub_impl(self, repl, text, count)
        finally:
            self._depth -= 1


def _apply_template_fixed(template, repl, text, match_start, offs, widths, ng) -> bytes:
    """matcher.mojo:1592-1621.

    Upstream copies ``text_ptr[gs : gs + width]`` unchecked; the widths come from the pattern text with
    quantifiers ignored (matcher.mojo:1002-1035), so near the end of a text the window can reach behind it
    ('x(\\d)?' matching "x" at the very end) and upstream reads whatever follows the string.  There is no
    defined result to restate for those bytes: the slice below ends with the text, and that is what the
    product is held to (tests/test_gpu_parity.py::test_fixed_width_group_windows_end_with_the_text)."""
    out = b""
    for (gref, s, ln) in template:
        if gref > 0 and gref <= ng:
            gs = match_start + offs[gref]
            out += text[gs:gs + widths[gref]]
        else:
            out += repl[s:s + ln]
    return out


def _sub_impl(compiled: CompiledRegex, repl: bytes, text: bytes, count: int = 0) -> bytes:
    """matcher.mojo:1664-1854."""
    use_groups = _has_group_refs(repl)
    template = _parse_repl_template(repl) if use_groups else []
    tl = len(text)
    if tl == 0:
        return text
    result = b""
    pos = 0
    reps = 0
    if use_groups:
        if compiled.fixed_total_width >= 0:
            offs, widths = compiled.fixed_offsets, compiled.fixed_widths
            ng, total = compiled.fixed_num_groups, compiled.fixed_total_width
            if compiled.fixed_concat and tl == total:
                if all(ord("0") <= b <= ord("9") for b in text[:total]):
                    return _apply_template_fixed(template, repl, text, 0, offs, widths, ng)
                return text
            while pos <= tl:
                m = compiled.match_next(text, pos)
                if m is None:
                    break
                ms, me = m
                if (me + 1 if me == ms else me) <= pos:
                    raise ReferenceDoesNotTerminate(text)
                if ms > pos:
                    result += text[pos:ms]
                result += _apply_template_fixed(template, repl, text, ms, offs, widths, ng)
                reps += 1
                if me == ms:
                    if pos < tl:
                        result += text[pos:pos + 1]
                    pos = me + 1
                else:
                    pos = me
                if count > 0 and reps >= count:
                    break
        else:
            # general group path, matcher.mojo:1781-1822
            if compiled.matcher.nfa_matcher is None:
                raise UnsupportedByOracle("'.*' has no NFA matcher to ask for groups")
            bt = compiled.matcher.nfa_matcher.backtrack
            while pos <= tl:
                m, groups = bt.match_next_with_groups(text, pos)
                if m is None:
                    break
                ms, me = m
                if (me + 1 if me == ms else me) <= pos:
                    raise ReferenceDoesNotTerminate(text)
                if ms > pos:
                    result += text[pos:ms]
                group_idx = [-1] * 10
                for gi, (gid, _gs, _ge) in enumerate(groups):
                    if 1 <= gid <= 9:
                        group_idx[gid] = gi          # a later entry of the same group wins
                for (gref, s_, ln) in template:      # _apply_template_groups, matcher.mojo:1624-1646
                    if gref > 0:
                        idx = group_idx[gref] if gref <= 9 else -1
                        if idx >= 0:
                            # Match.get_match_text (matching.mojo:39-46) reads [start, end) unchecked; a span that
                            # ends behind the text is cut at its end here, as in _apply_template_fixed
                            result += text[groups[idx][1]:groups[idx][2]]
                    else:
                        result += repl[s_:s_ + ln]
                reps += 1
                if me == ms:
                    if pos < tl:
                        result += text[pos:pos + 1]
                    pos = me + 1
                else:
                    pos = me
                if count > 0 and reps >= count:
                    break
    else:
        while pos <= tl:
            m = compiled.match_next(text, pos)
            if m is None:
                break
            ms, me = m
            if (me + 1 if me == ms else me) <= pos:
                raise ReferenceDoesNotTerminate(text)
            if ms > pos:
                result += text[pos:ms]
            result += repl
            reps += 1
            if me == ms:
                if pos < tl:
                    result += text[pos:pos + 1]
                pos = me + 1
            else:
                pos = me
            if count > 0 and reps >= count:
                break
    if pos < tl:
        result += text[pos:]
    return result


# ---------------------------------------------------------------------------
# module-level API (matcher.mojo:1292-1415, 1857-1917)
# ---------------------------------------------------------------------------
_CACHE = {}


def _b(x) -> bytes:
    return x.encode("latin-1") if isinstance(x, str) else bytes(x)


def compile_regex(pattern) -> CompiledRegex:
    pattern = _b(pattern)
    c = _CACHE.get(pattern)
    if c is None:
        c = CompiledRegex(pattern)
        _CACHE[pattern] = c
    return c


def clear_regex_cache():
    _CACHE.clear()


def search(pattern, text) -> Optional[Span]:
    return compile_regex(pattern).match_next(_b(text))


def findall(pattern, text) -> List[Span]:
    return compile_regex(pattern).match_all(_b(text))


def match_first(pattern, text) -> Optional[Span]:
    """matcher.mojo:1396-1415: keep the result only if it starts at 0."""
    r = compile_regex(pattern).match_first(_b(text), 0)
    if r is not None and r[0] == 0:
        return r
    return None


def split(pattern, text, maxsplit: int = 0) -> List[bytes]:
    """matcher.mojo:1357-1393."""
    text = _b(text)
    out = []
    prev = 0
    done = 0
    for (s, e) in findall(pattern, text):
        if maxsplit != 0 and done >= maxsplit:
            break
        out.append(text[prev:s])
        prev = e
        done += 1
    out.append(text[prev:])
    return out


def sub(pattern, repl, text, count: int = 0) -> bytes:
    return compile_regex(pattern).sub(_b(repl), _b(text), count)
