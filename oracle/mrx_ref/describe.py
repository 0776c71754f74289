"""ORACLE (test infrastructure) -- text dump of what the oracle derives from a
pattern, in the same line format as the product's mrx_describe() so that
tests/test_host_tables.py can diff the two independent implementations."""
from __future__ import annotations

from .analysis import COMPLEXITY_NAMES
from .hybrid import CompiledRegex


def describe(pattern: bytes) -> str:
    c = CompiledRegex(pattern)
    m = c.matcher
    o = []
    o.append("pattern=" + pattern.decode("latin-1"))
    o.append("engine_type=" + m.get_engine_type())
    o.append("complexity=" + COMPLEXITY_NAMES[m.complexity])
    o.append("use_dfa=%d wildcard_any=%d use_pure_dfa=%d" % (m.use_dfa, m.is_wildcard_match_any,
                                                           m.use_pure_dfa))
    o.append("exact_literal=%d literal_has_anchors=%d prefilter=%d required_byte=%d" % (
        m.is_exact_literal, m.literal_has_anchors, m.prefilter_literal is not None,
        m.required_byte))
    o.append("best_literal=" + (m.best_literal or b"").hex())
    if m.use_dfa:
        e = m.dfa
        o.append("dfa.shape=" + e.shape)
        o.append("dfa.nstates=%d" % len(e.states))
        o.append("dfa.flags start_anchor=%d end_anchor=%d pure_literal=%d has_matcher=%d "
                 "scan_eligible=%d" % (e.has_start_anchor, e.has_end_anchor, e.is_pure_literal,
                                       e.has_simd_matcher, e.simd_scan_eligible))
        o.append("dfa.literal=" + e.literal_pattern.hex())
        o.append("dfa.accepting=" + "".join("1" if s.is_accepting else "0" for s in e.states))
        if e.has_simd_matcher:
            o.append("dfa.matcher.num_ranges=%d" % e.matcher.num_ranges)
            o.append("dfa.matcher.lookup=" + "".join(str(x) for x in e.matcher.lookup))
        for si, s in enumerate(e.states):
            parts = []
            c0 = 0
            while c0 < 256:
                t = s.transitions[c0]
                c1 = c0
                while c1 + 1 < 256 and s.transitions[c1 + 1] == t:
                    c1 += 1
                if t != -1:
                    parts.append("%d-%d:%d" % (c0, c1, t))
                c0 = c1 + 1
            o.append("dfa.row%d=%s" % (si, ",".join(parts)))
    elif not m.is_wildcard_match_any:
        prog = m.nfa_matcher.program
        o.append("nfa.program_len=%d" % len(prog))
        o.append("nfa.program=" + "".join("%d:%d:%d;" % tuple(i) for i in prog.instructions))
        e = m.nfa_matcher.engine
        o.append("nfa.literal_opt=%d starts_dotstar=%d ends_dotstar=%d" % (
            e.has_literal_optimization, e.starts_with_dotstar, e.ends_with_dotstar))
    o.append("fixed_groups=%d fixed_total=%d fixed_concat=%d" % (
        c.fixed_num_groups, c.fixed_total_width, c.fixed_concat))
    return "\n".join(o) + "\n"
