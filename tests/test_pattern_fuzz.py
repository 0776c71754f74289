"""Generated patterns: the product's C++ compiler and the oracle's Python restatement must agree on
everything mrx_describe() prints (syntax errors, routing, DFA tables, matcher sets, PikeVM program)
-- CPU only; the GPU half (results on random texts) is in test_gpu_parity.py."""
import mojo_regex_amd as M
from mrx_ref import RegexSyntaxError as OracleSyntaxError
from mrx_ref.describe import describe
from pattern_gen import patterns, patterns2

N_PATTERNS = 3000


def _product(p: bytes) -> str:
    try:
        d = M.CompiledRegex(p).describe()
    except M.RegexSyntaxError as e:
        return "SYNTAX:" + str(e)
    return "\n".join(l for l in d.strip().split("\n")
                     if not l.startswith(("support.", "device.", "nfa.has_filter")))


def _oracle(p: bytes) -> str:
    try:
        return describe(p).strip()
    except OracleSyntaxError as e:
        return "SYNTAX:" + str(e)


def test_generated_patterns_compile_to_the_same_tables():
    bad = []
    kinds = {"SYNTAX": 0, "DFA": 0, "NFA": 0}
    for p in patterns(20260501, N_PATTERNS):
        pb = p.encode()
        a, b = _oracle(pb), _product(pb)
        if a != b:
            bad.append(p)
        if a.startswith("SYNTAX"):
            kinds["SYNTAX"] += 1
        elif "engine_type=DFA" in a:
            kinds["DFA"] += 1
        else:
            kinds["NFA"] += 1
    assert not bad, bad[:10]
    # the generator must actually exercise all three outcomes
    assert kinds["DFA"] > 300 and kinds["NFA"] > 300 and kinds["SYNTAX"] > 20, kinds


def test_second_generator_compiles_to_the_same_tables():
    """tests/pattern_gen.py's second generator ('.*' with literals, word alternations, escaped specials, \\D \\W \\S,
    larger counts, anchors inside alternations): the shapes the first one rarely reaches."""
    bad = [p for p in patterns2(20260502, 1500) if _oracle(p.encode()) != _product(p.encode())]
    assert not bad, bad[:10]
