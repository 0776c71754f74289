"""GPU parity on the reference's own benchmark list (benchmarks/bench_engine.mojo:578-1100, restated
as data in tests/bench_engine_cases.py): every case's operation on the case's text and on rotations
of it, bit-exact against the oracle; the cases the reference sends to its backtracking NFA must be
refused, not approximated."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import mojo_regex_amd as M  # noqa: E402
import bench_engine_cases as B  # noqa: E402
from mojo_regex_amd.api import UnsupportedPattern  # noqa: E402
from mrx_ref import hybrid as O  # noqa: E402  (oracle: checker only)
from mrx_ref import UnsupportedByOracle  # noqa: E402


def _oracle(case, text):
    """The case's operation on one text through the oracle."""
    rx = O.compile_regex(case.pattern)
    if case.op == "match_first":
        return rx.match_first(text)
    if case.op == "search":
        return rx.match_next(text, 0)
    if case.op == "findall":
        return rx.match_all(text)
    if case.op == "is_match":
        return rx.is_match(text)
    if case.op == "sub":
        return O.sub(case.pattern, case.repl, text, case.count)
    raise ValueError(case.op)


def _product(case, rows):
    rx = M.compile_regex(case.pattern)
    if case.op == "match_first":
        s, e = rx.match_first(rows)
        return [None if a < 0 else (int(a), int(b)) for a, b in zip(s, e)]
    if case.op == "search":
        s, e = rx.match_next(rows)
        return [None if a < 0 else (int(a), int(b)) for a, b in zip(s, e)]
    if case.op == "findall":
        return rx.findall_lists(rows)
    if case.op == "is_match":
        return [bool(x) for x in rx.is_match(rows)]
    if case.op == "sub":
        return rx.sub(case.repl, rows, case.count)
    raise ValueError(case.op)


@pytest.mark.parametrize("case", B.CASES, ids=[c.name for c in B.CASES])
def test_reference_benchmark_case(case):
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU: the HIP path has no fallback")
    rows = B.case_rows(case, max(2, min(64, 200000 // max(len(case.text), 1))))
    try:
        want = [_oracle(case, t) for t in rows]
    except UnsupportedByOracle:
        with pytest.raises(UnsupportedPattern):
            _product(case, rows)
        return
    got = _product(case, rows)
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, "%s row %d: product %r oracle %r" % (case.name, i, str(g)[:200], str(w)[:200])


def test_suite_lists_every_reference_benchmark():
    names = [c.name for c in B.CASES] + [a for c in B.CASES for a in c.aliases]
    assert len(names) == len(set(names)) == 89
