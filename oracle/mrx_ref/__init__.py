"""ORACLE -- test infrastructure only, never the product path.

CPU restatement of the msaelices/mojo-regex matching hot path (reference
v0.21.0, Mojo 1.0.0), written by following the reference source function by
function.  Every function cites the reference file:line it restates (paths are
relative to the reference checkout, e.g. ``src/regex/dfa.mojo:1906``).

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product (``mojo_regex_amd``) never
imports, links or executes anything under ``oracle/``.

Parity status: PINNED against the reference's own known-answer tests
(``tests/golden/*.json``: matching results, routing and ``sub`` from
``tests/test_{matcher,dfa,comptime_regex,predefined_classes,split,simd}.mojo``, OnePass,
classifier / literal helpers, class matcher, lexer tokens, parser AST shapes and required
syntax errors from the remaining test files -- transcribed as data with file:line per
vector by the two scripts under ``tests/golden/`` or by hand where noted).  The reference itself cannot
be executed here (no Mojo toolchain in the image), see DESIGN.md.  One routing
case is derived from source only and is labelled PARITY-UNPINNED: quantified
literal alternation such as ``(x|y|foo|bar)+``.

SIMD width: the reference picks ``SIMD_WIDTH = simd_width_of[uint8]()`` at
compile time (``src/regex/simd_ops.mojo:58``).  Results depend on it only via
nibble-table false positives (``simd_ops.mojo:63-134``); the oracle models
``SIMD_WIDTH = 32`` (AVX2 x86-64) and exposes it as ``SIMD_WIDTH``.
"""

SIMD_WIDTH = 32

from .frontend import parse, scan, RegexSyntaxError  # noqa: E402,F401
from .hybrid import (  # noqa: E402,F401
    CompiledRegex,
    compile_regex,
    match_first,
    search,
    findall,
    split,
    sub,
    UnsupportedByOracle,
    ReferenceDoesNotTerminate,
)
