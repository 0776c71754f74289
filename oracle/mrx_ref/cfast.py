"""ORACLE (test infrastructure) -- ctypes bridge to oracle/c/mrx_oracle.c.

Runs the C restatement of the DFAEngine loops over whole batches, with tables
taken from the Python oracle.  Only plain DFAEngine routes are handled here
(which is what the BASELINE.json configs use); anything else stays in Python.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import SIMD_WIDTH
from .hybrid import CompiledRegex, UnsupportedByOracle

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.dirname(_HERE)
_SO = os.path.join(_ORACLE_DIR, "_build", "libmrx_oracle.so")
_lib = None


class _Dfa(C.Structure):
    _fields_ = [
        ("nstates", C.c_int32), ("trans", C.c_void_p), ("accepting", C.c_void_p),
        ("has_start_anchor", C.c_int32), ("has_end_anchor", C.c_int32),
        ("is_pure_literal", C.c_int32), ("has_simd_matcher", C.c_int32),
        ("simd_scan_eligible", C.c_int32), ("lookup", C.c_void_p), ("num_ranges", C.c_int32),
        ("lo_tbl", C.c_void_p), ("hi_tbl", C.c_void_p), ("literal", C.c_void_p),
        ("literal_len", C.c_int32), ("simd_width", C.c_int32), ("ranges", C.c_void_p),
    ]


_NATIVE_SO = None


def build_native() -> str:
    """The same C file built -O3 -march=native on THIS machine (bench.py's cpu_baseline leg, SURVEY.md
    8(d)); the checked-in Makefile target is -march=x86-64-v3 so that one binary runs on every box.
    Falls back to the portable build when there is no compiler here."""
    global _NATIVE_SO
    if _NATIVE_SO is None:
        import tempfile
        out = os.path.join(tempfile.gettempdir(), "libmrx_oracle_native_%d.so" % os.getuid())
        src = os.path.join(_ORACLE_DIR, "c", "mrx_oracle.c")
        try:
            subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-fopenmp", "-shared", "-o", out, src],
                           check=True, capture_output=True)
            _NATIVE_SO = out
        except Exception:
            _NATIVE_SO = ""
    return _NATIVE_SO


def load(native: bool = False):
    global _lib
    if native and build_native():
        return _bind(C.CDLL(build_native()))
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)
        _lib = _bind(C.CDLL(_SO))
    return _lib


def _bind(lib):
    P = C.POINTER(_Dfa)
    lib.mrx_oracle_findall_batch.restype = C.c_int64
    lib.mrx_oracle_findall_batch.argtypes = [P, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_int64]
    lib.mrx_oracle_count_batch_mt.restype = C.c_int64
    lib.mrx_oracle_count_batch_mt.argtypes = [P, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    lib.mrx_oracle_findall_at_mt.restype = C.c_int64
    lib.mrx_oracle_findall_at_mt.argtypes = [P, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.mrx_oracle_span_batch.restype = None
    lib.mrx_oracle_span_batch.argtypes = [P, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p]
    lib.mrx_oracle_match_first_bytes.restype = C.c_int64
    lib.mrx_oracle_match_first_bytes.argtypes = [P, C.c_void_p, C.c_void_p, C.c_int64]
    return lib


class CDfa:
    """A DFAEngine of the Python oracle, frozen into C-readable arrays."""

    def __init__(self, pattern: bytes, native: bool = False):
        c = CompiledRegex(pattern)
        m = c.matcher
        if (m.is_wildcard_match_any or not m.use_dfa or m.is_exact_literal
                or m.prefilter_literal is not None or m.required_byte >= 0):
            raise UnsupportedByOracle("C oracle covers plain DFAEngine routes only")
        e = m.dfa
        self.pattern = pattern
        self._trans = np.array([s.transitions for s in e.states], dtype=np.int32).reshape(-1, 256)
        self._acc = np.array([1 if s.is_accepting else 0 for s in e.states], dtype=np.uint8)
        self._lookup = np.array(e.matcher.lookup, dtype=np.uint8)
        self._lo = np.array(e.matcher.lo_tbl, dtype=np.uint8)
        self._hi = np.array(e.matcher.hi_tbl, dtype=np.uint8)
        self._lit = np.frombuffer(e.literal_pattern + b"\0", dtype=np.uint8).copy()
        d = _Dfa()
        d.nstates = len(e.states)
        d.trans = self._trans.ctypes.data
        d.accepting = self._acc.ctypes.data
        d.has_start_anchor = int(e.has_start_anchor)
        d.has_end_anchor = int(e.has_end_anchor)
        d.is_pure_literal = int(e.is_pure_literal)
        d.has_simd_matcher = int(e.has_simd_matcher)
        d.simd_scan_eligible = int(e.simd_scan_eligible)
        d.lookup = self._lookup.ctypes.data
        d.num_ranges = e.matcher.num_ranges
        d.lo_tbl = self._lo.ctypes.data
        d.hi_tbl = self._hi.ctypes.data
        d.literal = self._lit.ctypes.data
        d.literal_len = len(e.literal_pattern)
        d.simd_width = SIMD_WIDTH
        # contiguous ranges of the first class (CharacterClassSIMD._detect_ranges, simd_ops.mojo:364-401)
        self._ranges = None
        if 1 <= e.matcher.num_ranges <= 3:
            rs, inside = [], False
            for c_ in range(257):
                m_ = c_ < 256 and self._lookup[c_] != 0
                if m_ and not inside:
                    rs.append(c_)
                elif not m_ and inside:
                    rs.append(c_ - 1)
                inside = m_
            if len(rs) == 2 * e.matcher.num_ranges:
                self._ranges = np.array(rs, dtype=np.uint8)
        d.ranges = self._ranges.ctypes.data if self._ranges is not None else None
        self._d = d
        self._lib = load(native)

    def findall_batch(self, data: np.ndarray, offsets: np.ndarray, want_spans: bool = True):
        """(counts int32[n], spans int32[total,2] or None, total)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        counts = np.zeros(n, np.int32)
        total = self._lib.mrx_oracle_findall_batch(C.byref(self._d), data.ctypes.data,
                                                   offsets.ctypes.data, n, counts.ctypes.data, None, 0)
        if not want_spans:
            return counts, None, int(total)
        spans = np.empty((max(int(total), 1), 2), np.int32)
        self._lib.mrx_oracle_findall_batch(C.byref(self._d), data.ctypes.data, offsets.ctypes.data, n,
                                           counts.ctypes.data, spans.ctypes.data, int(total))
        return counts, spans[: int(total)], int(total)

    def count_batch_mt(self, data: np.ndarray, offsets: np.ndarray, threads: int):
        """findall counts with the texts split over `threads` host threads: (counts, total)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        counts = np.zeros(n, np.int32)
        total = self._lib.mrx_oracle_count_batch_mt(C.byref(self._d), data.ctypes.data, offsets.ctypes.data,
                                                    n, counts.ctypes.data, int(threads))
        return counts, int(total)

    def findall_at_mt(self, data: np.ndarray, offsets: np.ndarray, prefix: np.ndarray, threads: int):
        """findall of every text on `threads` host threads, text i's spans written where `prefix` (int64[n + 1],
        e.g. the device's CSR offsets; only differences and prefix[0] matter) puts them: (counts int32[n] -- the
        true counts, whatever room prefix left --, spans int32[prefix[n] - prefix[0], 2], total)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        prefix = np.ascontiguousarray(prefix, dtype=np.int64)
        n = len(offsets) - 1
        assert len(prefix) == n + 1
        counts = np.zeros(n, np.int32)
        spans = np.full((max(int(prefix[-1] - prefix[0]), 1), 2), -7, np.int32)
        total = self._lib.mrx_oracle_findall_at_mt(C.byref(self._d), data.ctypes.data, offsets.ctypes.data, n,
                                                   prefix.ctypes.data, spans.ctypes.data, counts.ctypes.data, int(threads))
        return counts, spans[: int(prefix[-1] - prefix[0])], int(total)

    def span_batch(self, which: str, data: np.ndarray, offsets: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        s = np.empty(n, np.int32)
        e = np.empty(n, np.int32)
        self._lib.mrx_oracle_span_batch(C.byref(self._d), 0 if which == "match_first" else 1,
                                        data.ctypes.data, offsets.ctypes.data, n, s.ctypes.data,
                                        e.ctypes.data)
        return s, e

    def match_first_bytes(self, data: np.ndarray, offsets: np.ndarray) -> int:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        return int(self._lib.mrx_oracle_match_first_bytes(C.byref(self._d), data.ctypes.data,
                                                          offsets.ctypes.data, len(offsets) - 1))
