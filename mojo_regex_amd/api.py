"""ctypes binding of include/mrx.h plus the reference-shaped Python interface.

See the package docstring for the mapping to src/regex/matcher.mojo.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libmrx_hip.so"
_lib = None

MRX_OK, MRX_E_SYNTAX, MRX_E_UNSUPPORTED, MRX_E_NO_DEVICE, MRX_E_CAPACITY, MRX_E_ARGUMENT = range(6)


class MrxError(RuntimeError):
    """Any failure reported by libmrx_hip.so."""


class RegexSyntaxError(MrxError):
    """The reference's lexer/parser raises on this pattern (same message)."""


class UnsupportedPattern(MrxError):
    """The reference routes this pattern/operation to an engine outside the hot
    path this library implements (backtracking NFA, OnePass)."""


def library_path() -> str:
    # MRX_LIB: measurement hook (tools/ablate.sh loads instrumented builds of the same library)
    return os.environ.get("MRX_LIB") or os.path.join(_HERE, _LIB_NAME)


def load_library():
    """Load libmrx_hip.so (built in-tree by __graft_entry__.build()).  Loud
    failure if it is missing: there is no other implementation to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise MrxError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    try:
        # PyTorch wheels bundle their own HIP runtime.  Load it first so that the
        # process holds ONE libamdhip64 (ours resolves to the copy already mapped);
        # two runtimes in one process do not share the device.
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    H = C.c_void_p
    u8p, i32p, i64p = C.c_void_p, C.c_void_p, C.c_void_p
    sigs = {
        "mrx_compile": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(H)]),
        "mrx_compile_ex": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(H)]),
        "mrx_free": (None, [H]),
        "mrx_last_error": (C.c_char_p, []),
        "mrx_engine_type": (C.c_char_p, [H]),
        "mrx_stats": (C.c_char_p, [H]),
        "mrx_describe": (C.c_size_t, [H, C.c_char_p, C.c_size_t]),
        "mrx_num_groups": (C.c_int, [H]),
        "mrx_match_first_dev": (C.c_int, [H, u8p, i64p, C.c_int64, i32p, i32p, C.c_void_p]),
        "mrx_search_dev": (C.c_int, [H, u8p, i64p, C.c_int64, i32p, i32p, C.c_void_p]),
        "mrx_match_first_strided_dev": (C.c_int, [H, u8p, C.c_int64, i32p, C.c_int32, C.c_int64, i32p,
                                                  i32p, C.c_void_p]),
        "mrx_search_strided_dev": (C.c_int, [H, u8p, C.c_int64, i32p, C.c_int32, C.c_int64, i32p, i32p,
                                             C.c_void_p]),
        "mrx_is_match_dev": (C.c_int, [H, u8p, i64p, C.c_int64, u8p, C.c_void_p]),
        "mrx_is_match_strided_dev": (C.c_int, [H, u8p, C.c_int64, i32p, C.c_int32, C.c_int64, u8p, C.c_void_p]),
        "mrx_match_first_at_dev": (C.c_int, [H, u8p, i64p, C.c_int64, C.c_int32, i32p, i32p, i32p, C.c_void_p]),
        "mrx_search_at_dev": (C.c_int, [H, u8p, i64p, C.c_int64, C.c_int32, i32p, i32p, i32p, C.c_void_p]),
        "mrx_is_match_at_dev": (C.c_int, [H, u8p, i64p, C.c_int64, C.c_int32, i32p, u8p, C.c_void_p]),
        "mrx_match_first_at_strided_dev": (C.c_int, [H, u8p, C.c_int64, i32p, C.c_int32, C.c_int64, C.c_int32, i32p,
                                                     i32p, i32p, C.c_void_p]),
        "mrx_search_at_strided_dev": (C.c_int, [H, u8p, C.c_int64, i32p, C.c_int32, C.c_int64, C.c_int32, i32p,
                                                i32p, i32p, C.c_void_p]),
        "mrx_is_match_at_strided_dev": (C.c_int, [H, u8p, C.c_int64, i32p, C.c_int32, C.c_int64, C.c_int32, i32p,
                                                  u8p, C.c_void_p]),
        "mrx_findall_dev": (C.c_int, [H, u8p, i64p, C.c_int64, i64p, i32p, C.c_int64,
                                      C.POINTER(C.c_int64), C.c_void_p]),
        "mrx_findall_known_dev": (C.c_int, [H, u8p, i64p, C.c_int64, C.c_int64, C.c_int64, i64p, i32p, C.c_int64,
                                           C.POINTER(C.c_int64), C.c_void_p]),
        "mrx_findall_strided_dev": (C.c_int, [H, u8p, C.c_int64, i32p, C.c_int32, C.c_int64, i64p,
                                              i32p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
        "mrx_count_dev": (C.c_int, [H, u8p, i64p, C.c_int64, i32p, C.c_void_p]),
        "mrx_count_strided_dev": (C.c_int, [H, u8p, C.c_int64, i32p, C.c_int32, C.c_int64, i32p, C.c_void_p]),
        "mrx_captures_strided_dev": (C.c_int, [H, u8p, C.c_int64, i32p, C.c_int32, C.c_int64, i32p, C.c_void_p]),
        "mrx_captures_dev": (C.c_int, [H, u8p, i64p, C.c_int64, i32p, C.c_void_p]),
        "mrx_sub_dev": (C.c_int, [H, C.c_char_p, C.c_size_t, C.c_int64, u8p, i64p, C.c_int64, i64p,
                                  u8p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
        "mrx_sub_known_dev": (C.c_int, [H, C.c_char_p, C.c_size_t, C.c_int64, u8p, i64p, C.c_int64, C.c_int64, C.c_int64,
                                        i64p, u8p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
        "mrx_sub_strided_dev": (C.c_int, [H, C.c_char_p, C.c_size_t, C.c_int64, u8p, C.c_int64, i32p, C.c_int32,
                                          C.c_int64, i64p, u8p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
        "mrx_match_first_batch": (C.c_int, [H, u8p, i64p, C.c_int64, i32p, i32p]),
        "mrx_search_batch": (C.c_int, [H, u8p, i64p, C.c_int64, i32p, i32p]),
        "mrx_is_match_batch": (C.c_int, [H, u8p, i64p, C.c_int64, u8p]),
        "mrx_split_batch": (C.c_int, [H, u8p, i64p, C.c_int64, C.c_int64, i64p, i32p, C.c_int64, C.POINTER(C.c_int64)]),
        "mrx_split_dev": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.POINTER(C.c_int64), C.c_void_p]),
        "mrx_split_strided_dev": (C.c_int, [H, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p,
                                            C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
        "mrx_findall_batch": (C.c_int, [H, u8p, i64p, C.c_int64, i64p, i32p, C.c_int64,
                                        C.POINTER(C.c_int64)]),
        "mrx_captures_batch": (C.c_int, [H, u8p, i64p, C.c_int64, i32p]),
        "mrx_sub_batch": (C.c_int, [H, C.c_char_p, C.c_size_t, C.c_int64, u8p, i64p, C.c_int64, i64p,
                                    u8p, C.c_int64, C.POINTER(C.c_int64)]),
        "mrx_timing_reset": (None, []),
        "mrx_timing_enable": (None, [C.c_int]),
        "mrx_timing_scan_ms": (C.c_double, [C.POINTER(C.c_int64)]),
        "mrx_last_kernel_name": (C.c_char_p, []),
        "mrx_debug_force_generic": (None, [C.c_int]),
        "mrx_debug_long_text_kernels": (None, [C.c_int]),
        "mrx_debug_fused_findall": (None, [C.c_int]),
        "mrx_debug_stream_bits": (None, [C.c_int]),
        "mrx_debug_stream_bits_trace": (None, [C.c_void_p]),
        "mrx_debug_dynamic_texts": (None, [C.c_int]),
        "mrx_debug_subs_group": (None, [C.c_int]),
        "mrx_debug_split_findall": (None, [C.c_int]),
        "mrx_debug_dense_rows": (None, [C.c_int]),
        "mrx_debug_tries_always": (None, [C.c_int]),
        "mrx_debug_chain_sub_general": (None, [C.c_int]),
        "mrx_testing_emptywalk_findall": (C.c_int, [H, C.c_char_p, C.c_int, i32p, C.c_int]),
        "mrx_debug_litscan_pieces": (None, [C.c_int]),
        "mrx_debug_multiwalk": (None, [C.c_int]),
        "mrx_debug_rec_skew": (None, [C.c_int64]),
        "mrx_release_scratch": (None, []),
        "mrx_debug_scratch_bytes": (C.c_size_t, []),
        "mrx_version": (C.c_char_p, []),
        # include/mrx_comm.h: results exchange between ranks (RCCL, opened on first use)
        "mrx_comm_unique_id": (C.c_int, [C.c_void_p]),
        "mrx_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(H)]),
        "mrx_comm_free": (None, [H]),
        "mrx_comm_rank": (C.c_int, [H]),
        "mrx_comm_size": (C.c_int, [H]),
        "mrx_comm_spans_staging_bytes": (C.c_size_t, [H, C.c_int64, C.c_int64]),
        "mrx_comm_reserve": (C.c_int, [H, C.c_size_t]),
        "mrx_allgather_fixed": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
        "mrx_allgatherv_rows": (C.c_int, [H, C.c_void_p, C.c_int64, C.c_size_t, C.c_void_p, C.c_int64,
                                          C.POINTER(C.c_int64), C.c_void_p]),
        "mrx_allgatherv_spans": (C.c_int, [H, i64p, C.c_int64, i32p, C.c_int64, C.c_int64, i64p, C.c_int64, i32p,
                                           C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), i32p, C.c_void_p]),
        "mrx_testing_comm_shift": (C.c_int, [i64p, C.c_int64, i64p, C.c_int, C.c_int, i64p, C.c_int64, C.c_void_p]),
        "mrx_testing_comm_compact": (C.c_int, [i64p, C.c_int, C.c_int, i64p, C.c_int64, i32p, C.c_int64, i64p, C.c_int64,
                                               i32p, C.c_int64, i32p, C.c_void_p]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


# include/mrx.h (the drop-in boundary) and include/mrx_testing.h (measurement / testing hooks)
EXPORTED_SYMBOLS = [
    "mrx_compile", "mrx_compile_ex", "mrx_free", "mrx_last_error", "mrx_engine_type", "mrx_stats", "mrx_describe",
    "mrx_num_groups", "mrx_match_first_dev", "mrx_search_dev", "mrx_match_first_strided_dev",
    "mrx_search_strided_dev", "mrx_is_match_dev", "mrx_is_match_strided_dev",
    "mrx_findall_dev", "mrx_findall_known_dev", "mrx_findall_strided_dev", "mrx_count_dev", "mrx_count_strided_dev",
    "mrx_captures_strided_dev", "mrx_captures_dev",
    "mrx_match_first_at_dev", "mrx_search_at_dev", "mrx_is_match_at_dev", "mrx_match_first_at_strided_dev",
    "mrx_search_at_strided_dev", "mrx_is_match_at_strided_dev",
    "mrx_sub_dev", "mrx_sub_known_dev", "mrx_sub_strided_dev", "mrx_split_dev", "mrx_split_strided_dev", "mrx_split_batch", "mrx_match_first_batch", "mrx_search_batch", "mrx_is_match_batch",
    "mrx_findall_batch", "mrx_captures_batch", "mrx_sub_batch", "mrx_version", "mrx_release_scratch",
]
TESTING_SYMBOLS = [
    "mrx_timing_reset", "mrx_timing_enable", "mrx_timing_scan_ms", "mrx_last_kernel_name",
    "mrx_debug_force_generic", "mrx_debug_long_text_kernels", "mrx_debug_scratch_bytes",
    "mrx_debug_fused_findall", "mrx_debug_stream_bits", "mrx_debug_stream_bits_trace", "mrx_debug_dynamic_texts", "mrx_debug_subs_group",
    "mrx_debug_split_findall", "mrx_debug_dense_rows", "mrx_debug_tries_always", "mrx_debug_chain_sub_general", "mrx_testing_emptywalk_findall", "mrx_debug_litscan_pieces", "mrx_debug_multiwalk", "mrx_debug_rec_skew", "mrx_testing_comm_shift", "mrx_testing_comm_compact",
]
COMM_SYMBOLS = [
    "mrx_comm_unique_id", "mrx_comm_init", "mrx_comm_free", "mrx_comm_rank", "mrx_comm_size",
    "mrx_comm_spans_staging_bytes", "mrx_comm_reserve",
    "mrx_allgather_fixed", "mrx_allgatherv_rows", "mrx_allgatherv_spans",
]


def _b(x) -> bytes:
    if isinstance(x, str):
        return x.encode("utf-8")
    return bytes(x)


def _check(rc: int):
    if rc == MRX_OK:
        return
    msg = load_library().mrx_last_error().decode("utf-8", "replace")
    if rc == MRX_E_SYNTAX:
        raise RegexSyntaxError(msg)
    if rc == MRX_E_UNSUPPORTED:
        raise UnsupportedPattern(msg)
    raise MrxError("mrx error %d: %s" % (rc, msg))


def pack_texts(texts: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """Pack texts back to back: (data uint8[total], offsets int64[n+1])."""
    bs = [_b(t) for t in texts]
    offsets = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        np.cumsum([len(x) for x in bs], out=offsets[1:])
    data = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if offsets[-1] else np.zeros(0, np.uint8)
    return data, offsets


class DeviceBatch:
    """Texts already resident in HBM (torch tensors on a cuda device).

    CSR form:      DeviceBatch(data_u8, offsets_i64)
    strided form:  DeviceBatch.strided(data_u8[n*stride], stride, length or lens_i32)
    """

    def __init__(self, data, offsets=None, *, stride: int = 0, length: int = 0, lens=None, n=None):
        import torch
        if data.dtype != torch.uint8 or not data.is_contiguous():
            raise MrxError("batch data must be a contiguous uint8 tensor")
        self.data, self.offsets, self.stride, self.length, self.lens = data, offsets, stride, length, lens
        # CSR batches whose offsets were built on the host: offsets[n] and the longest text, so that findall does not
        # have to read them back from the device (mrx_findall_known_dev); None = not known.  Private: set by
        # from_texts / from_arrow only, from the very offsets they upload -- the C side sizes its record scratch from
        # them ("neither may be too small"), so a stale or hand-set value would be an out-of-bounds write, not an error
        self._end_offset: Optional[int] = None
        self._max_len: Optional[int] = None
        if offsets is not None:
            if offsets.dtype != torch.int64 or not offsets.is_contiguous() or offsets.device != data.device:
                raise MrxError("offsets must be a contiguous int64 tensor on the data's device")
            if offsets.numel() < 1:
                raise MrxError("offsets needs n + 1 entries")
            self.n = int(offsets.numel()) - 1
        else:
            self.n = int(n)

    @classmethod
    def csr_known(cls, data, offsets, end_offset: int, max_len: int):
        """A CSR batch whose builder knows offsets[n] and the longest text (it produced the offsets, or holds an Arrow
        array's): findall and sub then need no look at the device before their first kernel (mrx_findall_known_dev,
        mrx_sub_known_dev).  Upper bounds are fine; neither may be too small -- the C side sizes scratch from them that
        its kernels index with the real offsets."""
        b = cls(data, offsets)
        if int(end_offset) < 0 or int(max_len) < 0:
            raise MrxError("end_offset and max_len must not be negative")
        b._end_offset, b._max_len = int(end_offset), int(max_len)
        return b

    @classmethod
    def strided(cls, data, stride: int, length: Optional[int] = None, lens=None):
        """Texts at a fixed pitch: text i = data[i*stride : i*stride + (lens[i] | length)].  `length`
        (common to all texts) or `lens` (int32[n] on the data's device; takes precedence) must be given;
        lens[i] <= stride is the caller's contract (the kernels read lens[i] bytes of row i)."""
        import torch
        stride = int(stride)
        if stride <= 0 or data.numel() % stride:
            raise MrxError("data size %d is not a multiple of the stride %d" % (data.numel(), stride))
        n = data.numel() // stride
        if length is None and lens is None:
            raise MrxError("give length= (common to all texts) or lens= (per text; it takes precedence)")
        if lens is not None:
            if lens.dtype != torch.int32 or not lens.is_contiguous() or lens.device != data.device or lens.numel() != n:
                raise MrxError("lens must be a contiguous int32[n] tensor on the data's device")
            length = 0
        elif not 0 <= int(length) <= stride:
            raise MrxError("length must be in [0, stride]")
        return cls(data, None, stride=stride, length=int(length), lens=lens, n=n)

    @classmethod
    def from_texts(cls, texts: Sequence, device="cuda"):
        import torch
        data, offsets = pack_texts(texts)
        d = torch.from_numpy(data).to(device) if data.size else torch.zeros(0, dtype=torch.uint8, device=device)
        b = cls(d, torch.from_numpy(offsets).to(device))
        b._end_offset = int(offsets[-1])
        b._max_len = int(np.diff(offsets).max()) if len(offsets) > 1 else 0
        return b

    @classmethod
    def from_arrow(cls, arr, device="cuda"):
        """Batch from a pyarrow LargeBinary / LargeString array: the array's
        two buffers ARE the packed form of the C ABI -- text bytes back to back and int64
        offsets[n+1] -- so they are uploaded as they stand (a sliced array keeps its offsets; the
        data buffer is cut to the referenced range).  Nulls are treated as empty texts."""
        import numpy as np
        import pyarrow as pa
        import torch
        if isinstance(arr, pa.ChunkedArray):
            arr = arr.combine_chunks()
        if pa.types.is_binary(arr.type) or pa.types.is_string(arr.type):
            arr = arr.cast(pa.large_binary())
        if not (pa.types.is_large_binary(arr.type) or pa.types.is_large_string(arr.type)):
            raise MrxError("from_arrow needs a (large_)binary or (large_)string array")
        if arr.null_count:
            arr = arr.fill_null(b"")
        bufs = arr.buffers()   # [validity, offsets, data]
        n = len(arr)
        offs = np.frombuffer(bufs[1], dtype=np.int64, count=n + 1, offset=arr.offset * 8).copy()
        lo, hi = int(offs[0]), int(offs[-1])
        data = np.frombuffer(bufs[2], dtype=np.uint8, count=hi - lo, offset=lo) if hi > lo else np.zeros(0, np.uint8)
        offs -= lo
        d = torch.from_numpy(data.copy()).to(device) if data.size else torch.zeros(0, dtype=torch.uint8, device=device)
        b = cls(d, torch.from_numpy(offs).to(device))
        b._end_offset = int(offs[-1])
        b._max_len = int(np.diff(offs).max()) if n > 0 else 0
        return b

    def csr_offsets(self):
        """CSR offsets for the generic kernels (built on device for strided batches)."""
        import torch
        if self.offsets is not None:
            return self.offsets
        if self.lens is None and self.length == self.stride:
            return torch.arange(0, (self.n + 1) * self.stride, self.stride, dtype=torch.int64,
                                device=self.data.device)
        raise MrxError("this operation needs a CSR batch (strided batch with padding given)")


def _ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())


class CompiledRegex:
    """Compile once, match many batches (reference: matcher.mojo:929-1163)."""

    def __init__(self, pattern, lazydfa_semantics: bool = False, bitset_nfa: bool = False,
                 nfa_engine: bool = False, dfa_engine: bool = False):
        """lazydfa_semantics: MRX_COMPILE_LAZYDFA_SEMANTICS (include/mrx.h) -- NOT the
        reference's result for SIMPLE patterns; off by default.
        bitset_nfa: MRX_COMPILE_BITSET_NFA -- same results, LazyDFA walks run on the bitset NFA.
        nfa_engine: MRX_COMPILE_NFA_ENGINE -- NFAEngine itself as the Engine (regex.nfa, nfa.mojo:66-143,
        1733-1769), no hybrid router in front.
        dfa_engine: MRX_COMPILE_DFA_ENGINE -- compile_dfa_pattern's DFAEngine as the Engine (comptime API)."""
        self._lib = load_library()
        self.pattern = _b(pattern)
        self.lazydfa_semantics = bool(lazydfa_semantics)
        h = C.c_void_p()
        _check(self._lib.mrx_compile_ex(self.pattern, len(self.pattern),
                                        (1 if lazydfa_semantics else 0) | (2 if bitset_nfa else 0)
                                        | (4 if nfa_engine else 0) | (8 if dfa_engine else 0),
                                        C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.mrx_free(self._h)
                self._h = None
        except Exception:
            pass

    # -- introspection (get_engine_type :900-918, get_stats :1139-1163) ------------
    def get_engine_type(self) -> str:
        return self._lib.mrx_engine_type(self._h).decode()

    def get_stats(self) -> str:
        return self._lib.mrx_stats(self._h).decode("utf-8", "replace")

    def describe(self) -> str:
        need = self._lib.mrx_describe(self._h, None, 0)
        buf = C.create_string_buffer(need + 1)
        self._lib.mrx_describe(self._h, buf, need + 1)
        return buf.value.decode("utf-8", "replace")

    @property
    def num_groups(self) -> int:
        return self._lib.mrx_num_groups(self._h)

    # -- host-buffer batches --------------------------------------------------------
    def _spans_call(self, fn, texts):
        data, offsets = pack_texts(texts)
        n = len(offsets) - 1
        s = np.empty(n, np.int32)
        e = np.empty(n, np.int32)
        _check(fn(self._h, data.ctypes.data, offsets.ctypes.data, n, s.ctypes.data, e.ctypes.data))
        return s, e

    def match_first(self, texts) -> Tuple[np.ndarray, np.ndarray]:
        """regex.match_first per text: (start[n], end[n]), -1/-1 where none."""
        if isinstance(texts, DeviceBatch):
            return self._dev_spans(self._lib.mrx_match_first_dev, self._lib.mrx_match_first_strided_dev,
                                   texts)
        return self._spans_call(self._lib.mrx_match_first_batch, texts)

    def match_next(self, texts) -> Tuple[np.ndarray, np.ndarray]:
        """regex.search per text."""
        if isinstance(texts, DeviceBatch):
            return self._dev_spans(self._lib.mrx_search_dev, self._lib.mrx_search_strided_dev, texts)
        return self._spans_call(self._lib.mrx_search_batch, texts)

    search = match_next

    def is_match(self, texts):
        """CompiledRegex.is_match(text, 0): uint8[n] -- numpy for host texts, a device tensor for a
        DeviceBatch."""
        if isinstance(texts, DeviceBatch):   # device tensor uint8[n]
            import torch
            f = torch.empty(texts.n, dtype=torch.uint8, device=texts.data.device)
            if texts.offsets is not None:
                _check(self._lib.mrx_is_match_dev(self._h, _ptr(texts.data), _ptr(texts.offsets), texts.n, _ptr(f),
                                                  self._stream_ptr()))
            else:
                _check(self._lib.mrx_is_match_strided_dev(self._h, _ptr(texts.data), texts.stride, _ptr(texts.lens),
                                                          texts.length, texts.n, _ptr(f), self._stream_ptr()))
            return f
        data, offsets = pack_texts(texts)
        n = len(offsets) - 1
        f = np.empty(n, np.uint8)
        _check(self._lib.mrx_is_match_batch(self._h, data.ctypes.data, offsets.ctypes.data, n,
                                            f.ctypes.data))
        return f

    # -- the Engine / RegexMatcher seam with its `start` argument (engine.mojo:4-37) ------------------
    def _at(self, op: str, texts, start):
        """op in {"match_first", "search", "is_match"}; start: one int for all texts, or int32[n]
        (numpy / device tensor).  Host texts are uploaded; a DeviceBatch is used in place."""
        import torch
        batch = texts if isinstance(texts, DeviceBatch) else DeviceBatch.from_texts(texts)
        dev = batch.data.device
        d_starts = None
        s0 = 0
        if isinstance(start, (int, np.integer)):
            s0 = int(start)
        else:
            d_starts = start if isinstance(start, torch.Tensor) else torch.from_numpy(np.asarray(start, dtype=np.int32))
            d_starts = d_starts.to(device=dev, dtype=torch.int32).contiguous()
            if d_starts.numel() != batch.n:
                raise MrxError("starts needs one entry per text")
        csr = batch.offsets is not None
        lay = ([_ptr(batch.data), _ptr(batch.offsets), batch.n] if csr else
               [_ptr(batch.data), batch.stride, _ptr(batch.lens), batch.length, batch.n])
        if op == "is_match":
            f = torch.empty(batch.n, dtype=torch.uint8, device=dev)
            fn = self._lib.mrx_is_match_at_dev if csr else self._lib.mrx_is_match_at_strided_dev
            _check(fn(self._h, *lay, s0, _ptr(d_starts), _ptr(f), self._stream_ptr()))
            return f if isinstance(texts, DeviceBatch) else f.cpu().numpy()
        s = torch.empty(batch.n, dtype=torch.int32, device=dev)
        e = torch.empty(batch.n, dtype=torch.int32, device=dev)
        name = "mrx_%s_at_%sdev" % ("match_first" if op == "match_first" else "search", "" if csr else "strided_")
        _check(getattr(self._lib, name)(self._h, *lay, s0, _ptr(d_starts), _ptr(s), _ptr(e), self._stream_ptr()))
        return (s, e) if isinstance(texts, DeviceBatch) else (s.cpu().numpy(), e.cpu().numpy())

    def match_first_at(self, texts, start):
        """CompiledRegex.match_first(text, start) (matcher.mojo:1049-1062): a match beginning at start."""
        return self._at("match_first", texts, start)

    def match_next_at(self, texts, start):
        """CompiledRegex.match_next(text, start) (matcher.mojo:1064-1077): leftmost match from start on."""
        return self._at("search", texts, start)

    def is_match_at(self, texts, start):
        """CompiledRegex.is_match(text, start) (matcher.mojo:1103-1115)."""
        return self._at("is_match", texts, start)

    def test(self, texts):
        """CompiledRegex.test (matcher.mojo:1091-1101): does search() match.  bool[n]: numpy for host
        texts, a device tensor for a DeviceBatch."""
        s, _ = self.match_next(texts)
        if isinstance(texts, DeviceBatch):
            return s >= 0
        return (np.asarray(s) >= 0)

    def match_all(self, texts):
        """regex.findall per text: (counts_prefix int64[n+1], spans int32[total, 2])."""
        if isinstance(texts, DeviceBatch):
            return self._dev_findall(texts)
        data, offsets = pack_texts(texts)
        n = len(offsets) - 1
        prefix = np.zeros(n + 1, np.int64)
        cap = max(64, int(offsets[-1]) // 4 + n)
        while True:
            spans = np.empty((cap, 2), np.int32)
            total = C.c_int64(0)
            rc = self._lib.mrx_findall_batch(self._h, data.ctypes.data, offsets.ctypes.data, n,
                                             prefix.ctypes.data, spans.ctypes.data, cap,
                                             C.byref(total))
            if rc == MRX_E_CAPACITY:
                cap = int(total.value)
                continue
            _check(rc)
            return prefix, spans[: total.value]

    findall = match_all

    def findall_lists(self, texts) -> List[List[Tuple[int, int]]]:
        prefix, spans = self.match_all(texts)
        out = []
        for i in range(len(prefix) - 1):
            out.append([(int(a), int(b)) for a, b in spans[prefix[i]:prefix[i + 1]]])
        return out

    def split(self, texts, maxsplit: int = 0) -> List[List[bytes]]:
        """regex.split per text: the pieces between successive matches (mrx_split_batch)."""
        bs = [_b(t) for t in texts]
        data, offsets = pack_texts(bs)
        n = len(bs)
        prefix = np.zeros(n + 1, dtype=np.int64)
        cap = max(16, 2 * n + len(data) // 8)
        total = C.c_int64(0)
        while True:
            pieces = np.empty((cap, 2), dtype=np.int32)
            rc = self._lib.mrx_split_batch(self._h, data.ctypes.data, offsets.ctypes.data, n, int(maxsplit), prefix.ctypes.data,
                                           pieces.ctypes.data, cap, C.byref(total))
            if rc == MRX_E_CAPACITY and int(total.value) > cap:
                cap = int(total.value)
                continue
            _check(rc)
            break
        out = []
        for i, t in enumerate(bs):
            out.append([t[int(a):int(b)] for a, b in pieces[prefix[i]:prefix[i + 1]]])
        return out

    def split_dev(self, batch: "DeviceBatch", maxsplit: int = 0, piece_cap: Optional[int] = None):
        """regex.split of a device-resident batch: (piece_prefix int64[n + 1], pieces int32[total, 2], total)."""
        import torch
        dev = batch.data.device
        n = batch.n
        prefix = torch.empty(n + 1, dtype=torch.int64, device=dev)
        cap = piece_cap if piece_cap is not None else max(16, 2 * n + batch.data.numel() // 8)
        total = C.c_int64(0)
        while True:
            pieces = torch.empty((cap, 2), dtype=torch.int32, device=dev)
            if batch.offsets is not None:
                rc = self._lib.mrx_split_dev(self._h, _ptr(batch.data), _ptr(batch.offsets), n, int(maxsplit), _ptr(prefix),
                                             _ptr(pieces), cap, C.byref(total), self._stream_ptr())
            else:
                rc = self._lib.mrx_split_strided_dev(self._h, _ptr(batch.data), batch.stride, _ptr(batch.lens), batch.length, n,
                                                     int(maxsplit), _ptr(prefix), _ptr(pieces), cap, C.byref(total), self._stream_ptr())
            if rc == MRX_E_CAPACITY and piece_cap is None and int(total.value) > cap:
                cap = int(total.value)
                continue
            _check(rc)
            break
        return prefix, pieces, int(total.value)

    def captures(self, texts) -> np.ndarray:
        """search + capture groups, int32[n, g+1, 2] in the order the reference's
        NFAEngine._match_group appends them: groups 1..g, then group 0."""
        data, offsets = pack_texts(texts)
        n = len(offsets) - 1
        g = self.num_groups
        out = np.empty((n, g + 1, 2), np.int32)
        _check(self._lib.mrx_captures_batch(self._h, data.ctypes.data, offsets.ctypes.data, n,
                                            out.ctypes.data))
        return out

    def sub(self, repl, texts, count: int = 0) -> List[bytes]:
        repl = _b(repl)
        data, offsets = pack_texts(texts)
        n = len(offsets) - 1
        out_off = np.zeros(n + 1, np.int64)
        cap = max(64, int(offsets[-1]) * 2 + 16 * n)
        while True:
            out = np.empty(cap, np.uint8)
            total = C.c_int64(0)
            rc = self._lib.mrx_sub_batch(self._h, repl, len(repl), count, data.ctypes.data,
                                         offsets.ctypes.data, n, out_off.ctypes.data,
                                         out.ctypes.data, cap, C.byref(total))
            if rc == MRX_E_CAPACITY:
                cap = int(total.value)
                continue
            _check(rc)
            raw = out[: total.value].tobytes()
            return [raw[out_off[i]:out_off[i + 1]] for i in range(n)]

    def captures_dev(self, batch: "DeviceBatch"):
        """search + capture groups on a device-resident batch: int32[n, g+1, 2] (device tensor)."""
        import torch
        g = self.num_groups
        out = torch.empty((batch.n, g + 1, 2), dtype=torch.int32, device=batch.data.device)
        if batch.offsets is not None:
            _check(self._lib.mrx_captures_dev(self._h, _ptr(batch.data), _ptr(batch.offsets), batch.n, _ptr(out),
                                              self._stream_ptr()))
        else:
            _check(self._lib.mrx_captures_strided_dev(self._h, _ptr(batch.data), batch.stride, _ptr(batch.lens),
                                                      batch.length, batch.n, _ptr(out), self._stream_ptr()))
        return out

    def sub_dev(self, repl, batch: "DeviceBatch", count: int = 0, out_cap: Optional[int] = None):
        """regex.sub on a device-resident batch (CSR, or fixed pitch: mrx_sub_strided_dev):
        (out_offsets int64[n+1], out_data uint8[total])."""
        import torch
        repl = _b(repl)
        dev = batch.data.device
        off = batch.csr_offsets() if batch.offsets is not None else None
        cap = int(out_cap) if out_cap else int(batch.data.numel()) * 2 + 16 * batch.n + 64
        out_off = torch.empty(batch.n + 1, dtype=torch.int64, device=dev)
        while True:
            out = torch.empty(cap, dtype=torch.uint8, device=dev)
            total = C.c_int64(0)
            if off is None:
                rc = self._lib.mrx_sub_strided_dev(self._h, repl, len(repl), count, _ptr(batch.data), batch.stride,
                                                   _ptr(batch.lens), batch.length, batch.n, _ptr(out_off), _ptr(out),
                                                   cap, C.byref(total), self._stream_ptr())
            elif batch.offsets is not None and batch._end_offset is not None:
                rc = self._lib.mrx_sub_known_dev(self._h, repl, len(repl), count, _ptr(batch.data), _ptr(off), batch.n,
                                                 batch._end_offset, batch._max_len, _ptr(out_off), _ptr(out), cap,
                                                 C.byref(total), self._stream_ptr())
            else:
                rc = self._lib.mrx_sub_dev(self._h, repl, len(repl), count, _ptr(batch.data), _ptr(off), batch.n,
                                           _ptr(out_off), _ptr(out), cap, C.byref(total), self._stream_ptr())
            if rc == MRX_E_CAPACITY:
                cap = int(total.value)
                continue
            _check(rc)
            return out_off, out[: total.value]

    # -- device-resident batches (torch tensors) ------------------------------------
    def _stream_ptr(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _dev_spans(self, fn_csr, fn_strided, batch: DeviceBatch):
        import torch
        s = torch.empty(batch.n, dtype=torch.int32, device=batch.data.device)
        e = torch.empty(batch.n, dtype=torch.int32, device=batch.data.device)
        if batch.offsets is not None:
            _check(fn_csr(self._h, _ptr(batch.data), _ptr(batch.offsets), batch.n, _ptr(s), _ptr(e),
                          self._stream_ptr()))
        else:
            _check(fn_strided(self._h, _ptr(batch.data), batch.stride, _ptr(batch.lens), batch.length,
                              batch.n, _ptr(s), _ptr(e), self._stream_ptr()))
        return s, e

    def findall_async(self, batch: DeviceBatch, out):
        """Enqueue findall on the current stream without reading anything back.
        out = (counts_prefix int64[n+1], spans int32[cap, 2]) device tensors; the total
        is counts_prefix[n] once the stream has drained (check it against cap)."""
        prefix, spans = out
        if batch.offsets is not None and batch._end_offset is not None:
            rc = self._lib.mrx_findall_known_dev(self._h, _ptr(batch.data), _ptr(batch.offsets), batch.n,
                                                 batch._end_offset, batch._max_len, _ptr(prefix), _ptr(spans),
                                                 spans.shape[0], None, self._stream_ptr())
        elif batch.offsets is not None:
            rc = self._lib.mrx_findall_dev(self._h, _ptr(batch.data), _ptr(batch.offsets), batch.n,
                                           _ptr(prefix), _ptr(spans), spans.shape[0], None,
                                           self._stream_ptr())
        else:
            rc = self._lib.mrx_findall_strided_dev(self._h, _ptr(batch.data), batch.stride,
                                                   _ptr(batch.lens), batch.length, batch.n,
                                                   _ptr(prefix), _ptr(spans), spans.shape[0], None,
                                                   self._stream_ptr())
        _check(rc)

    def _dev_findall(self, batch: DeviceBatch, span_cap: Optional[int] = None, out=None):
        """Returns (counts_prefix int64[n+1], spans int32[cap, 2], total) on device."""
        import torch
        dev = batch.data.device
        if out is None:
            if span_cap is None:
                span_cap = max(64, batch.data.numel() // 8 + batch.n)
            prefix = torch.empty(batch.n + 1, dtype=torch.int64, device=dev)
            spans = torch.empty((span_cap, 2), dtype=torch.int32, device=dev)
        else:
            prefix, spans = out
            span_cap = spans.shape[0]
        total = C.c_int64(0)
        while True:
            if batch.offsets is not None and batch._end_offset is not None:
                rc = self._lib.mrx_findall_known_dev(self._h, _ptr(batch.data), _ptr(batch.offsets), batch.n,
                                                     batch._end_offset, batch._max_len, _ptr(prefix), _ptr(spans),
                                                     span_cap, C.byref(total), self._stream_ptr())
            elif batch.offsets is not None:
                rc = self._lib.mrx_findall_dev(self._h, _ptr(batch.data), _ptr(batch.offsets), batch.n,
                                               _ptr(prefix), _ptr(spans), span_cap, C.byref(total),
                                               self._stream_ptr())
            else:
                rc = self._lib.mrx_findall_strided_dev(self._h, _ptr(batch.data), batch.stride,
                                                       _ptr(batch.lens), batch.length, batch.n,
                                                       _ptr(prefix), _ptr(spans), span_cap,
                                                       C.byref(total), self._stream_ptr())
            if rc == MRX_E_CAPACITY and out is None:
                span_cap = int(total.value)
                spans = torch.empty((span_cap, 2), dtype=torch.int32, device=dev)
                continue
            _check(rc)
            return prefix, spans, int(total.value)

    def count(self, batch: DeviceBatch):
        import torch
        counts = torch.empty(batch.n, dtype=torch.int32, device=batch.data.device)
        if batch.offsets is not None:
            _check(self._lib.mrx_count_dev(self._h, _ptr(batch.data), _ptr(batch.offsets), batch.n,
                                           _ptr(counts), self._stream_ptr()))
        else:
            _check(self._lib.mrx_count_strided_dev(self._h, _ptr(batch.data), batch.stride, _ptr(batch.lens),
                                                   batch.length, batch.n, _ptr(counts), self._stream_ptr()))
        return counts


# ---------------------------------------------------------------------------------
# module-level API with the reference's process-wide cache (matcher.mojo:1166-1321)
# ---------------------------------------------------------------------------------
_CACHE = {}


def compile_regex(pattern, lazydfa_semantics: bool = False, bitset_nfa: bool = False,
                  nfa_engine: bool = False, dfa_engine: bool = False) -> CompiledRegex:
    key = (_b(pattern), bool(lazydfa_semantics), bool(bitset_nfa), bool(nfa_engine), bool(dfa_engine))
    c = _CACHE.get(key)
    if c is None:
        c = CompiledRegex(key[0], lazydfa_semantics, bitset_nfa, nfa_engine, dfa_engine)
        _CACHE[key] = c
    return c


def clear_regex_cache():
    _CACHE.clear()


def match_first(pattern, texts):
    return compile_regex(pattern).match_first(texts)


def search(pattern, texts):
    return compile_regex(pattern).match_next(texts)


def findall(pattern, texts):
    return compile_regex(pattern).match_all(texts)


def sub(pattern, repl, texts, count: int = 0) -> List[bytes]:
    return compile_regex(pattern).sub(repl, texts, count)


def split(pattern, texts, maxsplit: int = 0) -> List[List[bytes]]:
    """regex.split (matcher.mojo:1357-1393) through the C ABI (mrx_split_batch): pieces as byte ranges per text."""
    return compile_regex(pattern).split(texts, maxsplit)
