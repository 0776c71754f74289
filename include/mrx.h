/*
 * mrx.h -- C ABI of the MI355X batch regex matcher (libmrx_hip.so).
 *
 * Drop-in boundary for the byte-scanning hot path of msaelices/mojo-regex.
 * The reference has no FFI; the seams this ABI replaces are its two Mojo traits
 * and the module functions built on them (all paths relative to the reference):
 *
 *   Engine        src/regex/engine.mojo:4-37      match_first / match_all
 *   RegexMatcher  src/regex/matcher.mojo:181-209  match_first / match_all (+ match_next)
 *   module API    src/regex/matcher.mojo:1325-1415 (search, findall, split, match_first)
 *                 src/regex/matcher.mojo:1857-1917 (sub)
 *   CompiledRegex src/regex/matcher.mojo:929-1163  (compile once, match many)
 *
 * The reference matches ONE text per call on the CPU.  This library matches a
 * BATCH of texts per call on the GPU (one wavefront lane per text) and returns,
 * for every text, exactly what the reference call returns for it: byte offsets,
 * half-open [start, end), leftmost start / longest end, restart-per-position
 * search (src/regex/dfa.mojo:1875-2130).
 *
 * Conventions (mirroring SURVEY.md 8(b)):
 *   - Ownership: the caller owns every input and output buffer; the library owns
 *     the opaque compiled handle until mrx_free().  A handle is immutable after
 *     mrx_compile(), so concurrent batch calls on one handle are safe.
 *   - Errors: integer status; mrx_last_error() gives the message of the last
 *     failing call on this thread.  Pattern syntax errors carry the reference's
 *     own message text (src/regex/lexer.mojo:150-152, parser.mojo:224-237,340,399).
 *     Matching never fails on data: "no match" is start = end = -1 / count 0.
 *   - Text is raw bytes (no UTF-8 awareness), as in the reference (all tables
 *     are 256 wide, src/regex/dfa.mojo:215-254).
 *   - Span offsets are int32 and relative to the start of each text (texts up to
 *     2 GiB - 1); batch offsets are int64.
 *   - Batch layout: `data` holds the texts back to back, text i occupies
 *     data[offsets[i] .. offsets[i+1]).  The *_strided entry points take texts
 *     at a fixed pitch instead: text i starts at data + i*stride and has length
 *     lens[i] (or `len` for all i when lens == NULL).  Every layout runs on the
 *     streaming kernels when the plan allows it; stride % 16 == 0 with a 16-byte
 *     aligned `data` is the fastest form (no per-text alignment frame).
 *   - Pointers named d_* are DEVICE pointers (HBM); everything else is host
 *     memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     The *_dev entry points enqueue work and return without synchronising
 *     unless they have to report a total (documented per function).
 *   - No CPU fallback exists: without a usable HIP device every matching entry
 *     point fails with MRX_E_NO_DEVICE.
 */
#ifndef MRX_H
#define MRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mrx_handle mrx_handle;

enum {
  MRX_OK = 0,
  MRX_E_SYNTAX = 1,       /* the reference's parser raises on this pattern          */
  MRX_E_UNSUPPORTED = 2,  /* refused, never approximated: backtracker programs beyond */
                          /* the flat form's limits, tables beyond their budgets ... */
                          /* mrx_last_error() carries the reason                    */
  MRX_E_NO_DEVICE = 3,    /* no HIP device / HIP runtime error                      */
  MRX_E_CAPACITY = 4,     /* output buffer too small; *total holds the need         */
  MRX_E_ARGUMENT = 5
};

/* ---- compile (replaces CompiledRegex(pattern), matcher.mojo:964-978) ---------- */
/* A CONTRACT OF THIS LIBRARY'S OWN -- `$` on the LazyDFA search (search / findall / count / sub of a pattern with `$`
 * that the reference routes to NFAMatcher: matcher.mojo:401-431).  Upstream's LazyDFA decides whether `$` holds when a
 * (state, byte) transition is FIRST computed -- it holds iff that byte is the last of the text in hand -- and caches the
 * answer for every later use, in that call and in every later call on the same CompiledRegex (pikevm.mojo:869-942):
 * `^[a-z]+$` finds "abc" in a fresh process, nothing in "abcabc", and after that nothing in "abc" either.  A batch
 * has no call order, so every text of every call is answered AS A FRESHLY COMPILED PATTERN WOULD answer it: the
 * cache is empty when a text begins and is carried through that text's walks (findall, the match_next calls of one
 * sub) exactly as upstream carries it.  The oracle restates this (oracle/mrx_ref), tests/test_oracle_golden.py holds
 * hand traces; no reference test pins it ("parity-unpinned").  match_first / is_match of `$` patterns run on
 * the reference's OnePass automaton and have no such history. */
int mrx_compile(const char* pattern, size_t pattern_len, mrx_handle** out);
/* Options.  MRX_COMPILE_LAZYDFA_SEMANTICS routes the pattern as the reference does when
 * DFAEngine compilation fails (matcher.mojo:666-672): NFAMatcher / LazyDFA, leftmost-longest
 * over the PikeVM program (pikevm.mojo:754-867).  It is NOT what the reference returns for
 * SIMPLE patterns -- e.g. it honours the `+` of (x|y|foo|bar)+ that the DFA alternation
 * compiler drops (dfa.mojo:873-928) -- and exists so that config-5 numbers can be reported
 * under both readings.  mrx_describe() shows the option.
 *
 * MRX_COMPILE_BITSET_NFA: patterns the reference routes to LazyDFA (pikevm.mojo:664-987)
 * normally run on its eagerly determinised table; with this option -- and always when that
 * table would exceed 4096 states -- the walk runs on the bitset NFA instead (state = bit
 * mask of live PikeVM positions, up to 256).  Results are identical; only the kernel differs.
 *
 * MRX_COMPILE_NFA_ENGINE: the handle is the reference's NFAEngine used directly as the Engine
 * (engine.mojo:4-37, nfa.mojo:66-143) -- what `regex.nfa.match_first / findall`
 * (nfa.mojo:1733-1769) and the reference's tests/test_nfa.mojo drive -- instead of the
 * HybridMatcher router: greedy backtracking, first alternative wins, NFAEngine's literal
 * prefilter and `.*` fast paths, none of HybridMatcher's shortcuts.  Served by the flat program
 * of the backtracking matcher; MRX_E_UNSUPPORTED at the first matching call when the pattern
 * exceeds that form (16 nesting levels, 30 open choices, 240 items).
 *
 * MRX_COMPILE_DFA_ENGINE: likewise, the DFAEngine that compile_dfa_pattern(parse(pattern))
 * returns (dfa.mojo:2385-2496), as the comptime API (comptime_regex.mojo:59-87, 176-233) and
 * the reference's tests/test_dfa.mojo use it: no classifier, no exact-literal / prefilter /
 * required-byte shortcut in front.  MRX_E_UNSUPPORTED at compile time, with the dispatcher's
 * message, when no shape compiler takes the pattern. */
enum { MRX_COMPILE_LAZYDFA_SEMANTICS = 1, MRX_COMPILE_BITSET_NFA = 2, MRX_COMPILE_NFA_ENGINE = 4,
       MRX_COMPILE_DFA_ENGINE = 8 };
int mrx_compile_ex(const char* pattern, size_t pattern_len, uint32_t options, mrx_handle** out);
void mrx_free(mrx_handle* h);
const char* mrx_last_error(void);
/* HybridMatcher.get_engine_type(), matcher.mojo:900-918: "DFA", "NFA", "+Prefilter"... */
const char* mrx_engine_type(const mrx_handle* h);
/* CompiledRegex.get_stats(), matcher.mojo:1139-1163 */
const char* mrx_stats(const mrx_handle* h);
/* Text dump of the compiled tables (states, transitions, flags, kernel plan).
 * Returns the number of bytes needed (excluding NUL); writes at most cap. */
size_t mrx_describe(const mrx_handle* h, char* buf, size_t cap);
/* number of capture groups usable by mrx_captures_* / group references in mrx_sub_*:
 * the fixed-width (\d{N}) form (matcher.mojo:1002-1035) or the general groups of
 * NFAEngine.match_next_with_groups (nfa.mojo:500-574), in _match_group order; 0 if none */
int mrx_num_groups(const mrx_handle* h);

/* ---- device-resident batches (the measured path) --------------------------- */
/* regex.match_first(pattern, text), matcher.mojo:1396-1415: anchored at 0.
 * d_start[i] / d_end[i] = span or -1/-1. */
int mrx_match_first_dev(const mrx_handle* h, const uint8_t* d_data,
                        const int64_t* d_offsets, int64_t n,
                        int32_t* d_start, int32_t* d_end, void* stream);
/* regex.search(pattern, text), matcher.mojo:1325-1338 (= match_next(text, 0)). */
int mrx_search_dev(const mrx_handle* h, const uint8_t* d_data,
                   const int64_t* d_offsets, int64_t n,
                   int32_t* d_start, int32_t* d_end, void* stream);
/* Same two operations for texts at a fixed pitch (see header comment). */
int mrx_match_first_strided_dev(const mrx_handle* h, const uint8_t* d_data, int64_t stride,
                                const int32_t* d_lens, int32_t len, int64_t n,
                                int32_t* d_start, int32_t* d_end, void* stream);
int mrx_search_strided_dev(const mrx_handle* h, const uint8_t* d_data, int64_t stride,
                           const int32_t* d_lens, int32_t len, int64_t n,
                           int32_t* d_start, int32_t* d_end, void* stream);
/* CompiledRegex.is_match(text, 0), matcher.mojo:1103-1115 (DFAEngine.is_match
 * quirk included, dfa.mojo:1815-1849).  d_flag[i] = 0/1. */
int mrx_is_match_dev(const mrx_handle* h, const uint8_t* d_data,
                     const int64_t* d_offsets, int64_t n, uint8_t* d_flag,
                     void* stream);
int mrx_is_match_strided_dev(const mrx_handle* h, const uint8_t* d_data, int64_t stride,
                             const int32_t* d_lens, int32_t len, int64_t n, uint8_t* d_flag,
                             void* stream);
/* The same three operations from a start position -- Engine.match_first(text, start)
 * (src/regex/engine.mojo:4-37), RegexMatcher / CompiledRegex.match_first / match_next / is_match
 * (matcher.mojo:181-209, 1049-1115): text i is matched from start (all texts) or d_starts[i]
 * (d_starts != NULL; int32[n] on the device).  Results are offsets from the beginning of the text, as
 * the reference's Match carries them.  Reference rules kept: a '^' pattern on the DFA or OnePass route
 * answers None for start > 0 (dfa.mojo:1866-1867, 1887-1891; onepass.mojo:445) while the LazyDFA treats
 * '^' as satisfied at every start (pikevm.mojo:714); start == len matches the empty rest; start > len
 * gives None, except that LazyDFA / OnePass match_first (and is_match, and DFAEngine.is_match with a
 * first-byte matcher) report the empty match (start, start) when the start state accepts, as upstream
 * does (pikevm.mojo:820-867, dfa.mojo:1832-1836).  start < 0 (undefined upstream) gives None.
 * match_first here is the engine-level operation: a match, if any, begins AT start. */
int mrx_match_first_at_dev(const mrx_handle* h, const uint8_t* d_data, const int64_t* d_offsets, int64_t n,
                           int32_t start, const int32_t* d_starts, int32_t* d_start, int32_t* d_end,
                           void* stream);
int mrx_search_at_dev(const mrx_handle* h, const uint8_t* d_data, const int64_t* d_offsets, int64_t n,
                      int32_t start, const int32_t* d_starts, int32_t* d_start, int32_t* d_end, void* stream);
int mrx_is_match_at_dev(const mrx_handle* h, const uint8_t* d_data, const int64_t* d_offsets, int64_t n,
                        int32_t start, const int32_t* d_starts, uint8_t* d_flag, void* stream);
int mrx_match_first_at_strided_dev(const mrx_handle* h, const uint8_t* d_data, int64_t stride,
                                   const int32_t* d_lens, int32_t len, int64_t n, int32_t start,
                                   const int32_t* d_starts, int32_t* d_start, int32_t* d_end, void* stream);
int mrx_search_at_strided_dev(const mrx_handle* h, const uint8_t* d_data, int64_t stride,
                              const int32_t* d_lens, int32_t len, int64_t n, int32_t start,
                              const int32_t* d_starts, int32_t* d_start, int32_t* d_end, void* stream);
int mrx_is_match_at_strided_dev(const mrx_handle* h, const uint8_t* d_data, int64_t stride,
                                const int32_t* d_lens, int32_t len, int64_t n, int32_t start,
                                const int32_t* d_starts, uint8_t* d_flag, void* stream);
/* regex.findall(pattern, text), matcher.mojo:1341-1354.
 * d_counts_prefix[n+1]: exclusive prefix sum of matches per text (CSR);
 * d_spans[2*k], d_spans[2*k+1] = start, end of match k (text-relative), in text
 * order then match order.  span_cap = capacity of d_spans in spans.  d_spans must be 8-byte aligned (a span is
 * stored as one 8-byte word; hipMalloc and every framework allocator give far more).
 * Synchronises the stream once to return *total; MRX_E_CAPACITY if
 * *total > span_cap (nothing is ever written to d_spans beyond capacity).
 * total == NULL: nothing is read back and the call returns without synchronising;
 * d_counts_prefix[n] holds the total once the stream has drained (compare it with
 * span_cap before using the spans). */
int mrx_findall_dev(const mrx_handle* h, const uint8_t* d_data,
                    const int64_t* d_offsets, int64_t n,
                    int64_t* d_counts_prefix, int32_t* d_spans, int64_t span_cap,
                    int64_t* total, void* stream);
/* Same for a caller that knows its offsets (it built them on the host, or holds an Arrow array's): end_offset =
 * d_offsets[n], max_text_len = the longest text's length -- both may be upper bounds, neither may be too small
 * (they size scratch the kernels index with the real offsets).  mrx_findall_dev has to read the two from the device
 * -- one small kernel and one stream synchronisation before its scan can be enqueued -- which this entry point
 * spares: with total == NULL the whole call is asynchronous. */
int mrx_findall_known_dev(const mrx_handle* h, const uint8_t* d_data,
                          const int64_t* d_offsets, int64_t n, int64_t end_offset, int64_t max_text_len,
                          int64_t* d_counts_prefix, int32_t* d_spans, int64_t span_cap,
                          int64_t* total, void* stream);
/* Same, texts at a fixed pitch (see header comment).  d_lens may be NULL. */
int mrx_findall_strided_dev(const mrx_handle* h, const uint8_t* d_data,
                            int64_t stride, const int32_t* d_lens, int32_t len,
                            int64_t n, int64_t* d_counts_prefix, int32_t* d_spans,
                            int64_t span_cap, int64_t* total, void* stream);
/* Scan only: matches per text and nothing else (no span output). */
int mrx_count_dev(const mrx_handle* h, const uint8_t* d_data,
                  const int64_t* d_offsets, int64_t n, int32_t* d_counts,
                  void* stream);
int mrx_count_strided_dev(const mrx_handle* h, const uint8_t* d_data, int64_t stride,
                          const int32_t* d_lens, int32_t len, int64_t n, int32_t* d_counts,
                          void* stream);
/* search + capture groups, in the order NFAEngine._match_group appends them
 * (src/regex/nfa.mojo:1057-1103): groups 1..g, then group 0 (whole match).
 * d_spans[(i*(g+1) + k)*2 + {0,1}]; -1 when text i has no match.
 * Fixed-width (\d{N}) groups run on the streaming kernel; other group structures on the flat-program
 * backtracker (refused only beyond its limits: 16 nesting levels, 30 open choices). */
int mrx_captures_dev(const mrx_handle* h, const uint8_t* d_data,
                     const int64_t* d_offsets, int64_t n, int32_t* d_spans,
                     void* stream);
int mrx_captures_strided_dev(const mrx_handle* h, const uint8_t* d_data, int64_t stride,
                             const int32_t* d_lens, int32_t len, int64_t n, int32_t* d_spans,
                             void* stream);
/* regex.sub(pattern, repl, text, count), matcher.mojo:1679-1854.
 * d_out_offsets[n+1] (CSR of output bytes), d_out_data (capacity out_cap bytes).
 * Waits for the sizes to return *total_bytes (MRX_E_CAPACITY if it exceeds out_cap); the output bytes
 * themselves are written by work enqueued on `stream`, like the results of every other _dev call. */
int mrx_sub_dev(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count,
                const uint8_t* d_data, const int64_t* d_offsets, int64_t n,
                int64_t* d_out_offsets, uint8_t* d_out_data, int64_t out_cap,
                int64_t* total_bytes, void* stream);
/* Same for a caller that knows its offsets (see mrx_findall_known_dev: end_offset = d_offsets[n], max_text_len = the
 * longest text; upper bounds are fine, neither may be too small): spares the small kernel and the stream
 * synchronisation with which mrx_sub_dev reads the two from the device before anything else can be enqueued. */
int mrx_sub_known_dev(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count,
                      const uint8_t* d_data, const int64_t* d_offsets, int64_t n, int64_t end_offset,
                      int64_t max_text_len, int64_t* d_out_offsets, uint8_t* d_out_data, int64_t out_cap,
                      int64_t* total_bytes, void* stream);

/* Same, texts at a fixed pitch (round 3; every other operation already had this form).  Rows without padding
 * (len == stride, d_lens == NULL) take every fast path of mrx_sub_dev; padded rows run on the lane-per-text kernels. */
int mrx_sub_strided_dev(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count,
                        const uint8_t* d_data, int64_t stride, const int32_t* d_lens, int32_t len, int64_t n,
                        int64_t* d_out_offsets, uint8_t* d_out_data, int64_t out_cap,
                        int64_t* total_bytes, void* stream);

/* regex.split (matcher.mojo:1357-1393): the text between successive non-overlapping matches, the matches themselves
 * removed; a separator at the very start or end, or two adjacent ones, yields an empty piece.  maxsplit == 0: no limit;
 * > 0: at most that many splits per text, the rest of the text is the last piece; < 0: no split at all (the whole text
 * is the only piece) -- the reference's loop `if maxsplit != 0 and splits_done >= maxsplit: break`.
 *   d_piece_prefix[n + 1]  CSR offsets of the texts' pieces (text i has min(matches, maxsplit) + 1 of them)
 *   d_pieces[piece_cap][2] byte ranges [start, end) within the piece's own text, text order then position
 * *total (host, may be NULL) = number of pieces; MRX_E_CAPACITY when piece_cap does not hold them (what fits is not
 * written then; *total holds the need when the limit is off, a lower bound otherwise).  One stream synchronisation. */
int mrx_split_dev(const mrx_handle* h, const uint8_t* d_data, const int64_t* d_offsets, int64_t n, int64_t maxsplit,
                  int64_t* d_piece_prefix, int32_t* d_pieces, int64_t piece_cap, int64_t* total, void* stream);
int mrx_split_strided_dev(const mrx_handle* h, const uint8_t* d_data, int64_t stride, const int32_t* d_lens, int32_t len,
                          int64_t n, int64_t maxsplit, int64_t* d_piece_prefix, int32_t* d_pieces, int64_t piece_cap,
                          int64_t* total, void* stream);

/* ---- host-buffer convenience wrappers (copy in, run, copy out) -------------- */
int mrx_match_first_batch(const mrx_handle* h, const uint8_t* data,
                          const int64_t* offsets, int64_t n, int32_t* start,
                          int32_t* end);
int mrx_search_batch(const mrx_handle* h, const uint8_t* data, const int64_t* offsets,
                     int64_t n, int32_t* start, int32_t* end);
int mrx_is_match_batch(const mrx_handle* h, const uint8_t* data, const int64_t* offsets,
                       int64_t n, uint8_t* flag);
int mrx_split_batch(const mrx_handle* h, const uint8_t* data, const int64_t* offsets, int64_t n, int64_t maxsplit,
                    int64_t* piece_prefix, int32_t* pieces, int64_t piece_cap, int64_t* total);
int mrx_findall_batch(const mrx_handle* h, const uint8_t* data, const int64_t* offsets,
                      int64_t n, int64_t* counts_prefix, int32_t* spans,
                      int64_t span_cap, int64_t* total);
int mrx_captures_batch(const mrx_handle* h, const uint8_t* data, const int64_t* offsets,
                       int64_t n, int32_t* spans);
int mrx_sub_batch(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count,
                  const uint8_t* data, const int64_t* offsets, int64_t n,
                  int64_t* out_offsets, uint8_t* out_data, int64_t out_cap,
                  int64_t* total_bytes);

/* Per-call scratch (counts, event records, block sums) is kept in a grow-only arena per calling
 * thread and stream and reused by the next call on that stream.  mrx_release_scratch() frees the
 * calling thread's arenas (it synchronises their streams first).  Size: findall needs up to one
 * byte of records per text byte (+ 4 KiB per 64 texts); on the stepper kernels with texts of 2 KiB
 * and more, two bytes of span slots per text byte.  The arena settles within two calls of a new
 * batch shape (a call that had to grow it is followed by one that merges its chunks).
 *
 * Batch shape and kernel choice (all automatic, results never depend on it): one lane per text by
 * default; batches of few long texts put a wavefront on each text (stepper plans) or cut the texts
 * into pieces at bytes after which the scan does not depend on its past (streaming plans); ragged
 * CSR batches with a few texts far longer than the rest are handled the same way for findall.
 * CSR entry points read the batch's byte count (and longest text) back once per call; the strided
 * entry points with total == NULL never synchronise. */
void mrx_release_scratch(void);

const char* mrx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MRX_H */
