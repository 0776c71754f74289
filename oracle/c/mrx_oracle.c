/*
 * ORACLE -- test infrastructure only, never the product path.
 *
 * Plain-C restatement of the reference's DFAEngine matching loops, for batches
 * too large for the pure-Python oracle and as the CPU baseline ("port") timed by
 * bench.py.  Scalar, single-threaded, one text at a time -- like the reference
 * (src/regex/matcher.mojo:1206-1210).  The tables come from the Python oracle
 * (oracle/mrx_ref/dfa_engine.py) through ctypes; tests check C == Python.
 *
 * Restates, from the reference (paths relative to its checkout):
 *   find_first_nibble_match     src/regex/simd_ops.mojo:566-648 (+ :90-134)
 *   count_consecutive_matches   src/regex/simd_ops.mojo:651-786
 *   simd_search / verify_match  src/regex/simd_ops.mojo:937-1024
 *   _try_match_at_position      src/regex/dfa.mojo:1906-2026
 *   _try_match_simd             src/regex/dfa.mojo:2133-2197
 *   match_first / match_next    src/regex/dfa.mojo:1852-1903
 *   _optimized_simd_search      src/regex/dfa.mojo:2200-2253
 *   match_all                   src/regex/dfa.mojo:2028-2130
 * Parity status: pinned through the Python oracle (tests/test_oracle_c.py checks
 * this file against it on the reference's own vectors and on random batches).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

typedef struct {
  int32_t nstates;
  const int32_t* trans;      /* [nstates][256], -1 = none (DFAState, dfa.mojo:215-254) */
  const uint8_t* accepting;  /* [nstates] */
  int32_t has_start_anchor, has_end_anchor, is_pure_literal;
  int32_t has_simd_matcher, simd_scan_eligible;
  const uint8_t* lookup;     /* [256] CharacterClassSIMD.lookup_table */
  int32_t num_ranges;        /* 0 => nibble-table scan */
  const uint8_t* lo_tbl;     /* [16] */
  const uint8_t* hi_tbl;     /* [16] */
  const uint8_t* literal;
  int32_t literal_len;
  int32_t simd_width;        /* SIMD_WIDTH of the modelled reference build */
  const uint8_t* ranges;     /* [2 * num_ranges] lo, hi of the class's contiguous ranges (1..3), else NULL */
} mrx_dfa;

/* The reference scans for the first-class byte 32 bytes at a time with range compares
 * (simd_ops.mojo:585-631: (c - lo) <= (hi - lo) per range, OR, movemask) and counts runs the same
 * way (simd_ops.mojo:682-692).  The CPU baseline does the same when built for an AVX2 machine; the
 * results are those of the scalar loops (every branch is exact for 1..3 ranges). */
#ifdef __AVX2__
#include <immintrin.h>
static inline uint32_t class_mask32(const mrx_dfa* d, const uint8_t* p) {
  const __m256i v = _mm256_loadu_si256((const __m256i*)p);
  __m256i in = _mm256_setzero_si256();
  for (int r = 0; r < d->num_ranges; ++r) {
    const __m256i x = _mm256_sub_epi8(v, _mm256_set1_epi8((char)d->ranges[2 * r]));
    const __m256i span = _mm256_set1_epi8((char)(d->ranges[2 * r + 1] - d->ranges[2 * r]));
    in = _mm256_or_si256(in, _mm256_cmpeq_epi8(_mm256_min_epu8(x, span), x));
  }
  return (uint32_t)_mm256_movemask_epi8(in);
}
#endif

typedef struct { int64_t s, e; } span_t;

/* simd_ops.mojo:566-648 */
static int64_t find_first_nibble_match(const mrx_dfa* d, const uint8_t* t, int64_t start, int64_t len) {
  int64_t pos = start;
  if (d->num_ranges >= 1 && d->num_ranges <= 3) {
#ifdef __AVX2__
    if (d->ranges)
      for (; pos + 32 <= len; pos += 32) {
        const uint32_t m = class_mask32(d, t + pos);
        if (m) return pos + __builtin_ctz(m);
      }
#endif
    for (; pos < len; ++pos)
      if (d->lookup[t[pos]]) return pos;
    return -1;
  }
  /* find_first_in_nibble_tables, simd_ops.mojo:90-134: the SIMD chunks use the
   * (inexact) nibble test, the scalar tail the exact table */
  const int64_t W = d->simd_width;
  while (pos + W <= len) {
    for (int64_t i = 0; i < W; ++i) {
      const uint8_t c = t[pos + i];
      if (d->lo_tbl[c & 15] & d->hi_tbl[c >> 4]) return pos + i;
    }
    pos += W;
  }
  for (; pos < len; ++pos)
    if (d->lookup[t[pos]]) return pos;
  return -1;
}

/* simd_ops.mojo:651-786 (every branch is exact) */
static int64_t count_consecutive_matches(const mrx_dfa* d, const uint8_t* t, int64_t start, int64_t len) {
  int64_t pos = start;
#ifdef __AVX2__
  if (d->ranges && d->num_ranges >= 1 && d->num_ranges <= 3)
    for (; pos + 32 <= len; pos += 32) {
      const uint32_t m = ~class_mask32(d, t + pos);
      if (m) return pos + __builtin_ctz(m) - start;
    }
#endif
  while (pos < len && d->lookup[t[pos]]) ++pos;
  return pos - start;
}

/* simd_ops.mojo:937-960 */
static int verify_match(const uint8_t* p, int64_t plen, const uint8_t* t, int64_t len, int64_t pos) {
  if (pos + plen > len) return 0;
  return memcmp(t + pos, p, (size_t)plen) == 0;
}

/* simd_ops.mojo:963-1024 */
static int64_t simd_search(const uint8_t* p, int64_t plen, const uint8_t* t, int64_t len, int64_t start) {
  if (plen == 0) return start;
  if (plen == 1) {
    for (int64_t pos = start; pos < len; ++pos)
      if (t[pos] == p[0]) return pos;
    return -1;
  }
  for (int64_t pos = start; pos + plen <= len; ++pos)
    if (t[pos] == p[0] && t[pos + plen - 1] == p[plen - 1] && verify_match(p, plen, t, len, pos))
      return pos;
  return -1;
}

/* dfa.mojo:2133-2197 */
static int try_match_simd(const mrx_dfa* d, const uint8_t* t, int64_t len, int64_t start_pos, span_t* out) {
  if (!d->has_simd_matcher || d->nstates == 0) return 0;
  const int start_acc = d->accepting[0];
  if (!start_acc && !d->simd_scan_eligible) return 0;
  const int64_t n = count_consecutive_matches(d, t, start_pos, len);
  int valid = 0;
  int64_t end = start_pos + n;
  if (n == 0) { if (start_acc) { valid = 1; end = start_pos; } }
  else valid = 1;
  if (!valid) return 0;
  if (d->has_end_anchor && end != len) return 0;
  out->s = start_pos; out->e = end;
  return 1;
}

/* dfa.mojo:1906-2026 */
static int try_match_at(const mrx_dfa* d, const uint8_t* t, int64_t len, int64_t start_pos,
                        int exact, span_t* out) {
  if (start_pos > len) return 0;
  if (d->is_pure_literal) {
    const int64_t plen = d->literal_len;
    if (exact) {
      if (!verify_match(d->literal, plen, t, len, start_pos)) return 0;
      out->s = start_pos; out->e = start_pos + plen;
      return 1;
    }
    const int64_t pos = simd_search(d->literal, plen, t, len, start_pos);
    if (pos < 0) return 0;
    out->s = pos; out->e = pos + plen;
    return 1;
  }
  if (d->has_simd_matcher && d->nstates > 0 && (d->simd_scan_eligible || d->accepting[0])) {
    if (try_match_simd(d, t, len, start_pos, out)) return 1;
  }
  if (start_pos == len) {
    if (d->nstates > 0 && d->accepting[0]) { out->s = out->e = start_pos; return 1; }
    return 0;
  }
  int32_t cur = 0;
  int64_t pos = start_pos;
  int64_t last = -1;
  if (d->nstates > 0 && d->accepting[cur]) last = pos;
  while (pos < len) {  /* the hot loop, dfa.mojo:1996-2009 */
    const int32_t nx = d->trans[(size_t)cur * 256 + t[pos]];
    if (nx == -1) break;
    cur = nx;
    ++pos;
    if (d->accepting[cur]) last = pos;
  }
  if (pos == len && d->accepting[cur]) last = pos;
  if (last == -1) return 0;
  if (d->has_end_anchor && last != len) return 0;
  out->s = start_pos; out->e = last;
  return 1;
}

/* dfa.mojo:2200-2253 */
static int optimized_simd_search(const mrx_dfa* d, const uint8_t* t, int64_t len, int64_t start, span_t* out) {
  if (!d->has_simd_matcher) return 0;
  int64_t pos = start;
  if (d->simd_scan_eligible) {
    while (pos < len) {
      const int64_t mp = find_first_nibble_match(d, t, pos, len);
      if (mp < 0) return 0;
      const int64_t ml = count_consecutive_matches(d, t, mp, len);
      if (ml > 0) {
        const int64_t me = mp + ml;
        if (d->has_end_anchor && me != len) { pos = me; continue; }
        out->s = mp; out->e = me;
        return 1;
      }
      pos = mp + 1;
    }
    return 0;
  }
  while (pos < len) {
    const int64_t fp = find_first_nibble_match(d, t, pos, len);
    if (fp < 0) return 0;
    if (try_match_at(d, t, len, fp, 0, out)) return 1;
    pos = fp + 1;
  }
  return 0;
}

/* dfa.mojo:1852-1872 */
int mrx_oracle_match_first(const mrx_dfa* d, const uint8_t* t, int64_t len, int64_t start, int64_t* s, int64_t* e) {
  span_t r;
  if (d->has_start_anchor && start > 0) return 0;
  if (!try_match_at(d, t, len, start, 1, &r)) return 0;
  *s = r.s; *e = r.e;
  return 1;
}

/* dfa.mojo:1875-1903 */
int mrx_oracle_match_next(const mrx_dfa* d, const uint8_t* t, int64_t len, int64_t start, int64_t* s, int64_t* e) {
  span_t r;
  int ok = 0;
  if (d->has_start_anchor) {
    ok = (start == 0) ? try_match_at(d, t, len, 0, 0, &r) : 0;
  } else if (d->has_simd_matcher && !d->has_end_anchor) {
    ok = optimized_simd_search(d, t, len, start, &r);
  } else {
    for (int64_t p = start; p <= len; ++p)
      if (try_match_at(d, t, len, p, 0, &r)) { ok = 1; break; }
  }
  if (!ok) return 0;
  *s = r.s; *e = r.e;
  return 1;
}

/* dfa.mojo:2028-2130; returns the number of matches, writes at most cap spans */
int64_t mrx_oracle_match_all(const mrx_dfa* d, const uint8_t* t, int64_t len, int32_t* spans, int64_t cap) {
  int64_t k = 0;
  span_t r;
#define EMIT(S, E) do { if (k < cap) { spans[2 * k] = (int32_t)(S); spans[2 * k + 1] = (int32_t)(E); } ++k; } while (0)
  if (d->has_start_anchor || d->has_end_anchor) {
    int64_t s, e;
    if (mrx_oracle_match_next(d, t, len, 0, &s, &e)) EMIT(s, e);
    return k;
  }
  int64_t pos = 0;
  if (d->is_pure_literal) {
    const int64_t plen = d->literal_len;
    while (pos <= len - plen) {
      const int64_t hit = simd_search(d->literal, plen, t, len, pos);
      if (hit < 0) break;
      EMIT(hit, hit + plen);
      pos = hit + plen;
    }
    return k;
  }
  if (d->has_simd_matcher && d->nstates > 0) {
    if (d->simd_scan_eligible) {
      while (pos < len) {
        const int64_t mp = find_first_nibble_match(d, t, pos, len);
        if (mp < 0) break;
        const int64_t ml = count_consecutive_matches(d, t, mp, len);
        if (ml > 0) { EMIT(mp, mp + ml); pos = mp + ml; }
        else pos = mp + 1;
      }
      return k;
    }
    while (pos < len) {
      const int64_t np = find_first_nibble_match(d, t, pos, len);
      if (np < 0) break;
      pos = np;
      if (try_match_at(d, t, len, pos, 0, &r)) {
        EMIT(r.s, r.e);
        pos = (r.e == r.s) ? pos + 1 : r.e;
      } else {
        ++pos;
      }
    }
    return k;
  }
  while (pos <= len) {
    if (try_match_at(d, t, len, pos, 0, &r)) {
      EMIT(r.s, r.e);
      pos = (r.e == r.s) ? pos + 1 : r.e;
    } else {
      ++pos;
    }
  }
#undef EMIT
  return k;
}

/* ---- batches (CSR: text i = data[offsets[i] .. offsets[i+1])) ---------------------- */
/* findall over a batch; counts[i] = matches in text i; spans packed in text order up
 * to cap spans.  Returns the total number of matches. */
int64_t mrx_oracle_findall_batch(const mrx_dfa* d, const uint8_t* data, const int64_t* offsets,
                                 int64_t n, int32_t* counts, int32_t* spans, int64_t cap) {
  int64_t total = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t room = cap > total ? cap - total : 0;
    const int64_t k = mrx_oracle_match_all(d, data + offsets[i], offsets[i + 1] - offsets[i],
                                           spans ? spans + 2 * total : (int32_t*)0, spans ? room : 0);
    if (counts) counts[i] = (int32_t)k;
    total += k;
  }
  return total;
}

/* Same scan, texts split over `threads` host threads (counts only; the reference itself is
 * single threaded -- this is the "all host cores" leg of SURVEY.md 8(d)).  dynamic schedule:
 * the adversarial texts cost ~1000x the others under restart-per-position search. */
int64_t mrx_oracle_count_batch_mt(const mrx_dfa* d, const uint8_t* data, const int64_t* offsets,
                                  int64_t n, int32_t* counts, int threads) {
  int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) reduction(+ : total)
  for (int64_t i = 0; i < n; ++i) {
    const int64_t k = mrx_oracle_match_all(d, data + offsets[i], offsets[i + 1] - offsets[i], (int32_t*)0, 0);
    if (counts) counts[i] = (int32_t)k;
    total += k;
  }
  return total;
}

/* findall of the whole batch on `threads` host threads, the spans of text i written at spans + 2 * prefix[i]
 * (room: prefix[i + 1] - prefix[i]; a text with more matches than that keeps what fits) and its true count at
 * counts[i]: the full-size parity tests hand in the DEVICE's offsets, so that one pass gives both the counts to
 * compare with them and the spans in the device's layout.  Same per-text function as above (match_all,
 * dfa.mojo:2028-2130). */
int64_t mrx_oracle_findall_at_mt(const mrx_dfa* d, const uint8_t* data, const int64_t* offsets, int64_t n,
                                 const int64_t* prefix, int32_t* spans, int32_t* counts, int threads) {
  int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) reduction(+ : total)
  for (int64_t i = 0; i < n; ++i) {
    const int64_t room = prefix[i + 1] - prefix[i];
    const int64_t k = mrx_oracle_match_all(d, data + offsets[i], offsets[i + 1] - offsets[i],
                                           spans + 2 * (prefix[i] - prefix[0]), room > 0 ? room : 0);
    counts[i] = (int32_t)k;
    total += k;
  }
  return total;
}

/* which: 0 = match_first (kept only if it starts at 0, matcher.mojo:1411-1415), 1 = search */
void mrx_oracle_span_batch(const mrx_dfa* d, int which, const uint8_t* data, const int64_t* offsets,
                           int64_t n, int32_t* start, int32_t* end) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t s = -1, e = -1;
    const uint8_t* t = data + offsets[i];
    const int64_t len = offsets[i + 1] - offsets[i];
    int ok = which == 0 ? mrx_oracle_match_first(d, t, len, 0, &s, &e)
                        : mrx_oracle_match_next(d, t, len, 0, &s, &e);
    if (ok && which == 0 && s != 0) ok = 0;
    start[i] = ok ? (int32_t)s : -1;
    end[i] = ok ? (int32_t)e : -1;
  }
}

/* bytes the reference examines for match_first: min(len, consumed + 1) */
int64_t mrx_oracle_match_first_bytes(const mrx_dfa* d, const uint8_t* data, const int64_t* offsets, int64_t n) {
  int64_t total = 0;
  for (int64_t i = 0; i < n; ++i) {
    const uint8_t* t = data + offsets[i];
    const int64_t len = offsets[i + 1] - offsets[i];
    int32_t cur = 0;
    int64_t pos = 0;
    while (pos < len) {
      const int32_t nx = d->trans[(size_t)cur * 256 + t[pos]];
      if (nx == -1) break;
      cur = nx;
      ++pos;
    }
    total += (pos + 1 < len) ? pos + 1 : len;
  }
  return total;
}
