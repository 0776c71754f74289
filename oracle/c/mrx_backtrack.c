/* ORACLE (test infrastructure; never linked into or called by the product) -- the reference's recursive
 * backtracking matcher (NFAEngine, src/regex/nfa.mojo:66-1731) once more in plain C, so that the group fuzz can
 * run texts of a kilobyte and more (oracle/mrx_ref/backtrack.py, the pinned Python restatement, needs minutes per
 * pattern there).  Control flow is restated function by function, with the same file:line anchors as
 * backtrack.py; the byte predicates come in as three 256-entry tables per leaf, filled by backtrack.py's own
 * predicate functions (oracle/mrx_ref/cbacktrack.py), so the two halves cannot drift apart on membership:
 *   first[c]  the test the leaf matcher makes on the byte at str_i      (_match_element .. _match_range, :757-995)
 *   chr[c]    ASTNode.is_match_char                                     (ast.mojo:415-462)
 *   simd[c]   the predicate of the "SIMD" quantifier loop for this leaf (_apply_quantifier_simd, :1446-1647)
 * tests/test_oracle_c.py requires this file and backtrack.py to agree on every reference vector and on generated
 * patterns; parity claims rest on backtrack.py + the vectors, this file only makes long texts affordable. */
#define _GNU_SOURCE   /* memmem */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { T_RE = 0, T_ELEMENT = 1, T_WILDCARD = 2, T_SPACE = 3, T_DIGIT = 4, T_WORD = 5, T_RANGE = 6, T_START = 7,
       T_END = 8, T_OR = 9, T_GROUP = 10 };   /* = oracle/mrx_ref/frontend.py's numbering, checked by the loader */

typedef struct {
  int32_t type, min, max;
  int32_t capturing, group_id;
  int32_t value_len;          /* byte length of get_value() (0 = None) */
  int32_t nchildren, child0;  /* children = kids[child0 .. child0 + nchildren) */
  int32_t tbl;                /* index of this leaf's three tables, -1 for inner nodes */
} BtNode;

typedef struct {
  const BtNode* nodes;
  const int32_t* kids;
  const uint8_t* tables;      /* [ntbl][3][256] */
  int32_t root;               /* -1: the pattern did not parse (self.regex is None) */
  const uint8_t* pattern; int32_t pattern_len;
  const uint8_t* literal; int32_t literal_len;
  int32_t has_literal_optimization, ends_with_dotstar, starts_with_dotstar, is_prefix_literal;
} BtProg;

typedef struct { int32_t gid, start, end; } BtGroup;
typedef struct { BtGroup* v; int64_t n, cap; } BtGroups;

typedef struct { int ok; int64_t pos; } R;

static void groups_push(BtGroups* g, int32_t gid, int64_t s, int64_t e) {
  if (!g) return;
  if (g->n == g->cap) {
    g->cap = g->cap ? g->cap * 2 : 64;
    g->v = (BtGroup*)realloc(g->v, sizeof(BtGroup) * (size_t)g->cap);
  }
  g->v[g->n].gid = gid; g->v[g->n].start = (int32_t)s; g->v[g->n].end = (int32_t)e; g->n++;
}

typedef struct {
  const BtProg* p;
  const uint8_t* s;
  int64_t n;
  BtGroups* matches;
} Ctx;

static const uint8_t* tbl(const Ctx* c, const BtNode* nd, int which) { return c->p->tables + ((size_t)nd->tbl * 3 + which) * 256; }

static R match_node(Ctx* c, int32_t ni, int64_t i, int mfm, int64_t req);
static R match_sequence(Ctx* c, const BtNode* parent, int child_index, int64_t i, int mfm, int64_t req);

static int has_quantifier(const BtNode* nd) { return nd->min != 1 || nd->max != 1; }

/* ASTNode.is_match_char, ast.mojo:415-462 (position-dependent for the anchors; false for inner nodes) */
static int is_match_char(const Ctx* c, const BtNode* nd, int ch, int64_t str_i, int64_t str_len) {
  if (nd->type == T_START) return str_i == 0;
  if (nd->type == T_END) return str_i == str_len;
  if (nd->tbl < 0) return 0;
  return tbl(c, nd, 1)[ch];
}

/* ASTNode.is_simd_optimizable, ast.mojo:372-400 */
static int is_simd_optimizable(const BtNode* nd, int64_t min_matches, int64_t max_matches) {
  const int t = nd->type;
  if (!(t == T_SPACE || t == T_DIGIT || t == T_WORD || t == T_RANGE)) return 0;
  if (min_matches == 1 && max_matches == 1) return 0;
  if (max_matches == -1) {
    if (t == T_DIGIT || t == T_WORD || t == T_SPACE) return min_matches >= 1;
    return min_matches > 3;
  }
  if (max_matches > 8) return 1;
  if (t == T_RANGE && nd->value_len > 0) return nd->value_len > 8;
  return 0;
}

/* apply_quantifier_simd_generic (simd_ops.mojo:1308-1354) / _quantifier_*_loop (nfa.mojo:1672-1731) */
static R run_pred(const uint8_t* pred, const Ctx* c, int64_t str_i, int64_t min_matches, int64_t max_matches) {
  int64_t pos = str_i, count = 0;
  const int64_t actual_max = max_matches != -1 ? max_matches : c->n - str_i;
  while (pos < c->n && count < actual_max) {
    if (pred[c->s[pos]]) { ++count; ++pos; } else break;
  }
  R r; r.ok = count >= min_matches; r.pos = r.ok ? pos : str_i;
  return r;
}

/* _apply_quantifier, nfa.mojo:1375-1443 (+ _apply_quantifier_simd :1446-1647 through the simd table) */
static R apply_quantifier(Ctx* c, const BtNode* nd, int64_t i, int char_consumed, int mfm, int64_t req) {
  int64_t min_matches = nd->min, max_matches = nd->max;
  if (max_matches == -1) max_matches = c->n - i;
  R r;
  if (min_matches == 1 && max_matches == 1) { r.ok = 1; r.pos = i + char_consumed; return r; }
  if (is_simd_optimizable(nd, min_matches, max_matches)) {
    if (nd->type == T_RANGE && nd->value_len == 0) { r.ok = 0; r.pos = i; return r; }   /* `return (False, i)` tail */
    return run_pred(tbl(c, nd, 2), c, i, min_matches, max_matches);
  }
  int64_t count = 0, pos = i;
  while (count < max_matches && pos < c->n) {
    if (mfm && req >= 0 && pos > req + 50) break;
    if (is_match_char(c, nd, c->s[pos], pos, c->n)) { ++count; ++pos; } else break;
  }
  r.ok = count >= min_matches; r.pos = r.ok ? pos : i;
  return r;
}

/* the leaf matchers, nfa.mojo:757-995 */
static R match_leaf(Ctx* c, const BtNode* nd, int64_t i, int mfm, int64_t req) {
  R no; no.ok = 0; no.pos = i;
  const int zero_ok = (nd->type == T_DIGIT || nd->type == T_WORD) && nd->min == 0;   /* _match_digit / _match_word :841-927 */
  if (i >= c->n) return zero_ok ? apply_quantifier(c, nd, i, 0, mfm, req) : no;
  if (tbl(c, nd, 0)[c->s[i]]) return apply_quantifier(c, nd, i, 1, mfm, req);
  return zero_ok ? apply_quantifier(c, nd, i, 0, mfm, req) : no;
}

/* _try_match_count, nfa.mojo:1313-1349 */
static int64_t try_match_count(const Ctx* c, const BtNode* nd, int64_t i, int64_t count, int mfm, int64_t req) {
  int64_t pos = i, matched = 0;
  while (matched < count && pos < c->n) {
    if (mfm && req >= 0 && pos > req + 100) return -1;
    if (is_match_char(c, nd, c->s[pos], pos, c->n)) { ++matched; ++pos; } else return -1;
  }
  return matched == count ? pos - i : -1;
}

/* _match_with_backtracking, nfa.mojo:1231-1311 */
static R match_with_backtracking(Ctx* c, const BtNode* q, const BtNode* parent, int remaining_index, int64_t i, int mfm, int64_t req) {
  int64_t min_matches = q->min, max_matches = q->max;
  R no; no.ok = 0; no.pos = i;
  if (max_matches == -1) max_matches = c->n - i;
  if (min_matches == max_matches) {
    const int64_t consumed = try_match_count(c, q, i, min_matches, mfm, req);
    if (consumed >= 0) {
      R r = match_sequence(c, parent, remaining_index, i + consumed, mfm, req);
      if (r.ok) return r;
    }
    return no;
  }
  for (int64_t match_count = max_matches; match_count >= min_matches; --match_count) {
    const int64_t consumed = try_match_count(c, q, i, match_count, mfm, req);
    if (consumed >= 0) {
      const int64_t new_pos = i + consumed;
      if (mfm && req >= 0 && new_pos > req + 100) return no;
      R r = match_sequence(c, parent, remaining_index, new_pos, mfm, req);
      if (r.ok) return r;
    }
  }
  return no;
}

/* _match_sequence, nfa.mojo:1158-1224 */
static R match_sequence(Ctx* c, const BtNode* parent, int child_index, int64_t i, int mfm, int64_t req) {
  R r; r.ok = 1; r.pos = i;
  if (child_index >= parent->nchildren) return r;
  const int32_t* kids = c->p->kids + parent->child0;
  if (child_index == parent->nchildren - 1) return match_node(c, kids[child_index], i, mfm, req);
  const BtNode* first = c->p->nodes + kids[child_index];
  if (has_quantifier(first)) return match_with_backtracking(c, first, parent, child_index + 1, i, mfm, req);
  r = match_node(c, kids[child_index], i, mfm, req);
  if (!r.ok) { r.pos = i; return r; }
  return match_sequence(c, parent, child_index + 1, r.pos, mfm, req);
}

/* _match_group_with_quantifier, nfa.mojo:1105-1156 */
static R match_group_with_quantifier(Ctx* c, const BtNode* nd, int64_t i, int mfm, int64_t req) {
  int64_t min_matches = nd->min, max_matches = nd->max, current_pos = i, group_matches = 0;
  if (max_matches == -1) max_matches = c->n - i;
  while (group_matches < max_matches && current_pos <= c->n) {
    R r = match_sequence(c, nd, 0, current_pos, mfm, req);
    if (!r.ok) break;
    ++group_matches;
    current_pos = r.pos;
    if (mfm && req >= 0 && current_pos > req + 100) break;
    if (nd->capturing) groups_push(c->matches, nd->group_id >= 0 ? nd->group_id : 0, i, current_pos);
  }
  R out; out.ok = group_matches >= min_matches; out.pos = out.ok ? current_pos : i;
  return out;
}

/* _match_node, nfa.mojo:657-755 (+ _match_or :1019-1055, _match_group :1057-1103, _match_re :1351-1373) */
static R match_node(Ctx* c, int32_t ni, int64_t i, int mfm, int64_t req) {
  const BtNode* nd = c->p->nodes + ni;
  R r; r.ok = 0; r.pos = i;
  switch (nd->type) {
    case T_ELEMENT: case T_WILDCARD: case T_SPACE: case T_DIGIT: case T_WORD: case T_RANGE:
      return match_leaf(c, nd, i, mfm, req);
    case T_START: r.ok = i == 0; return r;
    case T_END: r.ok = i == c->n; return r;
    case T_OR: {
      if (nd->nchildren < 2) return r;
      const int32_t* kids = c->p->kids + nd->child0;
      R left = match_node(c, kids[0], i, mfm, req);
      if (left.ok) return left;
      return match_node(c, kids[1], i, mfm, req);
    }
    case T_GROUP: {
      if (has_quantifier(nd)) return match_group_with_quantifier(c, nd, i, mfm, req);
      R s = match_sequence(c, nd, 0, i, mfm, req);
      if (!s.ok) return r;
      if (nd->capturing) groups_push(c->matches, nd->group_id >= 0 ? nd->group_id : 0, i, s.pos);
      return s;
    }
    case T_RE:
      if (nd->nchildren == 0) { r.ok = 1; return r; }
      return match_node(c, (c->p->kids + nd->child0)[0], i, mfm, req);
    default: return r;
  }
}

/* bytes.find / bytes.rfind */
static int64_t find_lit(const uint8_t* s, int64_t n, const uint8_t* lit, int64_t m, int64_t start) {
  if (start > n) return -1;
  if (m == 0) return start;
  if (n - start < m) return -1;
  const uint8_t* hit = (const uint8_t*)memmem(s + start, (size_t)(n - start), lit, (size_t)m);
  return hit ? (int64_t)(hit - s) : -1;
}
static int64_t rfind_lit(const uint8_t* s, int64_t n, const uint8_t* lit, int64_t m) {
  if (m > n) return -1;
  for (int64_t p = n - m; p >= 0; --p)
    if (memcmp(s + p, lit, (size_t)m) == 0) return p;
  return -1;
}
static int has_newline(const uint8_t* s, int64_t n) { return memchr(s, '\n', (size_t)n) != NULL; }

/* _match_contains_literal, nfa.mojo:642-655 */
static int match_contains_literal(const BtProg* p, const uint8_t* s, int64_t n, int64_t start, int64_t end) {
  if (!p->has_literal_optimization || p->literal_len == 0) return 1;
  const int64_t pos = find_lit(s, n, p->literal, p->literal_len, start);
  return pos != -1 && pos + p->literal_len <= end;
}
static void search_literal(const BtProg* p, const uint8_t** lit, int64_t* m) {   /* _get_search_literal_bytes :157-167 */
  if (p->has_literal_optimization) { *lit = p->literal; *m = p->literal_len; }
  else { *lit = p->pattern; *m = p->pattern_len; }
}

/* match_first, nfa.mojo:342-389.  returns 1 and *end on a match at `start` */
int mrx_bt_match_first(const BtProg* p, const uint8_t* s, int64_t n, int64_t start, int64_t* end) {
  if (p->root < 0) return 0;
  Ctx c = {p, s, n, NULL};
  R r = match_node(&c, p->root, start, 1, start);
  *end = r.pos;
  return r.ok;
}

/* match_next (groups == NULL, nfa.mojo:391-498) / match_next_with_groups (groups != NULL, :500-574: no `.*` fast paths) */
int mrx_bt_match_next(const BtProg* p, const uint8_t* s, int64_t n, int64_t start, int64_t* ms, int64_t* me, BtGroups* groups) {
  if (p->root < 0) return 0;
  Ctx c = {p, s, n, groups};
  int64_t search_pos = start;
  if (!groups) {
    if (p->starts_with_dotstar && p->has_literal_optimization && !has_newline(s, n)) {
      const int64_t last = rfind_lit(s, n, p->literal, p->literal_len);   /* _find_last_literal :577-585 */
      if (last >= start && last >= 0) { *ms = start; *me = last + p->literal_len; return 1; }
      return 0;
    }
    if (p->ends_with_dotstar && p->has_literal_optimization && p->is_prefix_literal && !has_newline(s, n)) {
      const int64_t pos = find_lit(s, n, p->literal, p->literal_len, start);
      if (pos >= 0) { *ms = pos; *me = n; return 1; }
      return 0;
    }
  }
  if (p->has_literal_optimization) {
    const uint8_t* lit; int64_t m;
    search_literal(p, &lit, &m);
    while (search_pos <= n) {
      const int64_t literal_pos = find_lit(s, n, lit, m, search_pos);
      if (literal_pos == -1) return 0;
      int64_t try_pos = literal_pos;
      if (p->literal_len > 0 && !p->is_prefix_literal) { try_pos = literal_pos - p->pattern_len; if (try_pos < 0) try_pos = 0; }
      for (; try_pos <= literal_pos; ++try_pos) {
        if (groups) groups->n = 0;
        R r = match_node(&c, p->root, try_pos, 0, -1);
        if (r.ok && match_contains_literal(p, s, n, try_pos, r.pos)) { *ms = try_pos; *me = r.pos; return 1; }
      }
      search_pos = literal_pos + 1;
    }
  } else {
    for (; search_pos <= n; ++search_pos) {
      if (groups) groups->n = 0;
      R r = match_node(&c, p->root, search_pos, 0, -1);
      if (r.ok) { *ms = search_pos; *me = r.pos; return 1; }
    }
  }
  if (groups) groups->n = 0;
  return 0;
}

/* match_all, nfa.mojo:169-340.  spans[2 * k], spans[2 * k + 1]; returns the number of matches (all counted, the first
 * `cap` stored) */
int64_t mrx_bt_match_all(const BtProg* p, const uint8_t* s, int64_t n, int32_t* spans, int64_t cap) {
  int64_t cnt = 0;
#define EMIT(a, b) do { if (cnt < cap) { spans[2 * cnt] = (int32_t)(a); spans[2 * cnt + 1] = (int32_t)(b); } ++cnt; } while (0)
  if (p->root < 0) return 0;
  Ctx c = {p, s, n, NULL};
  int64_t current_pos = 0;
  if (p->starts_with_dotstar && p->has_literal_optimization && !has_newline(s, n)) {
    const int64_t last = rfind_lit(s, n, p->literal, p->literal_len);
    if (last >= current_pos && last >= 0) EMIT(current_pos, last + p->literal_len);
    return cnt;
  }
  if (p->ends_with_dotstar && p->has_literal_optimization && p->is_prefix_literal && !has_newline(s, n)) {
    if (current_pos < n) {
      const int64_t pos = find_lit(s, n, p->literal, p->literal_len, current_pos);
      if (pos != -1) EMIT(pos, n);
    }
    return cnt;
  }
  if (p->has_literal_optimization) {
    const uint8_t* lit; int64_t m;
    search_literal(p, &lit, &m);
    while (current_pos <= n) {
      const int64_t literal_pos = find_lit(s, n, lit, m, current_pos);
      if (literal_pos == -1) break;
      if (literal_pos < current_pos) { current_pos = literal_pos + 1; continue; }
      int64_t try_pos = literal_pos;
      if (p->literal_len > 0 && !p->is_prefix_literal) { try_pos = literal_pos - 10; if (try_pos < current_pos) try_pos = current_pos; }
      int found = 0;
      int64_t max_search_positions = literal_pos - try_pos + 1; if (max_search_positions > 5) max_search_positions = 5;
      int64_t search_count = 0;
      while (try_pos <= literal_pos && try_pos <= n && search_count < max_search_positions) {
        R r = match_node(&c, p->root, try_pos, 0, -1);
        if (r.ok && match_contains_literal(p, s, n, try_pos, r.pos)) {
          EMIT(try_pos, r.pos);
          current_pos = r.pos == try_pos ? try_pos + 1 : r.pos;
          found = 1;
          break;
        }
        ++try_pos; ++search_count;
      }
      if (!found) current_pos = literal_pos + 1;
    }
  } else {
    while (current_pos <= n) {
      R r = match_node(&c, p->root, current_pos, 0, -1);
      if (r.ok) { EMIT(current_pos, r.pos); current_pos = r.pos == current_pos ? current_pos + 1 : r.pos; }
      else ++current_pos;
    }
  }
#undef EMIT
  return cnt;
}

void mrx_bt_groups_free(BtGroups* g) { if (g && g->v) { free(g->v); g->v = NULL; g->n = g->cap = 0; } }
int32_t mrx_bt_node_size(void) { return (int32_t)sizeof(BtNode); }
int32_t mrx_bt_prog_size(void) { return (int32_t)sizeof(BtProg); }
