// Device-side matching semantics shared by every kernel in mrx_kernels.hip.
//
// One wavefront lane owns one text.  The functions below restate, per lane, the
// reference's matching control flow so that every (pattern, text) pair yields the
// reference's spans:
//   walk()              DFAEngine._try_match_at_position table walk  src/regex/dfa.mojo:1979-2024
//                       LazyDFA._run_lazy                            src/regex/pikevm.mojo:819-867
//   try_match_at()      DFAEngine._try_match_at_position             dfa.mojo:1906-2026
//   try_match_simd()    DFAEngine._try_match_simd                    dfa.mojo:2133-2197
//   engine_match_next() DFAEngine.match_next/_optimized_simd_search  dfa.mojo:1875-1903, 2200-2253
//                       LazyDFA.match_next                           pikevm.mojo:754-780
//   hybrid_*()          HybridMatcher.match_first/next/all           matcher.mojo:733-898
//   for_each_match()    DFAEngine.match_all / LazyDFA.match_all      dfa.mojo:2028-2130, pikevm.mojo:782-817
//   sub_text()          _sub_impl_with_repl                          matcher.mojo:1679-1854
// Tables live in LDS (staged once per workgroup from the plan blob).
#pragma once
#include <hip/hip_runtime.h>

#include "mrx_plan.hpp"

namespace mrx {

struct Ctx {
  DevPlan p;
  const uint8_t* cls;     // [256] byte -> class
  const uint8_t* first;   // [256] first-class / first-byte filter
  const uint16_t* trans;  // [nstates][ncls], 0xFFFF dead, bit15 = target accepts
  const uint8_t* lit;     // engine / exact literal
  const uint8_t* pre;     // prefilter literal
  // bitset NFA (PF_BITSET)
  const uint8_t* bs_cls;       // [256] byte -> mask row
  const uint64_t* bs_mask;     // [bs_ncls][bs_nw] positions that consume a byte of the class
  const uint64_t* bs_follow;   // [bs_npos][bs_nw] closure after position i
  // backtracking matcher as a flat program (DevPlan::bt_*)
  const BtItem* bt_items;
  const uint8_t* bt_tbl;       // 32-byte membership bitmaps, 3 per leaf
  const uint8_t* bt_lit;       // NFAEngine.literal_prefix
};

// One lane's text.  Reads go through an 8-byte register window: the generic kernels
// walk a text mostly forwards, so one aligned 8-byte global load serves 8 bytes
// instead of one dependent byte load per step.  The window is aligned on the
// ADDRESS, so it may cover up to 7 bytes before/after the text inside the same
// aligned word (never another page); those bytes are loaded but never returned.
constexpr int kPreUnknown = (int)0x80000000;
struct Text {
  const uint8_t* ptr;
  int len;
  // what a pass over the whole batch found out about this text before the lane-per-text kernel started
  // (k_litscan, for the backtracking matcher's literal prefilter): first occurrence of the plan's literal
  // (-1 none, kPreUnknown: no such pass ran) and its last occurrence << 1 | "the text holds a newline"
  int pre_first = kPreUnknown, pre_last_nl = 0;
  mutable uint64_t win = 0;
  mutable int win_lo = 0x40000000;  // index of the window's first byte; invalid at start
  // PF_LAZY_END plans: the LazyDFA's transition cache as this text's call has filled it so far -- bit s of lz_mid /
  // lz_end = the transition of state s on the text's LAST byte value was first computed inside the text / while
  // the last byte was consumed (walk_lazy_end)
  mutable uint64_t lz_mid = 0, lz_end = 0;
  __device__ __forceinline__ Text(const uint8_t* p, int n) : ptr(p), len(n) {}
  __device__ __forceinline__ int at(int i) const {
    unsigned d = (unsigned)(i - win_lo);
    if (d >= 8u) {
      const uintptr_t a = (uintptr_t)(ptr + i);
      win = *(const uint64_t*)(a & ~(uintptr_t)7);
      d = (unsigned)(a & 7);
      win_lo = i - (int)d;
    }
    return (int)((win >> (8 * d)) & 0xFFu);
  }
};

__device__ __forceinline__ bool flag(const Ctx& c, uint32_t f) { return (c.p.flags & f) != 0; }

// leftmost occurrence of needle at or after start (simd_search, simd_ops.mojo:963-1024;
// String.find as used at matcher.mojo:775,835 and prefilter.mojo:420-430)
__device__ inline int find_literal(const uint8_t* needle, int nlen, const Text& t, int start) {
  if (nlen == 0) return start;
  if (start < 0) start = 0;
  const int last = t.len - nlen;
  const int n0 = needle[0];
  for (int pos = start; pos <= last; ++pos) {
    if (t.at(pos) != n0) continue;
    int k = 1;
    while (k < nlen && t.at(pos + k) == needle[k]) ++k;
    if (k == nlen) return pos;
  }
  return -1;
}

// ---- NFAEngine's backtracking matcher on its flat program (BtProg, mrx_engines.hpp) ---------------
// Spans of the capture groups 0..9 of one attempt; -1 = the group did not close.  (Plain per-lane arrays: keeping
// the twenty values in registers through select chains was measured 15 % slower on the group-swapping sub of the
// reference's benchmark list -- the interpreter is short of registers as it is.)
struct BtCaps {
  int s[10], e[10];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int g = 0; g < 10; ++g) s[g] = e[g] = -1;
  }
  __device__ __forceinline__ void set(int g, int a, int b) { s[g] = a; e[g] = b; }
  __device__ __forceinline__ int gs(int g) const { return s[g]; }
  __device__ __forceinline__ int ge(int g) const { return e[g]; }
};
// One int per nesting level (group starts, loop counters)
struct BtLevels {
  int v[17];
  __device__ __forceinline__ void set(int d, int x) { v[d < 17 ? d : 16] = x; }
  __device__ __forceinline__ int get(int d) const { return v[d < 17 ? d : 16]; }
};
__device__ __forceinline__ bool bt_in(const Ctx& c, int tbl, int which, int byte) {
  return (c.bt_tbl[(tbl * 3 + which) * 32 + (byte >> 3)] >> (byte & 7)) & 1;
}
// consecutive bytes of the set from pos on, at most maxc; positions beyond `stop_after` are not looked at
__device__ inline int bt_run(const Ctx& c, const Text& t, int tbl, int which, int pos, int maxc, int stop_after) {
  int k = 0;
  while (k < maxc && pos + k < t.len && pos + k <= stop_after && bt_in(c, tbl, which, t.at(pos + k))) ++k;
  return k;
}
// One attempt: NFAEngine._match_node(root, text, start, matches, match_first_mode, required_start_pos),
// nfa.mojo:657-1443 on the flat program.  Returns the end of the match or -1; caps receives the groups
// that closed (the last closing of a group wins, as the reference's append-only list read back to front).
constexpr int kBtChoices = 32, kBtDepth = 17;
// STACK = false: the plan is a deterministic chain (DevPlan::bt_flags bit 5, see mrx_plan.cpp): no ALT / LOOP items,
// and a shorter count of a quantified leaf can never rescue the sequence behind it, so a failure is final -- the
// choice stack, its arrays and its retry loop are compiled out.
template <bool STACK>
__device__ inline int bt_match_at(const Ctx& c, const Text& t, int start, BtCaps& caps, bool mfm, int req) {
  const int n = t.len, nitems = c.p.bt_nitems;
  // choice stack: the newest entry lives in registers (top_*), the rest in per-lane arrays -- a greedy leaf
  // that is retried count by count ('.*' in front of something that fails) then never touches the arrays,
  // which the compiler keeps in scratch memory (dynamic indexing): 3.7 ms -> 0.x ms on one such 1 KiB text
  uint8_t ch_ip[kBtChoices], ch_depth[kBtChoices];
  int ch_pos[kBtChoices], ch_cnt[kBtChoices];
  int top_ip = 0, top_depth = 0, top_pos = 0, top_cnt = 0;
  bool have_top = false;
  BtLevels gstart;        // OPEN / LOOP: where the group began, per nesting depth
  BtLevels lcount;        // LOOP: repetitions matched so far, per nesting depth
  int ip = 0, pos = start, depth = 0, sp = 0;
  // ALT / LOOP leave a mark on the choice stack (cnt = kBtAltMark / kBtLoopMark): popping it on a failure means
  // "branch A failed: try B" / "this repetition failed: the loop is over"
  constexpr int kBtAltMark = -0x40000000, kBtLoopMark = -0x40000001;
  auto push_entry = [&](int e_ip, int e_depth, int e_pos, int e_cnt) {
    if (have_top) { ch_ip[sp] = (uint8_t)top_ip; ch_depth[sp] = (uint8_t)top_depth; ch_pos[sp] = top_pos; ch_cnt[sp] = top_cnt; ++sp; }
    top_ip = e_ip; top_depth = e_depth; top_pos = e_pos; top_cnt = e_cnt; have_top = true;
  };
  // drop every entry made deeper than `d` (a group / branch / repetition has returned: no way back in)
  auto cut_above = [&](int d) {
    if (have_top && top_depth > d) have_top = false;
    if (!have_top) while (sp > 0 && ch_depth[sp - 1] > d) --sp;
  };
  // the newest entry (a mark the caller knows is there) is dropped
  auto pop_mark = [&]() {
    if (have_top) have_top = false; else if (sp > 0) --sp;
  };
  const int far100 = (mfm && req >= 0) ? req + 100 : 0x7FFFFFFF;   // match_first_mode cut-offs
  const int far50 = (mfm && req >= 0) ? req + 50 : 0x7FFFFFFF;
  bool failing = false;
  while (true) {
    if (failing) {   // back to the innermost open choice with a smaller count left (nfa.mojo:1276-1309)
      if constexpr (!STACK) return -1;
      bool resumed = false;
      while (have_top || sp > 0) {
        if (!have_top) { --sp; top_ip = ch_ip[sp]; top_depth = ch_depth[sp]; top_pos = ch_pos[sp]; top_cnt = ch_cnt[sp]; }
        have_top = false;   // popped
        const BtItem it = c.bt_items[top_ip];
        if (top_cnt == kBtAltMark) {   // branch A failed: branch B from the same position
          pos = top_pos; depth = top_depth + 1; ip = it.min;
          resumed = true;
          break;
        }
        if (top_cnt == kBtLoopMark) {   // this repetition failed: the loop ends with what it has
          depth = top_depth;
          if (lcount.get(depth) < it.min) continue;   // too few: the group fails
          pos = top_pos; ip = it.max;
          resumed = true;
          break;
        }
        int cnt = top_cnt - 1;
        if (cnt < it.min) continue;
        if (top_pos + cnt > far100) continue;   // "new_pos > required_start_pos + 100": the choice is given up
        // The item behind the leaf is a plain one-byte leaf ('.*' in front of a literal): counts at which that byte
        // does not match would each come straight back here (set pos, fetch the item, test, fail, pop) -- step over
        // them in place.  (A one-byte leaf fails without side effects; the cut-off above only ever applies to the
        // first count tried, the later ones are smaller.)
        if (top_ip + 1 < nitems) {
          const BtItem nx = c.bt_items[top_ip + 1];
          if (nx.kind == BT_LEAF && nx.min == 1 && nx.max == 1)
            while (cnt >= it.min && !(top_pos + cnt < n && bt_in(c, nx.tbl, 0, t.at(top_pos + cnt)))) --cnt;
          if (cnt < it.min) continue;
        }
        pos = top_pos + cnt;
        ip = top_ip + 1;
        depth = top_depth;
        if (cnt > it.min) { top_cnt = cnt; have_top = true; }   // back on the stack, still in registers
        resumed = true;
        break;
      }
      if (!resumed) return -1;
      failing = false;
      continue;
    }
    if (ip >= nitems) return pos;
    const BtItem it = c.bt_items[ip];
    if (it.kind == BT_START) { if (pos != 0) failing = true; else ++ip; continue; }
    if (it.kind == BT_END) { if (pos != n) failing = true; else ++ip; continue; }
    if (it.kind == BT_OPEN) { gstart.set(depth, pos); ++depth; ++ip; continue; }
    if (it.kind == BT_FAIL) { failing = true; continue; }
    if constexpr (!STACK)
      if (it.kind == BT_ALT || it.kind == BT_ALT_END || it.kind == BT_ALT_CLOSE || it.kind == BT_LOOP || it.kind == BT_LOOP_END)
        return -1;   // (never in a chain program)
    if (it.kind == BT_ALT) {
      if (sp + (have_top ? 1 : 0) >= kBtChoices) return -1;
      push_entry(ip, depth, pos, kBtAltMark);
      ++depth; ++ip;
      continue;
    }
    if (it.kind == BT_ALT_END) {   // branch A matched: its choices and the mark go, B is skipped
      --depth;
      cut_above(depth);
      pop_mark();
      ip = it.max;
      continue;
    }
    if (it.kind == BT_ALT_CLOSE) { --depth; cut_above(depth); ++ip; continue; }
    if (it.kind == BT_LOOP || it.kind == BT_LOOP_END) {
      int lp = ip;   // the LOOP item
      if (it.kind == BT_LOOP_END) {   // a repetition matched: its choices and its mark go
        --depth;
        cut_above(depth);
        pop_mark();
        lp = it.min;
        const int reps = lcount.get(depth) + 1;
        lcount.set(depth, reps);
        const BtItem li = c.bt_items[lp];
        if (mfm && req >= 0 && pos > req + 100) {   // nfa.mojo:1143-1144: the loop stops here, before the span is recorded
          if (reps >= li.min) ip = li.max; else failing = true;
          continue;
        }
        if ((li.flags & BTF_CAPTURING) && li.gid >= 0 && li.gid < 10) caps.set(li.gid, gstart.get(depth), pos);
      } else {
        gstart.set(depth, pos); lcount.set(depth, 0);
      }
      const BtItem li = c.bt_items[lp];
      const BtItem le = c.bt_items[li.max - 1];          // its LOOP_END carries max
      const int maxr = le.max == -1 ? n - gstart.get(depth) : le.max;
      const int reps_now = lcount.get(depth);
      if (reps_now < maxr && pos <= n) {               // one more repetition
        if (sp + (have_top ? 1 : 0) >= kBtChoices) return -1;
        push_entry(lp, depth, pos, kBtLoopMark);
        ++depth;
        ip = lp + 1;
      } else if (reps_now >= li.min) {
        ip = li.max;
      } else {
        failing = true;
      }
      continue;
    }
    if (it.kind == BT_CLOSE) {
      --depth;
      // the group's sequence has returned: no way back in
      cut_above(depth);
      if ((it.flags & BTF_CAPTURING) && it.gid >= 0 && it.gid < 10) caps.set(it.gid, gstart.get(depth), pos);
      ++ip;
      continue;
    }
    // LEAF
    const int maxc = it.max == -1 ? n - pos : it.max;
    if ((it.flags & BTF_QUANT) && !(it.flags & BTF_LAST)) {
      // _match_with_backtracking: the largest count _try_match_count accepts, smaller ones on failure
      const int r = bt_run(c, t, it.tbl, 1, pos, maxc, far100);
      int cnt;
      if (it.min == maxc) {
        if (r < it.min) { failing = true; continue; }
        cnt = it.min;
      } else {
        cnt = r < maxc ? r : maxc;
        if (cnt < it.min) { failing = true; continue; }
        if (pos + cnt > far100) { failing = true; continue; }
        if constexpr (STACK) if (cnt > it.min) {
          if (sp + (have_top ? 1 : 0) >= kBtChoices) return -1;   // (the host refuses programs that could get here)
          push_entry(ip, depth, pos, cnt);
        }
      }
      pos += cnt;
      ++ip;
      continue;
    }
    // the leaf matcher itself (nfa.mojo:757-995) and _apply_quantifier (nfa.mojo:1375-1443)
    int consumed;
    if (pos >= n) {
      if ((it.flags & BTF_ZERO_OK) && it.min == 0) consumed = 0; else { failing = true; continue; }
    } else if (bt_in(c, it.tbl, 0, t.at(pos))) {
      consumed = 1;
    } else if ((it.flags & BTF_ZERO_OK) && it.min == 0) {
      consumed = 0;
    } else { failing = true; continue; }
    if (it.min == 1 && maxc == 1) { pos += consumed; ++ip; continue; }
    const bool cached = (it.flags & BTF_SIMD_TYPE) && (maxc > 8 || (it.flags & BTF_RANGE_LONG));
    const int cnt = cached ? bt_run(c, t, it.tbl, 2, pos, maxc, 0x7FFFFFFF) : bt_run(c, t, it.tbl, 1, pos, maxc, far50);
    if (cnt < it.min) { failing = true; continue; }
    pos += cnt;
    ++ip;
  }
}
// String.find(literal, start) for the plan's literal: answered from the batch-wide pass where it can be
// (no occurrence at all; or the first one, when start does not lie behind it)
__device__ inline int bt_find_literal(const Ctx& c, const Text& t, int start) {
  if (t.pre_first != kPreUnknown && c.p.bt_lit_len > 0) {
    if (t.pre_first < 0) return -1;
    if (start <= t.pre_first) return t.pre_first;
  }
  return find_literal(c.bt_lit, c.p.bt_lit_len, t, start);
}
// NFAEngine._match_contains_literal, nfa.mojo:642-655
__device__ inline bool bt_contains_literal(const Ctx& c, const Text& t, int start, int end) {
  if (!(c.p.bt_flags & 1) || c.p.bt_lit_len == 0) return true;
  const int pos = bt_find_literal(c, t, start);
  return pos >= 0 && pos + c.p.bt_lit_len <= end;
}
// NFAEngine.match_next_with_groups, nfa.mojo:500-574
template <bool STACK>
__device__ inline bool bt_match_next_with_groups(const Ctx& c, const Text& t, int start, int& ms, int& me, BtCaps& caps) {
  int search_pos = start;
  if (c.p.bt_flags & 1) {   // literal prefilter
    while (search_pos <= t.len) {
      const int lp = bt_find_literal(c, t, search_pos);
      if (lp < 0) return false;
      int try_pos = lp;
      if (c.p.bt_lit_len > 0 && !(c.p.bt_flags & 2)) try_pos = lp - c.p.bt_pattern_len > 0 ? lp - c.p.bt_pattern_len : 0;
      while (try_pos <= lp) {
        caps.clear();
        const int end = bt_match_at<STACK>(c, t, try_pos, caps, false, -1);
        if (end >= 0 && bt_contains_literal(c, t, try_pos, end)) { ms = try_pos; me = end; return true; }
        ++try_pos;
      }
      search_pos = lp + 1;
    }
    return false;
  }
  while (search_pos <= t.len) {
    caps.clear();
    const int end = bt_match_at<STACK>(c, t, search_pos, caps, false, -1);
    if (end >= 0) { ms = search_pos; me = end; return true; }
    ++search_pos;
  }
  return false;
}

// NFAEngine.match_first, nfa.mojo:342-389 (match_first_mode, required_start_pos = start)
template <bool STACK>
__device__ inline bool bt_engine_match_first(const Ctx& c, const Text& t, int start, int& ms, int& me) {
  BtCaps caps;
  const int end = bt_match_at<STACK>(c, t, start, caps, true, start);
  if (end < 0) return false;
  ms = start; me = end;
  return true;
}
__device__ inline bool bt_has_newline(const Text& t) {
  if (t.pre_first != kPreUnknown) return (t.pre_last_nl & 1) != 0;
  for (int i = 0; i < t.len; ++i)
    if (t.at(i) == '\n') return true;
  return false;
}
// String.rfind(literal) (NFAEngine._find_last_literal, nfa.mojo:577-585)
__device__ inline int bt_rfind_literal(const Ctx& c, const Text& t) {
  if (t.pre_first != kPreUnknown) return t.pre_last_nl >> 1;   // (-1 << 1 | nl) >> 1 == -1
  const int ll = c.p.bt_lit_len;
  for (int pos = t.len - ll; pos >= 0; --pos) {
    int k = 0;
    while (k < ll && t.at(pos + k) == c.bt_lit[k]) ++k;
    if (k == ll) return pos;
  }
  return -1;
}
// NFAEngine.match_next, nfa.mojo:391-498.  starts_dotstar / ends_dotstar: DevPlan::bt_flags bits 2 / 3.
template <bool STACK>
__device__ inline bool bt_engine_match_next(const Ctx& c, const Text& t, int start, int& ms, int& me) {
  const bool lit_opt = (c.p.bt_flags & 1) != 0, prefix_lit = (c.p.bt_flags & 2) != 0;
  if ((c.p.bt_flags & 4) && lit_opt && !bt_has_newline(t)) {   // .* prefix with a literal behind it
    const int last = bt_rfind_literal(c, t);
    if (last >= start && last >= 0) { ms = start; me = last + c.p.bt_lit_len; return true; }
    return false;
  }
  if ((c.p.bt_flags & 8) && lit_opt && prefix_lit && !bt_has_newline(t)) {   // LITERAL.*: to the end of the text
    const int pos = bt_find_literal(c, t, start);
    if (pos >= 0) { ms = pos; me = t.len; return true; }
    return false;
  }
  BtCaps caps;
  int search_pos = start;
  if (lit_opt) {
    while (search_pos <= t.len) {
      const int lp = bt_find_literal(c, t, search_pos);
      if (lp < 0) return false;
      int try_pos = lp;
      if (c.p.bt_lit_len > 0 && !prefix_lit) try_pos = lp - c.p.bt_pattern_len > 0 ? lp - c.p.bt_pattern_len : 0;
      while (try_pos <= lp) {
        const int end = bt_match_at<STACK>(c, t, try_pos, caps, false, -1);
        if (end >= 0 && bt_contains_literal(c, t, try_pos, end)) { ms = try_pos; me = end; return true; }
        ++try_pos;
      }
      search_pos = lp + 1;
    }
    return false;
  }
  while (search_pos <= t.len) {
    const int end = bt_match_at<STACK>(c, t, search_pos, caps, false, -1);
    if (end >= 0) { ms = search_pos; me = end; return true; }
    ++search_pos;
  }
  return false;
}
// NFAEngine.match_all, nfa.mojo:169-340
template <bool STACK, class Emit>
__device__ inline void bt_engine_match_all(const Ctx& c, const Text& t, Emit&& emit) {
  const bool lit_opt = (c.p.bt_flags & 1) != 0, prefix_lit = (c.p.bt_flags & 2) != 0;
  int current_pos = 0;
  if ((c.p.bt_flags & 4) && lit_opt && !bt_has_newline(t)) {
    const int last = bt_rfind_literal(c, t);
    if (last >= 0) emit(current_pos, last + c.p.bt_lit_len);
    return;
  }
  if ((c.p.bt_flags & 8) && lit_opt && prefix_lit && !bt_has_newline(t)) {
    if (current_pos < t.len) {
      const int pos = bt_find_literal(c, t, current_pos);
      if (pos >= 0) emit(pos, t.len);
    }
    return;
  }
  BtCaps caps;
  if (lit_opt) {
    while (current_pos <= t.len) {
      const int lp = bt_find_literal(c, t, current_pos);
      if (lp < 0) break;
      int try_pos = lp;
      if (c.p.bt_lit_len > 0 && !prefix_lit) try_pos = lp - 10 > current_pos ? lp - 10 : current_pos;   // search_window = 10
      bool found = false;
      const int max_positions = lp - try_pos + 1 < 5 ? lp - try_pos + 1 : 5;
      int tried = 0;
      while (try_pos <= lp && try_pos <= t.len && tried < max_positions) {
        const int end = bt_match_at<STACK>(c, t, try_pos, caps, false, -1);
        if (end >= 0 && bt_contains_literal(c, t, try_pos, end)) {
          emit(try_pos, end);
          current_pos = end == try_pos ? try_pos + 1 : end;
          found = true;
          break;
        }
        ++try_pos;
        ++tried;
      }
      if (!found) current_pos = lp + 1;
    }
    return;
  }
  while (current_pos <= t.len) {
    const int end = bt_match_at<STACK>(c, t, current_pos, caps, false, -1);
    if (end >= 0) {
      emit(current_pos, end);
      current_pos = end == current_pos ? current_pos + 1 : end;
    } else {
      ++current_pos;
    }
  }
}

// table walk from `start`; returns the last accepting position or -1
__device__ inline int walk(const Ctx& c, const Text& t, int start) {
  int state = 0;
  int pos = start;
  int last = flag(c, PF_START_ACCEPTING) ? pos : -1;
  const int ncls = c.p.ncls;
  while (pos < t.len) {
    const uint32_t e = c.trans[state * ncls + c.cls[t.at(pos)]];
    if (e == 0xFFFFu) break;
    state = e & 0x7FFF;
    ++pos;
    if (e & 0x8000u) last = pos;
  }
  return last;
}

// LazyDFA._run_lazy (pikevm.mojo:819-867) of a '$' program with the cache a freshly compiled pattern starts the
// text's call with.  _compute_transition closes the target with '$' satisfied iff the byte it consumes is the last
// of the text (pos + 1 == text_len, pikevm.mojo:869-942) and the result is cached for every later use of the same
// (state, byte) -- within the call: the other walks of a findall, the match_next calls of a sub.  Only the value of
// the text's last byte can ever be looked at "at the end", so the cache's memory is two state masks: pairs
// (state, that byte) first computed inside the text take the plain row for good, pairs first computed on the last
// byte take row nstates/2 + state for good -- including later uses INSIDE the text (a restarted search walks the
// text again): upstream does exactly that.
__device__ inline int walk_lazy_end(const Ctx& c, const Text& t, int start) {
  int state = 0;
  int pos = start;
  int last = flag(c, PF_START_ACCEPTING) ? pos : -1;
  const int ncls = c.p.ncls, ns = c.p.nstates >> 1;
  const int last_byte = t.len > 0 ? t.at(t.len - 1) : -1;
  while (pos < t.len) {
    const int b = t.at(pos);
    int row = state;
    if (b == last_byte) {
      const uint64_t bit = 1ull << state;
      if (t.lz_end & bit) row = ns + state;
      else if (!(t.lz_mid & bit)) {
        if (pos + 1 == t.len) { t.lz_end |= bit; row = ns + state; }
        else t.lz_mid |= bit;
      }
    }
    const uint32_t e = c.trans[row * ncls + c.cls[b]];
    if (e == 0xFFFFu) break;
    state = e & 0x7FFF;
    ++pos;
    if (e & 0x8000u) last = pos;
  }
  return last;
}

__device__ inline int count_consecutive(const Ctx& c, const Text& t, int start) {
  int pos = start;
  while (pos < t.len && c.first[t.at(pos)]) ++pos;
  return pos - start;
}

__device__ inline int find_first_class(const Ctx& c, const Text& t, int start) {
  for (int pos = start; pos < t.len; ++pos)
    if (c.first[t.at(pos)]) return pos;
  return -1;
}

// DFAEngine._try_match_at_position.  On success sets ms/me and returns true.
__device__ inline bool try_match_at(const Ctx& c, const Text& t, int start_pos, bool exact,
                                    int& ms, int& me) {
  if (start_pos > t.len) return false;
  if (flag(c, PF_PURE_LITERAL)) {
    const int plen = c.p.lit_len;
    if (exact) {
      if (start_pos + plen > t.len) return false;  // verify_match, simd_ops.mojo:937-960
      for (int k = 0; k < plen; ++k)
        if (t.at(start_pos + k) != c.lit[k]) return false;
      ms = start_pos; me = start_pos + plen;
      return true;
    }
    const int pos = find_literal(c.lit, plen, t, start_pos);
    if (pos < 0) return false;
    ms = pos; me = pos + plen;
    return true;
  }
  if (flag(c, PF_HAS_MATCHER) && (flag(c, PF_SCAN_ELIGIBLE) || flag(c, PF_START_ACCEPTING))) {
    // _try_match_simd: the first class's run, or an empty match if the start accepts
    const int n = count_consecutive(c, t, start_pos);
    bool valid = false;
    int end = start_pos + n;
    if (n == 0) { if (flag(c, PF_START_ACCEPTING)) { valid = true; end = start_pos; } }
    else valid = true;
    if (valid && !(flag(c, PF_END_ANCHOR) && end != t.len)) {
      ms = start_pos; me = end;
      return true;
    }
  }
  if (start_pos == t.len) {
    if (flag(c, PF_START_ACCEPTING)) { ms = me = start_pos; return true; }
    return false;
  }
  const int last = walk(c, t, start_pos);
  if (last < 0) return false;
  if (flag(c, PF_END_ANCHOR) && last != t.len) return false;
  ms = start_pos; me = last;
  return true;
}

// The same walk on the bitset NFA: the state is the set of live PikeVM positions
// (pikevm.mojo:497-602 thread list, as a bit mask) instead of the id of its determinised
// LazyDFA state; transitions are computed, not looked up:
//     next = OR_{i in set & mask[byte]} follow[i],  dead iff next is empty.
// One lane still owns one text; the per-byte cost grows with the number of live positions.
template <int NW>
__device__ __noinline__ int walk_bitset(const Ctx& c, const Text& t, int start) {
  uint64_t S[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) S[w] = c.p.bs_start[w];
  int pos = start;
  int last = flag(c, PF_START_ACCEPTING) ? pos : -1;
  while (pos < t.len) {
    const uint64_t* m = c.bs_mask + (int)c.bs_cls[t.at(pos)] * NW;
    uint64_t N[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) N[w] = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      uint64_t hit = S[w] & m[w];
      while (hit) {
        const int i = w * 64 + __builtin_ctzll(hit);
        hit &= hit - 1;
        const uint64_t* f = c.bs_follow + i * NW;
#pragma unroll
        for (int v = 0; v < NW; ++v) N[v] |= f[v];
      }
    }
    uint64_t any = 0, acc = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { any |= N[w]; acc |= N[w] & c.p.bs_match[w]; S[w] = N[w]; }
    if (!any) break;  // LAZY_DFA_DEAD
    ++pos;
    if (acc) last = pos;
  }
  return last;
}

__device__ inline int lazy_walk(const Ctx& c, const Text& t, int start) {
  if (flag(c, PF_LAZY_END)) return walk_lazy_end(c, t, start);
  if (!flag(c, PF_BITSET)) return walk(c, t, start);
  if (c.p.bs_nw == 1) return walk_bitset<1>(c, t, start);
  if (c.p.bs_nw == 2) return walk_bitset<2>(c, t, start);
  return walk_bitset<4>(c, t, start);
}

// LazyDFA._run_lazy
__device__ inline bool lazy_run(const Ctx& c, const Text& t, int start, int& ms, int& me) {
  if (flag(c, PF_START_DEAD)) return false;
  if (start > t.len) {  // no byte is read; the start set alone decides (pikevm.mojo:861-866)
    if (!flag(c, PF_START_ACCEPTING)) return false;
    ms = me = start;
    return true;
  }
  const int last = lazy_walk(c, t, start);
  if (last < 0) return false;
  ms = start; me = last;
  return true;
}

// BT (here and in the callers below): the kernel instantiation that carries the backtracking matcher's
// interpreter.  Plans without a backtracker route run the BT = false instantiations, which are a third of the
// registers (80 against 256 VGPRs and 3.6 KB of scratch: with the interpreter inlined the compiler keeps one
// wavefront per SIMD resident, whatever the plan at hand needs).
template <int BT>
__device__ inline bool engine_match_first(const Ctx& c, const Text& t, int start, int& ms, int& me) {
  if constexpr (BT)
    if (flag(c, PF_BT_FIRST)) return bt_engine_match_first<BT == 1>(c, t, start, ms, me);   // NFAMatcher -> NFAEngine, matcher.mojo:380
  if (c.p.kind == PLAN_LAZY) return lazy_run(c, t, start, ms, me);
  if (flag(c, PF_START_ANCHOR) && start > 0) return false;   // dfa.mojo:1866-1867
  return try_match_at(c, t, start, true, ms, me);
}

template <int BT>
__device__ inline bool engine_match_next(const Ctx& c, const Text& t, int start, int& ms, int& me) {
  if constexpr (BT)
    if (flag(c, PF_BT_SEARCH)) return bt_engine_match_next<BT == 1>(c, t, start, ms, me);   // matcher.mojo:419
  if (c.p.kind == PLAN_LAZY) {
    if (flag(c, PF_HAS_MATCHER)) {  // first-byte filter
      int pos = start;
      while (pos < t.len) {
        const int cand = find_first_class(c, t, pos);
        if (cand < 0) break;
        if (lazy_run(c, t, cand, ms, me)) return true;
        pos = cand + 1;
      }
      return lazy_run(c, t, t.len, ms, me);
    }
    for (int p = start; p <= t.len; ++p)
      if (lazy_run(c, t, p, ms, me)) return true;
    return false;
  }
  if (flag(c, PF_START_ANCHOR)) {
    if (start == 0) return try_match_at(c, t, 0, false, ms, me);
    return false;
  }
  if (flag(c, PF_HAS_MATCHER) && !flag(c, PF_END_ANCHOR)) {
    // _optimized_simd_search
    int pos = start;
    if (flag(c, PF_SCAN_ELIGIBLE)) {
      while (pos < t.len) {
        const int mp = find_first_class(c, t, pos);
        if (mp < 0) return false;
        const int ml = count_consecutive(c, t, mp);
        if (ml > 0) { ms = mp; me = mp + ml; return true; }
        pos = mp + 1;
      }
      return false;
    }
    while (pos < t.len) {
      const int fp = find_first_class(c, t, pos);
      if (fp < 0) return false;
      if (try_match_at(c, t, fp, false, ms, me)) return true;
      pos = fp + 1;
    }
    return false;
  }
  if (flag(c, PF_PURE_LITERAL)) {
    // every try_pos runs simd_search from try_pos: the first hit is the answer
    if (start > t.len) return false;
    return try_match_at(c, t, start, false, ms, me);
  }
  for (int p = start; p <= t.len; ++p)
    if (try_match_at(c, t, p, false, ms, me)) return true;
  return false;
}

// HybridMatcher.match_first, matcher.mojo:733-753
template <int BT>
__device__ inline bool hybrid_match_first(const Ctx& c, const Text& t, int start, int& ms, int& me) {
  if (c.p.kind == PLAN_ANY) {
    if (start <= t.len) { ms = start; me = t.len; return true; }
    return false;
  }
  return engine_match_first<BT>(c, t, start, ms, me);
}

// HybridMatcher.match_next, matcher.mojo:755-802
template <int BT>
__device__ inline bool hybrid_match_next(const Ctx& c, const Text& t, int start, int& ms, int& me) {
  if (c.p.kind == PLAN_ANY) {
    if (start <= t.len) { ms = start; me = t.len; return true; }
    return false;
  }
  if (flag(c, PF_EXACT_LITERAL)) {
    if (start >= t.len) return false;
    const int pos = find_literal(c.lit, c.p.lit_len, t, start);
    if (pos < 0 || pos + c.p.lit_len > t.len) return false;
    ms = pos; me = pos + c.p.lit_len;
    return true;
  }
  if (flag(c, PF_PREFILTER)) {
    if (start >= t.len) return false;
    // (a batch-wide pass only runs for this plan when its literal is the prefilter's: bt_prepass())
    int cand;
    if (t.pre_first != kPreUnknown && t.pre_first < 0) return false;
    if (t.pre_first != kPreUnknown && start <= t.pre_first) cand = t.pre_first;
    else cand = find_literal(c.pre, c.p.pre_len, t, start);
    if (cand < 0) return false;
    return engine_match_next<BT>(c, t, cand, ms, me);
  }
  return engine_match_next<BT>(c, t, start, ms, me);
}

// DFAEngine.is_match through HybridMatcher.is_match, matcher.mojo:721-731, dfa.mojo:1815-1849
template <int BT>
__device__ inline bool hybrid_is_match(const Ctx& c, const Text& t, int start) {
  int ms, me;
  if (c.p.kind == PLAN_ANY) return start <= t.len;
  if constexpr (BT)
    if (flag(c, PF_BT_FIRST)) return bt_engine_match_first<BT == 1>(c, t, start, ms, me);
  if (c.p.kind == PLAN_LAZY) return lazy_run(c, t, start, ms, me);
  if (flag(c, PF_START_ANCHOR) && start > 0) return false;
  if (flag(c, PF_HAS_MATCHER) && c.p.nstates > 0) {
    if (start >= t.len) return flag(c, PF_START_ACCEPTING);
    if (c.first[t.at(start)]) return true;
    return flag(c, PF_START_ACCEPTING);
  }
  return try_match_at(c, t, start, true, ms, me);
}

// HybridMatcher.match_all: calls emit(start, end) for every match, in order.
template <int BT, class Emit>
__device__ inline void for_each_match(const Ctx& c, const Text& t, Emit&& emit) {
  int ms, me;
  if (c.p.kind == PLAN_ANY) { emit(0, t.len); return; }
  if (flag(c, PF_EXACT_LITERAL)) {
    // matcher.mojo:815-847: start = pos + 1, i.e. overlapping occurrences
    const int ll = c.p.lit_len;
    if (ll > t.len) return;
    const int max_start = t.len - ll;
    int start = 0;
    while (start <= max_start) {
      const int pos = find_literal(c.lit, ll, t, start);
      if (pos < 0) break;
      emit(pos, pos + ll);
      start = pos + 1;
    }
    return;
  }
  if (c.p.required_byte >= 0) {
    // _match_all_required_byte, matcher.mojo:864-898
    int pos = 0;
    while (pos < t.len) {
      int hit = -1;
      for (int k = pos; k < t.len; ++k)
        if (t.at(k) == c.p.required_byte) { hit = k; break; }
      if (hit < 0) break;
      int start = hit;
      while (start > 0 && c.first[t.at(start - 1)]) --start;
      if (engine_match_first<BT>(c, t, start, ms, me) && me > hit) {
        emit(ms, me);
        pos = me;
        if (pos <= hit) pos = hit + 1;
      } else {
        pos = hit + 1;
      }
    }
    return;
  }
  if constexpr (BT)
    if (flag(c, PF_BT_SEARCH)) { bt_engine_match_all<BT == 1>(c, t, emit); return; }   // matcher.mojo:431
  if (c.p.kind == PLAN_LAZY) {
    int pos = 0;
    if (flag(c, PF_HAS_MATCHER)) {
      while (pos < t.len) {
        const int cand = find_first_class(c, t, pos);
        if (cand < 0) break;
        pos = cand;
        if (lazy_run(c, t, pos, ms, me)) { emit(ms, me); pos = (pos + 1 > me) ? pos + 1 : me; }
        else ++pos;
      }
      return;
    }
    while (pos <= t.len) {
      if (lazy_run(c, t, pos, ms, me)) { emit(ms, me); pos = (pos + 1 > me) ? pos + 1 : me; }
      else ++pos;
    }
    return;
  }
  // DFAEngine.match_all
  if (flag(c, PF_START_ANCHOR) || flag(c, PF_END_ANCHOR)) {
    if (engine_match_next<BT>(c, t, 0, ms, me)) emit(ms, me);
    return;
  }
  int pos = 0;
  if (flag(c, PF_PURE_LITERAL)) {
    const int plen = c.p.lit_len;
    while (pos <= t.len - plen) {
      const int hit = find_literal(c.lit, plen, t, pos);
      if (hit < 0) break;
      emit(hit, hit + plen);
      pos = hit + plen;
    }
    return;
  }
  if (flag(c, PF_HAS_MATCHER) && c.p.nstates > 0) {
    if (flag(c, PF_SCAN_ELIGIBLE)) {
      while (pos < t.len) {
        const int mp = find_first_class(c, t, pos);
        if (mp < 0) break;
        const int ml = count_consecutive(c, t, mp);
        if (ml > 0) { emit(mp, mp + ml); pos = mp + ml; }
        else pos = mp + 1;
      }
      return;
    }
    while (pos < t.len) {
      const int np = find_first_class(c, t, pos);
      if (np < 0) break;
      pos = np;
      if (try_match_at(c, t, pos, false, ms, me)) {
        emit(ms, me);
        pos = (me == ms) ? pos + 1 : me;
      } else {
        ++pos;
      }
    }
    return;
  }
  while (pos <= t.len) {
    if (try_match_at(c, t, pos, false, ms, me)) {
      emit(ms, me);
      pos = (me == ms) ? pos + 1 : me;
    } else {
      ++pos;
    }
  }
}

// _sub_impl_with_repl.  `out` is a sink with bytes(ptr, n); `tpl` the parsed
// replacement template (only read when use_groups).
// use_groups: 0 = literal replacement, 1 = fixed-width group form, 2 = general groups
// (NFAEngine.match_next_with_groups, matcher.mojo:1781-1822)
template <int BT, class Sink>
__device__ inline void sub_text(const Ctx& c, const Text& t, const uint8_t* repl, int repl_len,
                                int use_groups, const ReplSeg* tpl, int ntpl, long long count,
                                Sink& out) {
  if (t.len == 0) return;
  if constexpr (BT) if (use_groups == 2) {
    int pos = 0, ms, me;
    long long reps = 0;
    BtCaps caps;
    while (pos <= t.len) {
      if (!bt_match_next_with_groups<BT == 1>(c, t, pos, ms, me, caps)) break;
      // NFAEngine's literal prefilter backs up to literal_pos - len(pattern) without looking at `start`
      // (nfa.mojo:531-533): it can hand back a match that lies in front of `pos`.  Upstream the loop of
      // CompiledRegex.sub then never advances; here the replacing stops (no result to agree with).
      if ((me == ms ? me + 1 : me) <= pos) break;
      if (ms > pos) out.bytes(t.ptr + pos, ms - pos);
      for (int k = 0; k < ntpl; ++k) {   // _apply_template_groups, matcher.mojo:1624-1646
        const ReplSeg sg = tpl[k];
        if (sg.group_ref > 0) {
          const int cs = sg.group_ref <= 9 ? caps.gs(sg.group_ref) : -1;
          // a repeated group at the end of the text can carry a span that ends one byte behind it ('^(\\d)+' on
          // "..1": _match_group_with_quantifier's span arithmetic); Match.get_match_text (matching.mojo:39-46)
          // reads that byte unchecked upstream.  Here, as in the oracle, the group's text ends with the text.
          const int ce = caps.ge(sg.group_ref) < t.len ? caps.ge(sg.group_ref) : t.len;
          if (cs >= 0 && ce > cs) out.bytes(t.ptr + cs, ce - cs);
        } else {
          out.bytes(repl + sg.start, sg.length);
        }
      }
      ++reps;
      if (me == ms) {
        if (pos < t.len) out.bytes(t.ptr + pos, 1);
        pos = me + 1;
      } else {
        pos = me;
      }
      if (count > 0 && reps >= count) break;
    }
    if (pos < t.len) out.bytes(t.ptr + pos, t.len - pos);
    return;
  }
  auto apply_tpl = [&](int match_start) {  // _apply_template_fixed, matcher.mojo:1592-1621
    for (int k = 0; k < ntpl; ++k) {
      const ReplSeg s = tpl[k];
      if (s.group_ref > 0 && s.group_ref <= c.p.fixed_ngroups) {
        // upstream reads the window unchecked; here (and in the oracle) it ends with the text -- see k_subs_reach
        const int gs = match_start + c.p.fixed_off[s.group_ref];
        const int gw = gs + c.p.fixed_w[s.group_ref] <= t.len ? c.p.fixed_w[s.group_ref] : t.len - gs;
        if (gw > 0) out.bytes(t.ptr + gs, gw);
      } else
        out.bytes(repl + s.start, s.length);
    }
  };
  if (use_groups && c.p.fixed_concat && t.len == c.p.fixed_total) {
    // matcher.mojo:1726-1744: whole-text fast path, no engine call
    bool digits = true;
    for (int i = 0; i < c.p.fixed_total; ++i) {
      const int b = t.at(i);
      if (b < '0' || b > '9') { digits = false; break; }
    }
    if (digits) apply_tpl(0);
    else out.bytes(t.ptr, t.len);
    return;
  }
  int pos = 0;
  long long reps = 0;
  int ms, me;
  while (pos <= t.len) {
    if (!hybrid_match_next<BT>(c, t, pos, ms, me)) break;
    if ((me == ms ? me + 1 : me) <= pos) break;   // (as above: a match in front of pos; upstream does not terminate)
    if (ms > pos) out.bytes(t.ptr + pos, ms - pos);
    if (use_groups) apply_tpl(ms);
    else out.bytes(repl, repl_len);
    ++reps;
    if (me == ms) {
      if (pos < t.len) out.bytes(t.ptr + pos, 1);  // sic: byte at pos, not at ms (matcher.mojo:1772-1776)
      pos = me + 1;
    } else {
      pos = me;
    }
    if (count > 0 && reps >= count) break;
  }
  if (pos < t.len) out.bytes(t.ptr + pos, t.len - pos);
}

}  // namespace mrx
