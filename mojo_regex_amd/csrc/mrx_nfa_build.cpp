// PikeVM bytecode (src/regex/pikevm.mojo:124-333), first-byte filter
// (:367-416) and eager determinisation of the LazyDFA (:664-987).
#include <map>

#include "mrx_engines.hpp"

namespace mrx {
namespace {

struct Emitter {
  const Ast& a;
  Program& p;

  int emit(Op op, int a0 = 0, int a1 = 0) {
    p.insts.push_back({op, a0, a1});
    return (int)p.insts.size() - 1;
  }
  int len() const { return (int)p.insts.size(); }

  int class_index(const std::array<uint8_t, 256>& t) {  // pikevm.mojo:101-109
    for (size_t i = 0; i < p.classes.size(); ++i)
      if (p.classes[i] == t) return (int)i;
    p.classes.push_back(t);
    return (int)p.classes.size() - 1;
  }

  void node(const Node& n) {  // pikevm.mojo:142-164
    switch (n.type) {
      case N_GROUP:
        for (int i = 0; i < a.nkids(n); ++i) quantified(a.child(n, i));
        break;
      case N_OR: alt(n); break;
      case N_ELEMENT:
        if (a.has_value(n)) emit(OP_BYTE, (unsigned char)a.value(n)[0]);
        break;
      case N_DIGIT: case N_WORD: case N_SPACE: case N_RANGE: char_class(n); break;
      case N_WILDCARD: emit(OP_ANY); break;
      case N_START: emit(OP_START_ANCHOR); break;
      case N_END: emit(OP_END_ANCHOR); break;
      default: break;
    }
  }

  void quantified(const Node& n) {  // pikevm.mojo:174-229
    const int mn = n.min, mx = n.max;
    if (mn == 1 && mx == 1) { node(n); return; }
    if (mn == 0 && mx == 1) {
      const int sp = emit(OP_SPLIT);
      const int body = len();
      node(n);
      p.insts[sp].a0 = body; p.insts[sp].a1 = len();
      return;
    }
    if (mn == mx && mn > 1) { for (int i = 0; i < mn; ++i) node(n); return; }
    if (mx > 0) {
      for (int i = 0; i < mn; ++i) node(n);
      std::vector<int> splits;
      for (int i = 0; i < mx - mn; ++i) { splits.push_back(emit(OP_SPLIT)); node(n); }
      const int after = len();
      for (int s : splits) { p.insts[s].a0 = s + 1; p.insts[s].a1 = after; }
      return;
    }
    if (mx == -1) {
      for (int i = 0; i < mn; ++i) node(n);
      const int sp = emit(OP_SPLIT);
      const int body = len();
      node(n);
      emit(OP_JUMP, sp);
      p.insts[sp].a0 = body; p.insts[sp].a1 = len();
      return;
    }
  }

  void alt(const Node& n) {  // pikevm.mojo:232-254
    const int k = a.nkids(n);
    if (k == 0) return;
    if (k == 1) { node(a.child(n, 0)); return; }
    const int sp = emit(OP_SPLIT);
    const int l = len();
    node(a.child(n, 0));
    const int jp = emit(OP_JUMP);
    const int r = len();
    node(a.child(n, 1));
    p.insts[sp].a0 = l; p.insts[sp].a1 = r;
    p.insts[jp].a0 = len();
  }

  void char_class(const Node& n) {  // pikevm.mojo:271-333
    std::array<uint8_t, 256> t{};
    auto range = [&](int lo, int hi) { for (int c = lo; c <= hi && c < 256; ++c) t[c] = 1; };
    if (n.type == N_DIGIT) range('0', '9');
    else if (n.type == N_WORD) { range('a', 'z'); range('A', 'Z'); range('0', '9'); t['_'] = 1; }
    else if (n.type == N_SPACE) { for (unsigned char c : std::string(" \t\n\r\f")) t[c] = 1; }
    else if (n.type == N_RANGE && a.has_value(n)) {
      std::string_view in = a.value(n);
      if (in.size() >= 2 && in.front() == '[' && in.back() == ']') in = in.substr(1, in.size() - 2);
      const size_t m = in.size();
      for (size_t j = 0; j < m;) {
        const int c0 = (unsigned char)in[j];
        if (j + 1 < m && c0 == '\\') {
          const int nc = (unsigned char)in[j + 1];
          if (nc == 's') { for (unsigned char c : std::string(" \t\n\r\f")) t[c] = 1; }
          else if (nc == 'd') range('0', '9');
          else if (nc == 'w') { range('a', 'z'); range('A', 'Z'); range('0', '9'); t['_'] = 1; }
          else t[nc] = 1;
          j += 2;
        } else if (j + 2 < m && in[j + 1] == '-') {
          range(c0, (unsigned char)in[j + 2]);
          j += 3;
        } else {
          t[c0] = 1;
          ++j;
        }
      }
    }
    if (!n.positive)
      for (auto& x : t) x = 1 - x;
    emit(OP_CLASS, class_index(t));
  }
};

struct Closure {
  const Program& p;
  // _add_state, pikevm.mojo:604-648.  at_text_start / at_text_end decide ^ / $.
  void add(std::vector<uint8_t>& seen, int& leaves, int pc, bool at_start, bool at_end) const {
    if (pc >= (int)p.insts.size() || seen[pc]) return;
    const Inst& in = p.insts[pc];
    switch (in.op) {
      case OP_SPLIT:
        seen[pc] = 1;
        add(seen, leaves, in.a0, at_start, at_end);
        add(seen, leaves, in.a1, at_start, at_end);
        break;
      case OP_JUMP:
        seen[pc] = 1;
        add(seen, leaves, in.a0, at_start, at_end);
        break;
      case OP_START_ANCHOR:
        if (at_start) { seen[pc] = 1; add(seen, leaves, pc + 1, at_start, at_end); }
        break;
      case OP_END_ANCHOR:
        if (at_end) { seen[pc] = 1; add(seen, leaves, pc + 1, at_start, at_end); }
        break;
      default:
        seen[pc] = 1;
        ++leaves;
        break;
    }
  }
  bool has_match(const std::vector<uint8_t>& set) const {
    for (size_t pc = 0; pc < p.insts.size(); ++pc)
      if (set[pc] && p.insts[pc].op == OP_MATCH) return true;
    return false;
  }
  bool steps(int pc, int ch) const {
    const Inst& in = p.insts[pc];
    switch (in.op) {
      case OP_BYTE: return ch == in.a0;
      case OP_CLASS: return p.classes[in.a0][ch] != 0;
      case OP_ANY: return ch != 10;
      case OP_RANGE: return ch >= in.a0 && ch <= in.a1;
      default: return false;
    }
  }
};

}  // namespace

void compile_program(const Ast& a, Program& p) {  // pikevm.mojo:124-139
  p = Program();
  Emitter e{a, p};
  if (a.root.type == N_RE && a.nkids(a.root) > 0) e.node(a.child(a.root, 0));
  e.emit(OP_MATCH);
}

void build_lazy(const Program& p, LazyTables& out, int max_dfa_states) {
  out = LazyTables();
  const int n = (int)p.insts.size();
  out.supported = n <= kPikeMaxStates;  // PikeVMEngine.is_supported, pikevm.mojo:363-365
  if (!out.supported || n == 0) return;

  // first-byte filter, pikevm.mojo:367-416 (matching_bytes counts duplicates)
  {
    std::vector<uint8_t> seen(n, 0);
    std::vector<int> st{0};
    int matching = 0;
    bool abandoned = false;
    while (!st.empty() && !abandoned) {
      const int pc = st.back(); st.pop_back();
      if (pc >= n || seen[pc]) continue;
      seen[pc] = 1;
      const Inst& in = p.insts[pc];
      switch (in.op) {
        case OP_BYTE: out.first_byte[in.a0] = 1; ++matching; break;
        case OP_CLASS:
          for (int c = 0; c < 256; ++c)
            if (p.classes[in.a0][c]) { out.first_byte[c] = 1; ++matching; }
          break;
        case OP_RANGE:
          for (int c = in.a0; c <= in.a1; ++c) { out.first_byte[c] = 1; ++matching; }
          break;
        case OP_ANY: case OP_MATCH: abandoned = true; break;
        case OP_SPLIT: st.push_back(in.a0); st.push_back(in.a1); break;
        case OP_JUMP: st.push_back(in.a0); break;
        case OP_START_ANCHOR: case OP_END_ANCHOR: st.push_back(pc + 1); break;
      }
    }
    out.has_filter = !abandoned && matching < 128;
    if (out.has_filter) {
      int bucket = 0;
      for (int c = 0; c < 256; ++c)
        if (out.first_byte[c]) {
          const uint8_t bit = (uint8_t)(1u << (bucket & 7));
          out.lo_tbl[c & 15] |= bit; out.hi_tbl[(c >> 4) & 15] |= bit;
          ++bucket;
        }
    }
  }

  // Eager version of LazyDFA._get_or_create_state_for_pos / _compute_transition
  // (pikevm.mojo:869-978).  The start closure is built as upstream with pos = 0
  // and text_len = 0, so '^' (and '$') are satisfied there at every start
  // position; afterwards pos+1 is never 0, and '$' programs are refused by the
  // router, so closures after a byte are position independent.
  Closure cl{p};
  std::map<std::vector<uint8_t>, int> ids;
  std::vector<std::vector<uint8_t>> sets;
  {
    std::vector<uint8_t> seen(n, 0);
    int leaves = 0;
    cl.add(seen, leaves, 0, /*at_start=*/true, /*at_end=*/true);
    if (leaves == 0 && !cl.has_match(seen)) { out.start_dead = true; return; }
    ids[seen] = 0;
    sets.push_back(seen);
  }
  // '$' programs: a transition computed while the LAST byte of the text is consumed closes with '$' satisfied
  // (_compute_transition passes pos + 1 and text_len to _add_state, pikevm.mojo:869-942) and is cached like any
  // other, so every (state, byte) has two possible targets; which one a text sees depends on where the pair first
  // occurs in it (mrx_device.hpp: walk_lazy_end).  trans_end holds the "at the end" variant; the states reachable
  // through either variant are explored with both.
  const bool two = p.has_end_anchor();
  out.has_end_variant = two;
  auto new_row = [&]() {
    out.trans.emplace_back();
    out.trans.back().fill(-1);
    if (two) { out.trans_end.emplace_back(); out.trans_end.back().fill(-1); }
  };
  for (size_t s = 0; s < sets.size(); ++s) {
    new_row();
    out.is_match.push_back(cl.has_match(sets[s]) ? 1 : 0);
  }
  for (size_t s = 0; s < sets.size(); ++s) {
    for (int variant = 0; variant < (two ? 2 : 1); ++variant)
    for (int ch = 0; ch < 256; ++ch) {
      std::vector<uint8_t> nxt(n, 0);
      int leaves = 0;
      const std::vector<uint8_t> cur = sets[s];
      for (int pc = 0; pc < n; ++pc)
        if (cur[pc] && cl.steps(pc, ch)) cl.add(nxt, leaves, pc + 1, false, variant == 1);
      if (leaves == 0) continue;  // LAZY_DFA_DEAD
      auto it = ids.find(nxt);
      int id;
      if (it == ids.end()) {
        id = (int)sets.size();
        if (id >= max_dfa_states) { out.too_large = true; return; }
        ids[nxt] = id;
        sets.push_back(nxt);
        new_row();
        out.is_match.push_back(cl.has_match(nxt) ? 1 : 0);
      } else {
        id = it->second;
      }
      (variant ? out.trans_end : out.trans)[s][ch] = id;
    }
  }
}

namespace {
// _epsilon_close, onepass.mojo:64-110: every visited pc is marked; byte ops and MATCH are kept
std::vector<uint8_t> onepass_close(const Program& p, const std::vector<int>& start, bool at_start,
                                   bool at_end) {
  const int n = (int)p.insts.size();
  std::vector<uint8_t> r(n, 0);
  std::vector<int> st(start);
  while (!st.empty()) {
    const int pc = st.back(); st.pop_back();
    if (pc < 0 || pc >= n || r[pc]) continue;
    r[pc] = 1;
    const Inst& in = p.insts[pc];
    if (in.op == OP_SPLIT) { st.push_back(in.a0); st.push_back(in.a1); }
    else if (in.op == OP_JUMP) st.push_back(in.a0);
    else if (in.op == OP_START_ANCHOR) { if (at_start) st.push_back(pc + 1); }
    else if (in.op == OP_END_ANCHOR) { if (at_end) st.push_back(pc + 1); }
  }
  return r;
}
}  // namespace

void build_onepass(const Program& p, OnePassTables& out) {  // compile_onepass, onepass.mojo:180-355
  out = OnePassTables();
  const int n = (int)p.insts.size();
  if (n == 0 || n > kPikeMaxStates) return;
  for (const Inst& in : p.insts) {
    if (in.op == OP_START_ANCHOR) out.has_start_anchor = true;
    if (in.op == OP_END_ANCHOR) out.has_end_anchor = true;
  }
  Closure cl{p};
  std::vector<std::vector<uint8_t>> sets;
  std::map<std::vector<uint8_t>, int> index;
  sets.push_back(onepass_close(p, {0}, true, false));
  index[sets[0]] = 0;
  out.trans.emplace_back(); out.trans.back().fill(-1);
  std::vector<int> work{0};
  constexpr int kMax = 512;  // ONEPASS_MAX_STATES
  while (!work.empty()) {
    const int sid = work.back(); work.pop_back();
    const std::vector<uint8_t> cur = sets[sid];
    std::vector<int> active;
    for (int pc = 0; pc < n; ++pc) {
      const Op op = p.insts[pc].op;
      if (cur[pc] && (op == OP_BYTE || op == OP_CLASS || op == OP_ANY || op == OP_RANGE)) active.push_back(pc);
    }
    for (int b = 0; b < 256; ++b) {
      std::vector<int> nxt;
      for (int pc : active)
        if (cl.steps(pc, b)) nxt.push_back(pc + 1);
      if (nxt.empty()) continue;
      const std::vector<uint8_t> first = onepass_close(p, {nxt[0]}, false, false);
      for (size_t i = 1; i < nxt.size(); ++i)
        if (onepass_close(p, {nxt[i]}, false, false) != first) { out = OnePassTables(); return; }
      auto it = index.find(first);
      int idx;
      if (it != index.end()) idx = it->second;
      else {
        if ((int)sets.size() >= kMax) { out = OnePassTables(); return; }
        idx = (int)sets.size();
        sets.push_back(first);
        index[first] = idx;
        out.trans.emplace_back(); out.trans.back().fill(-1);
        work.push_back(idx);
      }
      out.trans[sid][b] = (int16_t)idx;
    }
  }
  for (const auto& s : sets) {
    out.is_match.push_back(cl.has_match(s) ? 1 : 0);
    uint8_t e = 0;
    if (out.has_end_anchor) {  // _closure_reaches_match_with_end_anchor, onepass.mojo:122-140
      std::vector<int> pcs;
      for (int pc = 0; pc < n; ++pc) if (s[pc]) pcs.push_back(pc);
      e = cl.has_match(onepass_close(p, pcs, false, true)) ? 1 : 0;
    }
    out.is_end_match.push_back(e);
  }
  out.ok = true;
}

void build_bitset(const Program& p, BitsetNfa& out) {
  out = BitsetNfa();
  const int n = (int)p.insts.size();
  if (n == 0 || n > kPikeMaxStates || p.has_end_anchor()) return;
  std::vector<int> pos_of(n, -1);
  for (int pc = 0; pc < n; ++pc) {
    const Op op = p.insts[pc].op;
    if (op == OP_BYTE || op == OP_CLASS || op == OP_ANY || op == OP_RANGE || op == OP_MATCH) {
      pos_of[pc] = (int)out.pos_pc.size();
      out.pos_pc.push_back(pc);
    }
  }
  out.npos = (int)out.pos_pc.size();
  if (out.npos == 0 || out.npos > 64 * kBitsetWords) return;
  out.nw = out.npos <= 64 ? 1 : out.npos <= 128 ? 2 : 4;
  Closure cl{p};
  auto to_bits = [&](const std::vector<uint8_t>& seen, std::array<uint64_t, kBitsetWords>& bits) {
    bits.fill(0);
    for (int pc = 0; pc < n; ++pc)
      if (seen[pc] && pos_of[pc] >= 0) bits[pos_of[pc] >> 6] |= 1ull << (pos_of[pc] & 63);
  };
  {
    // start closure as LazyDFA builds it: pos = 0, so '^' passes at every start position
    std::vector<uint8_t> seen(n, 0);
    int leaves = 0;
    cl.add(seen, leaves, 0, /*at_start=*/true, /*at_end=*/true);
    to_bits(seen, out.start);
  }
  out.follow.resize(out.npos);
  for (int i = 0; i < out.npos; ++i) {
    const int pc = out.pos_pc[i];
    if (p.insts[pc].op == OP_MATCH) {
      out.match[i >> 6] |= 1ull << (i & 63);
      out.follow[i].fill(0);
      continue;
    }
    std::vector<uint8_t> seen(n, 0);
    int leaves = 0;
    cl.add(seen, leaves, pc + 1, false, false);
    to_bits(seen, out.follow[i]);
    for (int ch = 0; ch < 256; ++ch)
      if (cl.steps(pc, ch)) out.byte_mask[ch][i >> 6] |= 1ull << (i & 63);
  }
  out.ok = true;
}

// ---- backtracking matcher -> flat program ---------------------------------------------------
namespace {
// ast.mojo:672-725
bool bt_has_range_seq(std::string_view p, int lo, int hi) {
  for (size_t i = 0; i + 2 < p.size(); ++i)
    if ((uint8_t)p[i] == lo && p[i + 1] == '-' && (uint8_t)p[i + 2] == hi) return true;
  return false;
}
enum { RK_LOWER = 1, RK_UPPER, RK_DIGITS, RK_ALNUM, RK_ALPHA, RK_COMPLEX_ALNUM, RK_OTHER };
int bt_range_kind(std::string_view p) {
  if (p.empty()) return RK_OTHER;
  if (p == "[a-z]") return RK_LOWER;
  if (p == "[A-Z]") return RK_UPPER;
  if (p == "[0-9]") return RK_DIGITS;
  if (p == "[a-zA-Z0-9]" || p == "[0-9a-zA-Z]") return RK_ALNUM;
  if (p == "[a-zA-Z]") return RK_ALPHA;
  if (p.front() == '[' && p.back() == ']' && p.size() - 2 > 10 && bt_has_range_seq(p, 'a', 'z') &&
      bt_has_range_seq(p, 'A', 'Z') && bt_has_range_seq(p, '0', '9'))
    return RK_COMPLEX_ALNUM;
  return RK_OTHER;
}
bool bt_lower(int c) { return c >= 'a' && c <= 'z'; }
bool bt_upper(int c) { return c >= 'A' && c <= 'Z'; }
bool bt_digit(int c) { return c >= '0' && c <= '9'; }
bool bt_word(int c) { return bt_lower(c) || bt_upper(c) || bt_digit(c) || c == '_'; }
bool bt_space5(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f'; }
// NibbleBasedMatcher.contains with the whitespace tables, simd_matchers.mojo:129-147, 285-342
bool bt_space_lut(int c) { return ((c & 15) == 0 || ((c & 15) >= 9 && (c & 15) <= 13)) && ((c >> 4) == 0 || (c >> 4) == 2); }
// ast.mojo:480-508
bool bt_code_matches_range(int ch, std::string_view syn) {
  size_t i = (!syn.empty() && syn[0] == '^') ? 1 : 0;
  while (i < syn.size()) {
    if (i + 2 < syn.size() && syn[i + 1] == '-') {
      if ((uint8_t)syn[i] <= ch && ch <= (uint8_t)syn[i + 2]) return true;
      i += 3;
    } else {
      if ((uint8_t)syn[i] == ch) return true;
      i += 1;
    }
  }
  return false;
}
bool bt_in_string(int ch, std::string_view s) { return s.find((char)ch) != std::string_view::npos; }
// ast.mojo:464-478
bool bt_in_range_by_code(int ch, std::string_view pat) {
  if (!pat.empty() && pat[0] == '[') return bt_code_matches_range(ch, pat.substr(1, pat.size() >= 2 ? pat.size() - 2 : 0));
  return bt_in_string(ch, pat);
}
// NFAEngine._match_char_in_range, nfa.mojo:1650-1670
bool bt_match_char_in_range(std::string_view pat, int ch) {
  if (pat.size() >= 2 && pat.front() == '[' && pat.back() == ']') {
    const std::string_view inner = pat.substr(1, pat.size() - 2);
    if (inner.size() == 3 && inner[1] == '-') return (uint8_t)inner[0] <= ch && ch <= (uint8_t)inner[2];
    return bt_in_string(ch, inner);
  }
  return bt_in_string(ch, pat);
}
bool bt_kind_member(int kind, int ch, std::string_view v) {   // the class tests shared by _match_range and the long-run loops
  switch (kind) {
    case RK_ALNUM: return bt_lower(ch) || bt_upper(ch) || bt_digit(ch);
    case RK_LOWER: return bt_lower(ch);
    case RK_UPPER: return bt_upper(ch);
    case RK_DIGITS: return bt_digit(ch);
    case RK_ALPHA: return bt_lower(ch) || bt_upper(ch);
    case RK_COMPLEX_ALNUM:
      return bt_lower(ch) || bt_upper(ch) || bt_digit(ch) || (v.size() >= 2 && bt_in_string(ch, v.substr(1, v.size() - 2)));
    default: return false;
  }
}

struct BtBuilder {
  const Ast& a;
  BtProg& out;
  int depth = 0;
  int choices = 0;   // items that can leave an entry on the choice stack (kBtChoices = 32 on the device)
  void fail(const std::string& why) { if (out.why_not.empty()) out.why_not = why; }
  void push_fail() { BtItem f{}; f.kind = BT_FAIL; out.items.push_back(f); }
  void set(std::array<uint8_t, 32>& t, int c, bool v) { if (v) t[c >> 3] |= (uint8_t)(1u << (c & 7)); }
  void leaf(const Node& n, bool last) {
    BtItem it{};
    it.kind = BT_LEAF;
    it.min = n.min; it.max = n.max;
    if (last) it.flags |= BTF_LAST;
    if (n.min != 1 || n.max != 1) it.flags |= BTF_QUANT;
    if ((it.flags & BTF_QUANT) && !last && ++choices > 30) { fail("more than 30 open choices"); return; }
    if (n.type == N_DIGIT || n.type == N_WORD) it.flags |= BTF_ZERO_OK;
    if (n.type == N_SPACE || n.type == N_DIGIT || n.type == N_WORD || n.type == N_RANGE) it.flags |= BTF_SIMD_TYPE;
    const std::string_view v = a.value(n);
    if (n.type == N_RANGE && v.empty()) { fail("character class without text"); return; }
    if (n.type == N_RANGE && v.size() > 8) it.flags |= BTF_RANGE_LONG;
    if (out.tables.size() / 3 >= 255) { fail("more than 255 leaves"); return; }
    it.tbl = (uint8_t)(out.tables.size() / 3);
    std::array<uint8_t, 32> first{}, scalar{}, cached{};
    const int kind = n.type == N_RANGE ? bt_range_kind(v) : 0;
    for (int c = 0; c < 256; ++c) {
      bool f = false, s = false, m = false;
      switch (n.type) {
        case N_ELEMENT:   // nfa.mojo:771-776 compares the first byte; ast.mojo:419-427 wants a one-byte value
          f = !v.empty() && (uint8_t)v[0] == c;
          s = v.size() == 1 && (uint8_t)v[0] == c;
          m = s;
          break;
        case N_WILDCARD: f = s = m = c != '\n'; break;
        case N_SPACE: f = s = bt_space5(c); m = bt_space_lut(c); break;
        case N_DIGIT: f = s = m = bt_digit(c); break;
        case N_WORD: f = s = m = bt_word(c); break;
        case N_RANGE: {
          bool found = false;   // _match_range, nfa.mojo:930-995
          if (kind != RK_OTHER) found = bt_kind_member(kind, c, v);
          else if (!v.empty()) found = bt_in_range_by_code(c, v);
          f = found == n.positive;
          s = !((!v.empty() && bt_in_range_by_code(c, v)) ^ n.positive);   // ast.mojo:448-456
          bool run = false;     // _apply_quantifier_simd, nfa.mojo:1476-1645 (needs a value)
          if (!v.empty()) run = (kind != RK_OTHER ? bt_kind_member(kind, c, v) : bt_match_char_in_range(v, c)) == n.positive;
          m = run;
          break;
        }
        default: break;
      }
      set(first, c, f); set(scalar, c, s); set(cached, c, m);
    }
    out.tables.push_back(first); out.tables.push_back(scalar); out.tables.push_back(cached);
    out.items.push_back(it);
  }
  void seq(const Node& parent) {
    const int k = a.nkids(parent);
    for (int i = 0; i < k && out.why_not.empty(); ++i) node(a.child(parent, i), i == k - 1);
  }
  void node(const Node& n, bool last) {
    switch (n.type) {
      case N_ELEMENT: case N_WILDCARD: case N_SPACE: case N_DIGIT: case N_WORD: case N_RANGE:
        leaf(n, last);
        break;
      case N_START: case N_END: {
        if (n.min != 1 || n.max != 1) { fail("quantified anchor"); return; }
        BtItem it{};
        it.kind = n.type == N_START ? BT_START : BT_END;
        it.min = it.max = 1;
        out.items.push_back(it);
        break;
      }
      case N_GROUP: {
        if ((n.min != 1 || n.max != 1) && !last) {
          // _match_sequence -> _match_with_backtracking(group node): _try_match_count asks is_match_char of
          // the GROUP node, which is false for every byte -- zero repetitions or nothing (nfa.mojo:1231-1349)
          if (n.min != 0) push_fail();
          if (n.capturing && n.group_id > out.ngroups) out.ngroups = n.group_id;
          break;
        }
        if (n.min != 1 || n.max != 1) {   // last child: the greedy loop of _match_group_with_quantifier
          if (++depth > 16) { fail("groups nested deeper than 16"); return; }
          if (depth > out.max_depth) out.max_depth = depth;
          if (++choices > 30) { fail("more than 30 open choices"); return; }
          BtItem lp{};
          lp.kind = BT_LOOP;
          lp.gid = (int8_t)(n.group_id >= 0 ? (n.group_id > 127 ? 127 : n.group_id) : 0);
          if (n.capturing) lp.flags |= BTF_CAPTURING;
          lp.min = n.min;
          const size_t at = out.items.size();
          out.items.push_back(lp);
          seq(n);
          BtItem le{};
          le.kind = BT_LOOP_END;
          le.min = (int32_t)at;
          le.max = n.max;            // -1: as many as there are bytes left (n - i)
          out.items.push_back(le);
          out.items[at].max = (int32_t)out.items.size();   // first item behind the loop
          if (n.capturing && n.group_id > out.ngroups) out.ngroups = n.group_id;
          --depth;
          break;
        }
        if (++depth > 16) { fail("groups nested deeper than 16"); return; }
        if (depth > out.max_depth) out.max_depth = depth;
        BtItem o{};
        o.kind = BT_OPEN;
        o.gid = (int8_t)(n.group_id >= 0 ? (n.group_id > 127 ? 127 : n.group_id) : 0);
        if (n.capturing) o.flags |= BTF_CAPTURING;
        o.min = o.max = 1;
        out.items.push_back(o);
        seq(n);
        BtItem c = o;
        c.kind = BT_CLOSE;
        out.items.push_back(c);
        if (n.capturing && n.group_id > out.ngroups) out.ngroups = n.group_id;
        --depth;
        break;
      }
      case N_OR: {
        if ((n.min != 1 || n.max != 1) && !last) { if (n.min != 0) push_fail(); break; }   // as a quantified group
        if (a.nkids(n) < 2) { push_fail(); break; }                                         // nfa.mojo:1030-1031
        if (++depth > 16) { fail("groups nested deeper than 16"); return; }
        if (depth > out.max_depth) out.max_depth = depth;
        if (++choices > 30) { fail("more than 30 open choices"); return; }
        BtItem alt{};
        alt.kind = BT_ALT;
        const size_t at = out.items.size();
        out.items.push_back(alt);
        node(a.child(n, 0), true);   // _match_node of the branch itself: a quantified branch is "last"
        BtItem ae{};
        ae.kind = BT_ALT_END;
        const size_t at_end = out.items.size();
        out.items.push_back(ae);
        out.items[at].min = (int32_t)out.items.size();   // first item of B
        node(a.child(n, 1), true);
        BtItem ac{};
        ac.kind = BT_ALT_CLOSE;
        out.items.push_back(ac);
        out.items[at].max = out.items[at_end].max = (int32_t)out.items.size();   // behind the alternation
        --depth;
        break;
      }
      case N_RE:   // _match_re, nfa.mojo:1351-1373: the first child only
        if (a.nkids(n) > 0) node(a.child(n, 0), true);
        break;
      default: fail("node type outside the flat form"); break;
    }
  }
};
}  // namespace

void build_bt(const Ast& a, BtProg& out) {
  out = BtProg();
  BtBuilder b{a, out};
  // _match_re, nfa.mojo:1351-1373: the root RE node matches its first child (the implicit root group)
  if (a.root.type == N_RE) {
    if (a.nkids(a.root) > 0) b.node(a.child(a.root, 0), true);
  } else {
    b.node(a.root, true);
  }
  if (!out.why_not.empty()) return;
  if (out.items.size() > 240) { out.why_not = "more than 240 program items"; return; }
  if (out.ngroups > 9) out.ngroups = 9;   // \1..\9 (matcher.mojo:1797-1802)
  out.ok = true;
}

}  // namespace mrx
