// DFA table construction: pattern-shape dispatcher and per-shape compilers.
// Mirrors src/regex/dfa.mojo (line ranges cited per function).  Later writes
// to the same (state, byte) cell win and negated classes overwrite the whole
// row -- both are reference behaviour the results depend on.
#include <algorithm>

#include "mrx_analysis.hpp"
#include "mrx_engines.hpp"

namespace mrx {

static const std::string kDigits = "0123456789";
static const std::string kLower = "abcdefghijklmnopqrstuvwxyz";
static const std::string kUpper = "ABCDEFGHIJKLMNOPQRSTUVWXYZ";
static const std::string kWord = kLower + kUpper + kDigits + "_";  // aliases.mojo:7-9
static const std::string kSpace = " \t\n\r\f";

static std::string printable_ascii() {  // aliases.mojo:1-3 (32..126)
  std::string s;
  for (int c = 32; c < 127; ++c) s.push_back((char)c);
  return s;
}

void ClassMatcher::build(const std::string& cc) {
  // get_character_class_matcher, simd_ops.mojo:1177-1216: every cached matcher
  // equals a fresh CharacterClassSIMD(cc) except whitespace, which adds \v
  lookup.fill(0);
  for (unsigned char b : cc) lookup[b] = 1;
  if (cc == " \t\n\r\f" || cc == " \t\n\r\f\v") lookup[0x0B] = 1;
  finish();
}

void ClassMatcher::finish() {
  // _detect_ranges, simd_ops.mojo:364-401: >3 runs => 0 (nibble-table scan)
  int count = 0;
  bool in = false;
  for (int c = 0; c < 256; ++c) {
    if (lookup[c]) { in = true; }
    else if (in) { ++count; in = false; }
  }
  if (in) ++count;
  if (count > 3) count = 0;
  num_ranges = count;
  // build_nibble_tables, simd_ops.mojo:63-86
  lo_tbl.fill(0); hi_tbl.fill(0);
  int bucket = 0;
  for (int c = 0; c < 256; ++c)
    if (lookup[c]) {
      const uint8_t bit = (uint8_t)(1u << (bucket & 7));
      lo_tbl[c & 15] |= bit;
      hi_tbl[(c >> 4) & 15] |= bit;
      ++bucket;
    }
}

namespace {

// dfa.mojo:71-168
std::string expand_range(NodeType node_type, std::string_view rs) {
  if (node_type == N_DIGIT) return kDigits;
  if (node_type == N_WORD) return kWord;
  if (node_type == N_SPACE) return kSpace;
  if (rs.empty() || rs.front() != '[' || rs.back() != ']') return std::string(rs);
  if (rs == "[a-z]") return kLower;
  if (rs == "[A-Z]") return kUpper;
  if (rs == "[0-9]") return kDigits;
  if (rs == "[a-zA-Z0-9]") return kLower + kUpper + kDigits;
  if (rs == "[a-zA-Z]") return kLower + kUpper;
  std::string_view inner = rs.substr(1, rs.size() - 2);
  if (!inner.empty() && inner.front() == '^') inner.remove_prefix(1);
  auto clamp_slice = [](const std::string& base, int lo, int hi) {
    lo = std::max(lo, 0); hi = std::max(hi, 0);
    lo = std::min(lo, (int)base.size()); hi = std::min(hi, (int)base.size());
    return lo < hi ? base.substr(lo, hi - lo) : std::string();
  };
  if (inner.size() == 3 && inner[1] == '-') {
    const int s = (unsigned char)inner[0], e = (unsigned char)inner[2];
    if (s >= 'a' && e <= 'z') return clamp_slice(kLower, s - 'a', e - 'a' + 1);
    else if (s >= 'a' && e <= 'Z') return clamp_slice(kUpper, s - 'A', e - 'A' + 1);
    else if (s >= '0' && e <= '9') return clamp_slice(kDigits, s - '0', e - '0' + 1);
  }
  std::string out;
  for (size_t i = 0; i < inner.size();) {
    if (i + 2 < inner.size() && inner[i + 1] == '-') {
      const int s = (unsigned char)inner[i], e = (unsigned char)inner[i + 2];
      for (int c = s; c <= e; ++c) out.push_back((char)c);
      i += 3;
    } else {
      out.push_back(inner[i]);
      ++i;
    }
  }
  return out;
}

struct SeqElement {
  std::string cc;
  int mn, mx;
  bool positive;
  std::vector<std::string> branches;
};
struct SeqInfo {
  std::vector<SeqElement> els;
  bool start_anchor = false, end_anchor = false;
};

struct Builder {
  const Ast& a;
  DfaEngine& d;

  // _add_character_class_transitions_with_logic, dfa.mojo:1748-1803
  void cc(int from, int to, const std::string& cls, bool positive) {
    if (from >= d.nstates()) return;
    auto& row = d.trans[from];
    if (positive) {
      for (unsigned char c : cls) row[c] = (int16_t)to;
    } else {
      row.fill((int16_t)to);
      for (unsigned char c : cls) row[c] = -1;
    }
  }
  void set(int from, int c, int to) { d.trans[from][(unsigned char)c] = (int16_t)to; }

  int find_or_create(int from, int c) {  // dfa.mojo:1245-1268
    const int t = d.trans[from][(unsigned char)c];
    if (t != -1) return t;
    const int ni = d.add_state();
    set(from, c, ni);
    return ni;
  }

  void accepting_only() { d.add_state(true); }  // _create_accepting_state

  // dfa.mojo:308-352
  void literal(const std::string& lit, bool hs, bool he) {
    d.has_start_anchor = hs; d.has_end_anchor = he; d.literal = lit;
    if (lit.empty()) { accepting_only(); return; }
    if (!hs && !he) d.is_pure_literal = true;
    for (size_t i = 0; i < lit.size(); ++i) {
      const int s = d.add_state();
      set(s, lit[i], (int)i + 1);
    }
    d.add_state(true);
  }

  // dfa.mojo:375-496
  void single_class(const std::string& cls, int mn, int mx, bool positive) {
    if (mn >= 0 && positive) {
      d.matcher.build(cls);
      d.has_matcher = true;
      d.scan_eligible = (mx == -1);
    }
    if (mn == 0) {
      d.add_state(true); d.add_state(true);
      cc(0, 1, cls, positive);
      if (mx == -1 || mx > 1) cc(1, 1, cls, positive);
    } else if (mn == 1) {
      d.add_state(false); d.add_state(true);
      cc(0, 1, cls, positive);
      if (mx == -1) cc(1, 1, cls, positive);
      else if (mx > 1)
        for (int k = 2; k <= mx; ++k) { d.add_state(true); cc(k - 1, k, cls, positive); }
    } else {
      for (int k = 0; k <= mn; ++k) {
        d.add_state(k >= mn);
        if (k > 0) cc(k - 1, k, cls, positive);
      }
      if (mx == -1) { const int last = d.nstates() - 1; cc(last, last, cls, positive); }
      else if (mx > mn)
        for (int k = mn + 1; k <= mx; ++k) { d.add_state(true); cc(k - 1, k, cls, positive); }
    }
  }

  // dfa.mojo:498-605
  void sequential(const SeqInfo& info) {
    d.has_start_anchor = info.start_anchor; d.has_end_anchor = info.end_anchor;
    if (info.els.empty()) { accepting_only(); return; }
    int cur = 0;
    const int n = (int)info.els.size();
    for (int idx = 0; idx < n; ++idx) {
      const SeqElement& el = info.els[idx];
      const bool last = idx == n - 1;
      if (el.mn == 0) {
        if (idx == 0) { d.add_state(!last); cur = 0; }
        const int m = d.add_state(true);
        cc(cur, m, el.cc, el.positive);
        if (el.mx == -1) cc(m, m, el.cc, el.positive);
        cur = m;
      } else {
        for (int k = 0; k < el.mn; ++k) {
          const int si = d.add_state((k >= el.mn - 1) && last);
          if (k == 0) cc(cur, si, el.cc, el.positive);
          else cc(si - 1, si, el.cc, el.positive);
          cur = si;
        }
        if (el.mx == -1) cc(cur, cur, el.cc, el.positive);
        else if (el.mx > el.mn)
          for (int k = 0; k < el.mx - el.mn; ++k) {
            const int si = d.add_state(last);
            cc(cur, si, el.cc, el.positive);
            cur = si;
          }
      }
    }
  }

  // dfa.mojo:607-871
  void multi_class(const SeqInfo& info) {
    d.has_start_anchor = info.start_anchor; d.has_end_anchor = info.end_anchor;
    if (info.els.empty()) { accepting_only(); return; }
    if (info.els[0].branches.empty() && !info.els[0].cc.empty()) {
      d.matcher.build(info.els[0].cc);
      d.has_matcher = true;  // class skip only; never scan eligible
    }
    int cur = 0;
    const int n = (int)info.els.size();
    for (int idx = 0; idx < n; ++idx) {
      const SeqElement& el = info.els[idx];
      const bool last = idx == n - 1;
      bool rest_optional = true;
      for (int j = idx + 1; j < n; ++j)
        if (info.els[j].mn > 0) { rest_optional = false; break; }
      if (!el.branches.empty()) {  // dfa.mojo:660-697
        if (idx == 0) { d.add_state(); cur = 0; }
        const int endi = d.add_state(last || rest_optional);
        for (const std::string& br : el.branches) {
          int prev = cur;
          for (size_t ci = 0; ci < br.size(); ++ci) {
            if (ci == br.size() - 1) set(prev, br[ci], endi);
            else { const int mid = d.add_state(); set(prev, br[ci], mid); prev = mid; }
          }
        }
        cur = endi;
        continue;
      }
      if (el.mn == 0) {  // dfa.mojo:699-744
        if (idx == 0) { d.add_state(rest_optional); cur = 0; }
        const int m = d.add_state(last || rest_optional);
        cc(cur, m, el.cc, el.positive);
        if (el.mx == -1) cc(m, m, el.cc, el.positive);
        cur = (idx == 0) ? 0 : m;
      } else if (el.mn == 1) {  // dfa.mojo:746-807
        if (idx == 0) { d.add_state(); cur = 0; }
        const int m = d.add_state(last || rest_optional);
        cc(cur, m, el.cc, el.positive);
        if (idx == 1 && cur == 0 && info.els[0].mn == 0) cc(1, m, el.cc, el.positive);
        if (el.mx == -1) cc(m, m, el.cc, el.positive);
        else if (el.mx > 1)
          for (int k = 2; k <= el.mx; ++k) {
            const int ai = d.add_state(last);
            cc(m + k - 2, ai, el.cc, el.positive);
          }
        cur = m;
      } else {  // dfa.mojo:809-869
        if (idx == 0) { d.add_state(); cur = 0; }
        for (int k = 0; k < el.mn; ++k) {
          const int si = d.add_state((k >= el.mn - 1) && last);
          if (k > 0) cc(si - 1, si, el.cc, el.positive);
          else cc(cur, si, el.cc, el.positive);
          cur = si;
        }
        if (el.mx == -1) cc(cur, cur, el.cc, el.positive);
        else if (el.mx > el.mn)
          for (int k = el.mn + 1; k <= el.mx; ++k) {
            const int oi = d.add_state(last);
            cc(cur, oi, el.cc, el.positive);
            cur = oi;
          }
      }
    }
  }

  // chain helpers shared by (p)?, (p)*, (p)+ builders (dfa.mojo:982-1074, 1147-1243)
  template <class F>
  void chain(const std::string& text, F last) {
    int cur = 0;
    for (size_t i = 0; i < text.size(); ++i) {
      if (i == text.size() - 1) last(cur, text[i]);
      else { const int ni = d.add_state(); set(cur, text[i], ni); cur = ni; }
    }
  }
  void one_or_more(const std::string& text) {
    chain(text, [&](int cur, char c) {
      const int li = d.add_state(true);
      set(cur, c, li);
      set(li, text[0], text.size() > 1 ? 1 : li);
    });
  }

  void fresh() { d.trans.clear(); d.accepting.clear(); d.add_state(); }

  // dfa.mojo:873-928
  void alternation(const std::vector<std::string>& branches) {
    fresh();
    const int acc = d.add_state(true);
    for (const std::string& br : branches) {
      if (br.empty()) continue;
      int cur = 0;
      for (size_t j = 0; j < br.size(); ++j) {
        if (j == br.size() - 1) set(cur, br[j], acc);
        else cur = find_or_create(cur, br[j]);
      }
    }
  }

  // dfa.mojo:930-980 (group) and :1076-1145 (simple quantifier)
  void quantified_text(const std::string& text, int mn, int mx, bool group_form) {
    fresh();
    if (text.empty())
      throw DfaCompileError(group_form ? "Empty quantified group" : "Empty quantifier pattern");
    const int acc = d.add_state(true);
    if (mn == 0 && mx == 1) {
      if (group_form) set(0, 0, acc);  // "epsilon" is a transition on byte 0 (dfa.mojo:989)
      else d.accepting[0] = 1;         // dfa.mojo:1154-1156
      chain(text, [&](int cur, char c) { set(cur, c, acc); });
    } else if (mn == 0 && mx == -1) {
      d.accepting[0] = 1;
      chain(text, [&](int cur, char c) { set(cur, c, 0); });
    } else if (mn == 1 && mx == -1) {
      one_or_more(text);
    } else {
      throw DfaCompileError(group_form ? "Unsupported quantifier range for group"
                                       : "Unsupported quantifier type for simple quantifier");
    }
  }

  // dfa.mojo:1270-1361
  void wildcard(int mn, int mx) {
    fresh();
    const int acc = d.add_state(true);
    auto all_but_nl = [&](int from, int to) {
      for (int c = 0; c < 256; ++c)
        if (c != '\n') d.trans[from][c] = (int16_t)to;
    };
    if (mn == 0 && mx == 1) { d.accepting[0] = 1; all_but_nl(0, acc); }
    else if (mn == 0 && mx == -1) { d.accepting[0] = 1; all_but_nl(0, 0); }
    else if (mn == 1 && mx == -1) {
      const int li = d.add_state(true);
      all_but_nl(0, li); all_but_nl(li, li);
    } else if (mn == 1 && mx == 1) { all_but_nl(0, acc); }
    else throw DfaCompileError("Unsupported quantifier type for wildcard quantifier");
  }

  // dfa.mojo:1363-1463
  void common_prefix_alt(const std::vector<std::string>& branches) {
    fresh();
    if (branches.empty()) return;
    const std::string prefix = common_prefix(branches);
    int cur = 0;
    for (char c : prefix) cur = find_or_create(cur, c);
    for (const std::string& br : branches) {
      if (br.size() == prefix.size()) { d.accepting[cur] = 1; continue; }
      int sc = cur;
      for (size_t j = prefix.size(); j < br.size(); ++j) {
        if (j == br.size() - 1) { const int t = find_or_create(sc, br[j]); d.accepting[t] = 1; }
        else sc = find_or_create(sc, br[j]);
      }
    }
  }

  // dfa.mojo:1507-1686
  void quantified_alt(const std::vector<std::string>& branches, int mn, int mx) {
    fresh();
    auto paths = [&](int origin, int final_state) {
      for (const std::string& br : branches) {
        int cur = origin;
        for (size_t j = 0; j < br.size(); ++j) {
          if (j == br.size() - 1) set(cur, br[j], final_state);
          else cur = find_or_create(cur, br[j]);
        }
      }
    };
    if (mn == 0 && mx == 1) { d.accepting[0] = 1; const int acc = d.add_state(true); paths(0, acc); }
    else if (mn == 0 && mx == -1) { d.accepting[0] = 1; paths(0, 0); }
    else if (mn == 1 && mx == -1) { const int li = d.add_state(true); paths(0, li); paths(li, li); }
    else throw DfaCompileError("Unsupported quantifier type for quantified alternation group");
  }
};

// ---- recognisers (dfa.mojo:2499-3589) ------------------------------------------
struct Shape {
  const Ast& a;
  const Node& root() const { return a.root; }
  bool single_kid(const Node& n) const { return a.nkids(n) == 1; }

  static bool dwr(NodeType t) { return t == N_DIGIT || t == N_WORD || t == N_RANGE; }

  std::string element_class(const Node& e) const {  // dfa.mojo:2885-2908
    switch (e.type) {
      case N_DIGIT: return kDigits;
      case N_WORD: return kWord;
      case N_RANGE: return expand_range(e.type, a.value(e));
      case N_SPACE: return kSpace;
      case N_WILDCARD: return printable_ascii();
      case N_ELEMENT: return a.has_value(e) ? std::string(a.value(e)) : std::string();
      default: return "";
    }
  }
  bool char_class_group(const Node& n) const {  // dfa.mojo:2711-2733
    if (n.type != N_GROUP) return false;
    bool has_cc = false;
    for (int i = 0; i < a.nkids(n); ++i) {
      const Node& c = a.child(n, i);
      if (element_class(c).empty()) return false;
      if (c.type != N_ELEMENT) has_cc = true;
    }
    return has_cc;
  }
  bool only_elements(const Node& g) const {  // dfa.mojo:3100-3120, 3222-3242
    if (g.type != N_GROUP) return false;
    for (int i = 0; i < a.nkids(g); ++i)
      if (a.child(g, i).type != N_ELEMENT) return false;
    return true;
  }
  bool literal_alt_group(const Node& n) const {  // dfa.mojo:2736-2771
    if (n.type != N_GROUP || a.nkids(n) != 1) return false;
    const Node& c = a.child(n, 0);
    if (c.type != N_OR) return false;
    bool has_branch = false;
    std::vector<const Node*> st{&c};
    while (!st.empty()) {
      const Node* cur = st.back(); st.pop_back();
      if (cur->type == N_GROUP) {
        if (!only_elements(*cur)) return false;
        has_branch = true;
      } else if (cur->type == N_OR) {
        if (a.nkids(*cur) != 2) return false;
        st.push_back(&a.child(*cur, 1));
        st.push_back(&a.child(*cur, 0));
      } else return false;
    }
    return has_branch;
  }
  std::vector<std::string> alt_group_branches(const Node& n) const {  // dfa.mojo:2774-2812
    std::vector<std::string> out;
    std::vector<const Node*> st;
    if (a.nkids(n) > 0) st.push_back(&a.child(n, 0));
    while (!st.empty()) {
      const Node* cur = st.back(); st.pop_back();
      if (cur->type == N_GROUP) {
        std::string br;
        for (int i = 0; i < a.nkids(*cur); ++i) {
          const Node& c = a.child(*cur, i);
          if (c.type == N_ELEMENT && a.has_value(c)) br += a.value(c);
        }
        out.push_back(br);
      } else if (cur->type == N_OR) {
        if (a.nkids(*cur) >= 2) st.push_back(&a.child(*cur, 1));
        if (a.nkids(*cur) >= 1) st.push_back(&a.child(*cur, 0));
      }
    }
    return out;
  }

  const Node* top_group() const {  // RE -> single GROUP child
    if (root().type != N_RE || a.nkids(root()) != 1) return nullptr;
    const Node& c = a.child(root(), 0);
    return c.type == N_GROUP ? &c : nullptr;
  }

  bool multi_class_sequence() const {  // dfa.mojo:2815-2882
    const Node* g = top_group();
    if (!g || a.nkids(*g) < 2) return false;
    int cc = 0;
    for (int i = 0; i < a.nkids(*g); ++i) {
      const Node& e = a.child(*g, i);
      if (dwr(e.type) || e.type == N_SPACE || e.type == N_WILDCARD) ++cc;
      else if (e.type == N_ELEMENT && e.min == 1 && e.max == 1) {}
      else if (e.type == N_GROUP && literal_alt_group(e)) {}
      else if (e.type == N_GROUP && char_class_group(e)) ++cc;
      else return false;
    }
    return cc >= 2;
  }
  SeqInfo multi_class_info() const {  // dfa.mojo:2911-2970
    SeqInfo info;
    std::tie(info.start_anchor, info.end_anchor) = pattern_has_anchors(a);
    const Node* g = top_group();
    if (!g) return info;
    for (int i = 0; i < a.nkids(*g); ++i) {
      const Node& e = a.child(*g, i);
      if (e.type == N_GROUP && literal_alt_group(e)) {
        SeqElement pe{"", 1, 1, true, alt_group_branches(e)};
        info.els.push_back(pe);
      } else if (e.type == N_GROUP && char_class_group(e)) {
        for (int j = 0; j < a.nkids(e); ++j) {
          const Node& sub = a.child(e, j);
          std::string sc = element_class(sub);
          if (!sc.empty()) info.els.push_back({sc, sub.min, sub.max, sub.positive, {}});
        }
      } else {
        std::string cc = element_class(e);
        if (!cc.empty()) info.els.push_back({cc, e.min, e.max, e.positive, {}});
      }
    }
    return info;
  }

  bool simple_class() const {  // dfa.mojo:2499-2529
    if (multi_class_sequence()) return false;
    if (root().type == N_RE && a.nkids(root()) == 1) {
      const Node& c = a.child(root(), 0);
      if (dwr(c.type)) return true;
      if (c.type == N_GROUP && a.nkids(c) == 1) return dwr(a.child(c, 0).type);
    }
    return false;
  }
  bool pure_anchor(const Node& n) const {  // dfa.mojo:2605-2629
    if (n.type == N_START || n.type == N_END) return true;
    if (n.type == N_RE) return a.nkids(n) > 0 && pure_anchor(a.child(n, 0));
    if (n.type == N_GROUP) {
      for (int i = 0; i < a.nkids(n); ++i)
        if (!pure_anchor(a.child(n, i))) return false;
      return true;
    }
    return false;
  }
  bool sequential_classes() const {  // dfa.mojo:2632-2663
    const Node* g = top_group();
    if (!g) return false;
    for (int i = 0; i < a.nkids(*g); ++i)
      if (!dwr(a.child(*g, i).type)) return false;
    return a.nkids(*g) >= 2;
  }
  SeqInfo sequential_info() const {  // dfa.mojo:2666-2708
    SeqInfo info;
    std::tie(info.start_anchor, info.end_anchor) = pattern_has_anchors(a);
    const Node* g = top_group();
    if (!g) return info;
    for (int i = 0; i < a.nkids(*g); ++i) {
      const Node& e = a.child(*g, i);
      std::string cc;
      if (e.type == N_DIGIT) cc = kDigits;
      else if (e.type == N_WORD) cc = kWord;
      else if (e.type == N_RANGE) cc = expand_range(e.type, a.value(e));
      else continue;
      info.els.push_back({cc, e.min, e.max, e.positive, {}});
    }
    return info;
  }
  bool mixed_sequential() const {  // dfa.mojo:2973-3015
    const Node* g = top_group();
    if (!g || a.nkids(*g) < 3) return false;
    bool has_cc = false, has_opt = false;
    for (int i = 0; i < a.nkids(*g); ++i) {
      const Node& e = a.child(*g, i);
      if (dwr(e.type)) has_cc = true;
      else if (e.type == N_ELEMENT && e.min == 0 && e.max == 1) has_opt = true;
    }
    return has_cc && has_opt;
  }

  bool simple_alt_branches(const Node& o) const {  // dfa.mojo:3054-3097
    if (o.type != N_OR) return false;
    for (int i = 0; i < a.nkids(o); ++i) {
      const Node& br = a.child(o, i);
      if (br.type == N_GROUP) {
        if (!only_elements(br)) {
          const Node* in = &br;
          while (in->type == N_GROUP && a.nkids(*in) == 1) in = &a.child(*in, 0);
          if (in->type == N_OR) { if (!simple_alt_branches(*in)) return false; }
          else return false;
        }
      } else if (br.type == N_ELEMENT) {
      } else if (br.type == N_OR) {
        if (!simple_alt_branches(br)) return false;
      } else return false;
    }
    return true;
  }
  bool alternation() const {  // dfa.mojo:3034-3051, 3309-3342
    if (root().type != N_RE || a.nkids(root()) != 1) return false;
    const Node* n = &a.child(root(), 0);
    if (n->type == N_OR) return simple_alt_branches(*n);
    // single-child GROUPs are unwrapped WITHOUT looking at their quantifier
    while (n->type == N_GROUP && a.nkids(*n) == 1) n = &a.child(*n, 0);
    return n->type == N_OR && simple_alt_branches(*n);
  }
  const Node* find_or(const Node& n) const {  // dfa.mojo:3146-3169
    if (n.type == N_OR) return &n;
    for (int i = 0; i < a.nkids(n); ++i)
      if (const Node* f = find_or(a.child(n, i))) return f;
    return nullptr;
  }
  std::string branch_text(const Node& br) const {  // dfa.mojo:3172-3194
    if (br.type == N_ELEMENT && a.has_value(br)) return std::string(a.value(br));
    std::string out;
    if (br.type == N_GROUP)
      for (int i = 0; i < a.nkids(br); ++i) {
        const Node& c = a.child(br, i);
        if (c.type == N_ELEMENT && a.has_value(c)) out += a.value(c);
      }
    return out;
  }
  void all_alt_branches(const Node& o, std::vector<std::string>& out) const {  // :3265-3306
    for (int i = 0; i < a.nkids(o); ++i) {
      const Node& br = a.child(o, i);
      if (br.type == N_OR) { all_alt_branches(br, out); continue; }
      if (br.type == N_GROUP) {
        const Node* in = &br;
        while (in->type == N_GROUP && a.nkids(*in) == 1) in = &a.child(*in, 0);
        if (in->type == N_OR) { all_alt_branches(*in, out); continue; }
      }
      std::string t = branch_text(br);
      if (!t.empty()) out.push_back(t);
    }
  }

  const Node* quantified_inner_group() const {  // RE -> GROUP(1 kid) -> GROUP
    const Node* g = top_group();
    if (!g || a.nkids(*g) != 1) return nullptr;
    const Node& in = a.child(*g, 0);
    return in.type == N_GROUP ? &in : nullptr;
  }
  bool quantified_group() const {  // dfa.mojo:3197-3242
    const Node* in = quantified_inner_group();
    return in && (in->min != 1 || in->max != 1) && only_elements(*in);
  }
  bool simple_quantifier() const {  // dfa.mojo:3345-3383
    const Node* g = top_group();
    if (!g || a.nkids(*g) == 0) return false;
    bool has_q = false;
    for (int i = 0; i < a.nkids(*g); ++i) {
      const Node& c = a.child(*g, i);
      if (c.type != N_ELEMENT) return false;
      if ((c.min == 0 && c.max == -1) || (c.min == 1 && c.max == -1) || (c.min == 0 && c.max == 1))
        has_q = true;
    }
    return has_q;
  }
  bool wildcard_quantifier() const {  // dfa.mojo:3386-3415
    const Node* g = top_group();
    if (!g || a.nkids(*g) != 1) return false;
    const Node& w = a.child(*g, 0);
    if (w.type != N_WILDCARD) return false;
    return (w.min == 0 && w.max == -1) || (w.min == 1 && w.max == -1) ||
           (w.min == 0 && w.max == 1) || (w.min == 1 && w.max == 1);
  }
  bool strict_branches(const Node& n, std::vector<std::string>& out) const {  // :3462-3486
    if (n.type == N_OR)
      return strict_branches(a.child(n, 0), out) && strict_branches(a.child(n, 1), out);
    if (n.type == N_GROUP) {
      std::string t;
      for (int i = 0; i < a.nkids(n); ++i) {
        const Node& e = a.child(n, i);
        if (e.type != N_ELEMENT) return false;
        t += a.value(e);
      }
      out.push_back(t);
      return true;
    }
    return false;
  }
  void lenient_branches(const Node& n, std::vector<std::string>& out) const {  // :1396-1418
    if (n.type == N_OR) {
      lenient_branches(a.child(n, 0), out);
      lenient_branches(a.child(n, 1), out);
    } else if (n.type == N_GROUP) {
      std::string t;
      for (int i = 0; i < a.nkids(n); ++i) {
        const Node& e = a.child(n, i);
        if (e.type == N_ELEMENT) t += a.value(e);
      }
      out.push_back(t);
    }
  }
  const Node* inner_or() const {  // RE -> GROUP -> GROUP(1 kid) -> OR
    const Node* in = quantified_inner_group();
    if (!in || a.nkids(*in) != 1) return nullptr;
    const Node& o = a.child(*in, 0);
    return o.type == N_OR ? &o : nullptr;
  }
  bool common_prefix_alternation() const {  // dfa.mojo:3418-3459
    const Node* o = inner_or();
    if (!o) return false;
    std::vector<std::string> br;
    if (!strict_branches(*o, br) || br.size() < 2) return false;
    return common_prefix(br).size() >= 2;
  }
  bool quantified_alternation_group() const {  // dfa.mojo:3525-3562
    const Node* in = quantified_inner_group();
    const Node* o = inner_or();
    if (!in || !o) return false;
    if (in->min == 1 && in->max == 1) return false;
    std::vector<std::string> br;
    return strict_branches(*o, br);
  }
};

}  // namespace

void compile_dfa_pattern(const Ast& a, DfaEngine& d) {
  // dfa.mojo:2385-2496: first matching shape wins, in this order
  d = DfaEngine();
  Builder b{a, d};
  Shape s{a};
  auto anchors = pattern_has_anchors(a);
  auto set_anchors = [&] { d.has_start_anchor = anchors.first; d.has_end_anchor = anchors.second; };
  if (is_literal_pattern(a)) {
    b.literal(get_literal_string(a), anchors.first, anchors.second);
    d.shape = "literal";
  } else if (s.pure_anchor(a.root)) {
    b.literal("", anchors.first, anchors.second);
    d.shape = "pure_anchor";
  } else if (s.simple_class()) {
    // _extract_character_class_info, dfa.mojo:2532-2602
    const Node& c = a.child(a.root, 0);
    const Node& cn = Shape::dwr(c.type) ? c : a.child(c, 0);
    std::string cls;
    if (cn.type == N_DIGIT) cls = kDigits;
    else if (cn.type == N_WORD) cls = kWord;
    else {
      if (!a.has_value(cn)) throw DfaCompileError("character class without value");
      cls = expand_range(N_RE, a.value(cn));
    }
    b.single_class(cls, cn.min, cn.max, cn.positive);
    set_anchors();
    d.shape = "single_class";
  } else if (s.multi_class_sequence()) {
    b.multi_class(s.multi_class_info());
    d.shape = "multi_class_sequence";
  } else if (s.sequential_classes()) {
    b.sequential(s.sequential_info());
    d.shape = "sequential";
  } else if (s.mixed_sequential()) {
    b.multi_class(s.multi_class_info());
    d.shape = "mixed_sequential";
  } else if (s.alternation()) {
    const Node* o = s.find_or(a.root);
    if (!o) throw DfaCompileError("No OR node found in alternation pattern");
    std::vector<std::string> br;
    s.all_alt_branches(*o, br);
    b.alternation(br);
    set_anchors();
    d.shape = "alternation";
  } else if (s.quantified_group()) {
    const Node* in = s.quantified_inner_group();
    std::string text;
    for (int i = 0; i < a.nkids(*in); ++i) text += a.value(a.child(*in, i));
    b.quantified_text(text, in->min, in->max, true);
    set_anchors();
    d.shape = "quantified_group";
  } else if (s.simple_quantifier()) {
    const Node* g = s.top_group();
    int mn = 1, mx = 1;
    std::string text;
    for (int i = 0; i < a.nkids(*g); ++i) {
      const Node& e = a.child(*g, i);
      if (e.min == 0 && e.max == -1) { mn = 0; mx = -1; }
      else if (e.min == 1 && e.max == -1) { mn = 1; mx = -1; }
      else if (e.min == 0 && e.max == 1) { mn = 0; mx = 1; }
      text += a.value(e);
    }
    b.quantified_text(text, mn, mx, false);
    set_anchors();
    d.shape = "simple_quantifier";
  } else if (s.wildcard_quantifier()) {
    const Node& w = a.child(*s.top_group(), 0);
    b.wildcard(w.min, w.max);
    set_anchors();
    d.shape = "wildcard_quantifier";
  } else if (s.common_prefix_alternation()) {
    std::vector<std::string> br;
    s.lenient_branches(*s.inner_or(), br);
    b.common_prefix_alt(br);
    set_anchors();
    d.shape = "common_prefix_alternation";
  } else if (s.quantified_alternation_group()) {
    const Node* in = s.quantified_inner_group();
    std::vector<std::string> br;
    s.lenient_branches(*s.inner_or(), br);
    b.quantified_alt(br, in->min, in->max);
    set_anchors();
    d.shape = "quantified_alternation_group";
  } else {
    throw DfaCompileError("Pattern too complex for current DFA implementation");
  }
}

}  // namespace mrx
