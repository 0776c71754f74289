// See mrx_analysis.hpp.  Reference lines are cited per function.
#include "mrx_analysis.hpp"

#include <algorithm>

namespace mrx {
namespace {

constexpr int kMaxLiteralQuantReps = 10;  // optimizer.mojo:33

inline bool is_class(NodeType t) {
  return t == N_RANGE || t == N_DIGIT || t == N_WORD || t == N_SPACE;
}

struct Analyzer {
  const Ast& a;

  // optimizer.mojo:320-349
  static Complexity quantifier(const Node& n) {
    if (n.min == 1 && n.max == 1) return CX_SIMPLE;
    if ((n.min == 0 && n.max == -1) || (n.min == 1 && n.max == -1)) return CX_SIMPLE;
    if (n.min == 0 && n.max == 1) return CX_SIMPLE;
    if (n.max != -1 && n.max - n.min <= 10) return CX_SIMPLE;
    if (n.max != -1 && n.max - n.min <= 100) return CX_MEDIUM;
    return CX_COMPLEX;
  }

  // optimizer.mojo:260-318
  Complexity node(const Node& n, int depth) const {
    switch (n.type) {
      case N_RE: return a.nkids(n) == 0 ? CX_SIMPLE : node(a.child(n, 0), depth);
      case N_ELEMENT: case N_WILDCARD: case N_SPACE: case N_DIGIT: case N_WORD: case N_RANGE:
        return quantifier(n);
      case N_START: case N_END: return CX_SIMPLE;
      case N_OR: return alternation(n, depth);
      case N_GROUP: return multi_class_seq(n) ? CX_SIMPLE : group(n, depth);
      default: return CX_COMPLEX;
    }
  }

  // optimizer.mojo:351-401
  Complexity alternation(const Node& n, int depth) const {
    if (depth > 2) {
      if (literal_heavy(n)) return CX_MEDIUM;
      if (depth <= 4 && common_prefix_tree(n)) return CX_SIMPLE;
      return CX_COMPLEX;
    }
    Complexity mx = CX_SIMPLE;
    for (int i = 0; i < a.nkids(n); ++i) {
      const Complexity c = node(a.child(n, i), depth + 1);
      if (c == CX_COMPLEX) return CX_COMPLEX;
      if (c == CX_MEDIUM) mx = CX_MEDIUM;
    }
    if (mx == CX_SIMPLE && a.nkids(n) <= 8 && !nested_alternation(n)) return CX_SIMPLE;
    return CX_MEDIUM;
  }

  // optimizer.mojo:403-490
  Complexity group(const Node& n, int depth) const {
    if (depth > 4) return CX_COMPLEX;
    const Complexity qc = quantifier(n);
    if (qc == CX_COMPLEX) return CX_COMPLEX;
    if (n.min != 1 || n.max != 1) {
      if (simple_quantified_group(n)) {
      } else if (quantified_alternation_group(n)) {
        return CX_SIMPLE;
      } else {
        return CX_MEDIUM;
      }
    }
    if (a.nkids(n) == 1) {
      const Node& only = a.child(n, 0);
      if ((only.type == N_OR || only.type == N_GROUP) && all_literal_branches(only))
        return CX_SIMPLE;
    }
    Complexity mx = CX_SIMPLE;
    for (int i = 0; i < a.nkids(n); ++i) {
      const Complexity c = node(a.child(n, i), depth + 1);
      if (c == CX_COMPLEX) return CX_COMPLEX;
      if (c == CX_MEDIUM) mx = CX_MEDIUM;
    }
    if (mx == CX_SIMPLE && qc == CX_SIMPLE) {
      bool all_lit = true;
      for (int i = 0; i < a.nkids(n); ++i) {
        const Node& c = a.child(n, i);
        const bool lit = c.type == N_ELEMENT && c.min == c.max && c.min >= 1 &&
                         c.min <= kMaxLiteralQuantReps;
        if (!(lit || c.type == N_START || c.type == N_END)) { all_lit = false; break; }
      }
      if (all_lit && a.nkids(n) <= 20) return CX_SIMPLE;
      if (a.nkids(n) <= 5) return CX_SIMPLE;
      return CX_MEDIUM;
    }
    return CX_MEDIUM;
  }

  // optimizer.mojo:492-510
  bool all_literal_branches(const Node& n) const {
    if (n.type == N_GROUP) {
      if (a.nkids(n) == 1) return all_literal_branches(a.child(n, 0));
      for (int j = 0; j < a.nkids(n); ++j)
        if (a.child(n, j).type != N_ELEMENT) return false;
      return true;
    }
    if (n.type == N_OR) {
      for (int i = 0; i < a.nkids(n); ++i)
        if (!all_literal_branches(a.child(n, i))) return false;
      return true;
    }
    return n.type == N_ELEMENT;
  }

  // optimizer.mojo:512-555
  bool multi_class_seq(const Node& n) const {
    if (n.type != N_GROUP || a.nkids(n) < 2) return false;
    int cc = 0;
    for (int i = 0; i < a.nkids(n); ++i) {
      const Node& e = a.child(n, i);
      if (is_class(e.type) || e.type == N_WILDCARD) ++cc;
      else if (e.type == N_ELEMENT && e.min == 1 && e.max == 1) {}
      else return false;
    }
    return cc >= 2;
  }

  // optimizer.mojo:557-586
  bool simple_quantified_group(const Node& n) const {
    if (!((n.min == 0 && n.max == 1) || (n.min == 0 && n.max == -1) ||
          (n.min == 1 && n.max == -1)))
      return false;
    for (int i = 0; i < a.nkids(n); ++i) {
      const Node& c = a.child(n, i);
      if (c.type != N_ELEMENT || c.min != 1 || c.max != 1) return false;
    }
    return true;
  }

  // optimizer.mojo:588-619
  bool group_contains_or(const Node& n) const {
    for (int i = 0; i < a.nkids(n); ++i) {
      const Node& c = a.child(n, i);
      if (c.type == N_OR) return true;
      if (c.type == N_GROUP && group_contains_or(c)) return true;
    }
    return false;
  }
  bool nested_alternation(const Node& n) const {
    for (int i = 0; i < a.nkids(n); ++i) {
      const Node& c = a.child(n, i);
      if (c.type == N_OR) { if (nested_alternation(c)) return true; }
      else if (c.type == N_GROUP) { if (group_contains_or(c)) return true; }
    }
    return false;
  }

  // optimizer.mojo:648-673 / :743-768
  bool literal_branches(const Node& n, std::vector<std::string>& out) const {
    if (n.type == N_OR)
      return literal_branches(a.child(n, 0), out) && literal_branches(a.child(n, 1), out);
    if (n.type == N_GROUP) {
      std::string s;
      for (int i = 0; i < a.nkids(n); ++i) {
        const Node& e = a.child(n, i);
        if (e.type != N_ELEMENT) return false;
        s += a.value(e);
      }
      out.push_back(s);
      return true;
    }
    return false;
  }
  bool common_prefix_tree(const Node& n) const {  // optimizer.mojo:621-646
    std::vector<std::string> br;
    if (!literal_branches(n, br) || br.size() < 2) return false;
    return common_prefix(br).size() >= 2;
  }
  bool quantified_alternation_group(const Node& n) const {  // optimizer.mojo:712-741
    if (n.min == 1 && n.max == 1) return false;
    if (a.nkids(n) != 1) return false;
    const Node& o = a.child(n, 0);
    if (o.type != N_OR) return false;
    std::vector<std::string> br;
    return literal_branches(o, br);
  }

  // optimizer.mojo:770-845
  bool simple_dfa_node(const Node& n) const {
    if (n.type == N_ELEMENT || is_class(n.type) || n.type == N_WILDCARD)
      return n.max <= 10 || n.max == -1;
    return n.type == N_START || n.type == N_END;
  }
  bool dfa_compatible_branch(const Node& n) const {
    if (n.type == N_ELEMENT || is_class(n.type)) return true;
    if (n.type == N_GROUP) {
      if (a.nkids(n) <= 4) {
        for (int i = 0; i < a.nkids(n); ++i)
          if (!simple_dfa_node(a.child(n, i))) return false;
        return true;
      }
    } else if (n.type == N_OR) {
      if (a.nkids(n) <= 4) {
        for (int i = 0; i < a.nkids(n); ++i)
          if (!dfa_compatible_branch(a.child(n, i))) return false;
        return true;
      }
    }
    return false;
  }
  bool literal_heavy(const Node& n) const {
    if (n.type != N_OR) return false;
    int ok = 0;
    const int total = a.nkids(n);
    for (int i = 0; i < total; ++i) ok += dfa_compatible_branch(a.child(n, i)) ? 1 : 0;
    return ok * 5 >= total * 4;
  }
};

// optimizer.mojo:866-900
bool literal_sequence(const Ast& a, const Node& n) {
  if (n.type == N_ELEMENT) {
    if (n.min == 1 && n.max == 1) return true;
    return n.min == n.max && n.min >= 1 && n.min <= kMaxLiteralQuantReps;
  }
  if (n.type == N_START || n.type == N_END) return true;
  if (n.type == N_GROUP) {
    for (int i = 0; i < a.nkids(n); ++i) {
      const Node& c = a.child(n, i);
      if (c.type == N_GROUP) return false;
      if (!literal_sequence(a, c)) return false;
    }
    return true;
  }
  return false;
}

// optimizer.mojo:921-953
std::string literal_chars(const Ast& a, const Node& n) {
  if (n.type == N_ELEMENT) {
    if (!a.has_value(n)) return "";
    std::string v(a.value(n));
    if (n.min <= 1) return v;
    std::string out;
    for (int i = 0; i < n.min; ++i) out += v;
    return out;
  }
  if (n.type == N_GROUP) {
    std::string out;
    for (int i = 0; i < a.nkids(n); ++i) out += literal_chars(a, a.child(n, i));
    return out;
  }
  return "";
}

std::pair<bool, bool> anchors_rec(const Ast& a, const Node& n) {
  if (n.type == N_START) return {true, false};
  if (n.type == N_END) return {false, true};
  if (n.type == N_GROUP) {
    bool s = false, e = false;
    for (int i = 0; i < a.nkids(n); ++i) {
      auto r = anchors_rec(a, a.child(n, i));
      s = s || r.first; e = e || r.second;
    }
    return {s, e};
  }
  return {false, false};
}

// ---- literal extraction (literal_optimizer.mojo:241-447) ----------------------
void add_lit(LiteralSet& ls, std::string lit, int off, bool prefix, bool required) {
  LiteralInfo li;
  li.literal = std::move(lit); li.start_offset = off; li.is_prefix = prefix;
  li.is_suffix = false; li.is_required = required;
  ls.literals.push_back(std::move(li));
}

void extract_sequence(const Ast& a, const Node& g, int start_off, bool required, bool at_start,
                      LiteralSet& ls) {
  std::string cur;
  int off = start_off;
  bool seq_start = at_start;
  for (int i = 0; i < a.nkids(g); ++i) {
    const Node& c = a.child(g, i);
    if (c.type == N_ELEMENT && c.min == 1 && c.max == 1 && a.has_value(c)) {
      cur += a.value(c);
    } else {
      if (!cur.empty()) {
        add_lit(ls, cur, off, seq_start, required);
        off += (int)cur.size();
        seq_start = false;
        cur.clear();
      }
      if (c.type == N_START || c.type == N_END) continue;
      seq_start = false;
      if (c.min > 0) off += 1;
    }
  }
  if (!cur.empty()) add_lit(ls, cur, off, seq_start, required);
}

std::string prefix_literal(const Ast& a, const Node& n) {
  if (n.type == N_ELEMENT && n.min >= 1 && n.max >= 1 && a.has_value(n))
    return std::string(a.value(n));
  if (n.type == N_GROUP && n.min >= 1) {
    LiteralSet tmp;
    extract_sequence(a, n, 0, true, true, tmp);
    if (!tmp.literals.empty()) return tmp.literals[0].literal;
  }
  return "";
}

void collect_or_prefixes(const Ast& a, const Node& n, std::vector<std::string>& out) {
  if (n.type != N_OR) {
    std::string p = prefix_literal(a, n);
    if (!p.empty()) out.push_back(p);
    return;
  }
  for (int i = 0; i < a.nkids(n); ++i) collect_or_prefixes(a, a.child(n, i), out);
}

std::string or_common_prefix(const Ast& a, const Node& o) {
  std::vector<std::string> pre;
  collect_or_prefixes(a, o, pre);
  if (pre.size() < 2) return "";
  std::string common = pre[0];
  for (size_t i = 1; i < pre.size(); ++i) {
    size_t k = 0;
    while (k < common.size() && k < pre[i].size() && common[k] == pre[i][k]) ++k;
    common.resize(k);
    if (common.empty()) return "";
  }
  return common;
}

void extract_from(const Ast& a, const Node& n, LiteralSet& ls, int off, bool required,
                  bool at_start) {
  if (n.type == N_ELEMENT) {
    if (n.min >= 1 && a.has_value(n)) {
      if (n.max == 1 || n.max == -1)
        add_lit(ls, std::string(a.value(n)), off, at_start, required);
    }
  } else if (n.type == N_GROUP) {
    if (n.min >= 1) {
      if (a.nkids(n) == 1) {
        const Node& c = a.child(n, 0);
        if (c.type == N_GROUP || c.type == N_OR) {
          extract_from(a, c, ls, off, required, at_start);
          return;
        }
      }
      extract_sequence(a, n, off, required, at_start, ls);
    }
  } else if (n.type == N_OR) {
    std::string cp = or_common_prefix(a, n);
    if (!cp.empty()) add_lit(ls, cp, off, at_start, true);
    for (int i = 0; i < a.nkids(n); ++i) extract_from(a, a.child(n, i), ls, off, false, at_start);
  }
}

bool literal_prefix_node(const Ast& a, const Node& n) {  // literal_optimizer.mojo:479-497
  if (n.type == N_START) return false;
  if (n.type == N_ELEMENT) return n.min >= 1 && n.max >= 1;
  if (n.type == N_GROUP) {
    if (n.min >= 1 && a.nkids(n) > 0) {
      const Node& f = a.child(n, 0);
      if (f.type == N_START && a.nkids(n) > 1) return literal_prefix_node(a, a.child(n, 1));
      return literal_prefix_node(a, f);
    }
    return false;
  }
  return false;
}

}  // namespace

Complexity classify(const Ast& a) { return Analyzer{a}.node(a.root, 0); }

int count_simd_nodes(const Ast& a, const Node& n) {  // optimizer.mojo:234-258
  int c = 0;
  if (is_class(n.type)) {
    c += (n.min > 1 || n.max == -1) ? 2 : 1;
  } else if (n.type == N_GROUP || n.type == N_RE || n.type == N_OR) {
    for (int i = 0; i < a.nkids(n); ++i) c += count_simd_nodes(a, a.child(n, i));
  }
  return c;
}

bool should_use_pure_dfa(const Ast& a) {  // optimizer.mojo:174-201
  if (classify(a) != CX_SIMPLE) return false;
  return count_simd_nodes(a, a.root) <= 1;
}

bool is_literal_pattern(const Ast& a) {  // optimizer.mojo:848-863
  if (a.root.type != N_RE) return false;
  if (a.nkids(a.root) == 0) return true;
  return literal_sequence(a, a.child(a.root, 0));
}

std::string get_literal_string(const Ast& a) {  // optimizer.mojo:903-918
  if (a.root.type == N_RE && a.nkids(a.root) > 0) return literal_chars(a, a.child(a.root, 0));
  return "";
}

std::pair<bool, bool> pattern_has_anchors(const Ast& a) {  // optimizer.mojo:956-972
  if (a.root.type == N_RE && a.nkids(a.root) > 0) return anchors_rec(a, a.child(a.root, 0));
  return {false, false};
}

std::string common_prefix(const std::vector<std::string>& br) {
  // optimizer.mojo:675-710 (= dfa.mojo:1465-1498, 3489-3522)
  if (br.empty()) return "";
  if (br.size() == 1) return br[0];
  size_t mn = br[0].size();
  for (const auto& b : br) mn = std::min(mn, b.size());
  std::string out;
  for (size_t p = 0; p < mn; ++p) {
    const char c = br[0][p];
    bool all = true;
    for (size_t i = 1; i < br.size(); ++i)
      if (br[i][p] != c) { all = false; break; }
    if (!all) break;
    out.push_back(c);
  }
  return out;
}

LiteralSet extract_literals(const Ast& a) {  // literal_optimizer.mojo:217-238, 166-206
  LiteralSet ls;
  if (a.root.type == N_RE && a.nkids(a.root) > 0)
    extract_from(a, a.child(a.root, 0), ls, 0, true, true);
  if (!ls.literals.empty()) {
    int best = 0, best_score = 0;
    for (int i = 0; i < (int)ls.literals.size(); ++i) {
      const LiteralInfo& l = ls.literals[i];
      int score = 0;
      if (l.is_required) score += 1000;
      score += (int)l.literal.size() * 10;
      if (l.is_prefix) score += 100;
      if (l.is_suffix) score += 100;
      score += l.start_offset;
      if (score > best_score) { best_score = score; best = i; }
    }
    ls.best = best;
  }
  return ls;
}

bool has_literal_prefix(const Ast& a) {  // literal_optimizer.mojo:463-476
  if (a.root.type != N_RE || a.nkids(a.root) == 0) return false;
  return literal_prefix_node(a, a.child(a.root, 0));
}

}  // namespace mrx
