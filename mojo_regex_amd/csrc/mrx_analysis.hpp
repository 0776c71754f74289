// Pattern analysis used by the router: complexity classes, literal helpers and
// literal extraction.  Behaviour mirrors src/regex/optimizer.mojo:103-999 and
// src/regex/literal_optimizer.mojo:24-517 of the reference.
#pragma once
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "mrx_ast.hpp"

namespace mrx {

enum Complexity { CX_SIMPLE = 0, CX_MEDIUM = 1, CX_COMPLEX = 2 };

Complexity classify(const Ast& a);
int count_simd_nodes(const Ast& a, const Node& n);
bool should_use_pure_dfa(const Ast& a);

bool is_literal_pattern(const Ast& a);
std::string get_literal_string(const Ast& a);
std::pair<bool, bool> pattern_has_anchors(const Ast& a);
std::string common_prefix(const std::vector<std::string>& branches);

struct LiteralInfo {
  std::string literal;
  int start_offset = 0;
  bool is_prefix = false, is_suffix = false, is_required = true;
};
struct LiteralSet {
  std::vector<LiteralInfo> literals;
  int best = -1;
  const LiteralInfo* best_literal() const { return best >= 0 ? &literals[best] : nullptr; }
};
LiteralSet extract_literals(const Ast& a);
bool has_literal_prefix(const Ast& a);

}  // namespace mrx
