// Host-side pattern front end of libmrx_hip: pattern bytes -> tokens -> AST.
//
// The compiled tables the GPU kernels walk are a pure function of the pattern,
// and bit-exact results require the reference's exact (quirky) construction, so
// the front end mirrors the reference's behaviour:
//   tokens   src/regex/lexer.mojo:61-195, src/regex/tokens.mojo:44-103
//   parser   src/regex/parser.mojo:56-114 (quantifiers), :130-464, :467-510
//   AST      src/regex/ast.mojo:168-558 (arena of value nodes, 1-based child ids,
//            node value = raw pattern slice)
// It is an independent C++ implementation (flat arena of PODs, string_view
// values); the oracle under oracle/ is a separate Python restatement and the two
// are cross-checked table-for-table by tests/test_host_tables.py.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

namespace mrx {

struct SyntaxError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

enum TokType : uint8_t {
  TK_ELEMENT, TK_WILDCARD, TK_SPACE, TK_DIGIT, TK_WORD, TK_START, TK_END, TK_COMMA,
  TK_LPAREN, TK_RPAREN, TK_LCURLY, TK_RCURLY, TK_LBRACKET, TK_RBRACKET, TK_ASTERISK,
  TK_PLUS, TK_QMARK, TK_VBAR, TK_CIRCUMFLEX, TK_DASH
};

struct Token {
  TokType type;
  int ch;
  int pos;  // start_pos in the pattern
};

enum NodeType : uint8_t {
  N_RE = 0, N_ELEMENT = 1, N_WILDCARD = 2, N_SPACE = 3, N_DIGIT = 4, N_WORD = 5,
  N_RANGE = 6, N_START = 7, N_END = 8, N_OR = 9, N_NOT = 10, N_GROUP = 11
};

struct Ast;

struct Node {
  NodeType type = N_RE;
  int start_idx = 0, end_idx = 0;
  bool capturing = false;
  std::vector<uint16_t> kids;  // 1-based indices into Ast::arena
  int min = 0, max = 0;
  bool positive = true;
  int group_id = -1;
};

struct Ast {
  std::string pattern;
  std::vector<Node> arena;
  Node root;

  int add(const Node& n) {  // returns the 1-based id (ast.mojo:156-165)
    arena.push_back(n);
    return (int)arena.size();
  }
  const Node& child(const Node& n, int i) const { return arena[n.kids[i] - 1]; }
  int nkids(const Node& n) const { return (int)n.kids.size(); }
  // ast.mojo:546-558: raw pattern slice, empty view when start == end
  std::string_view value(const Node& n) const {
    if (n.start_idx == n.end_idx) return std::string_view();
    return std::string_view(pattern).substr(n.start_idx, n.end_idx - n.start_idx);
  }
  bool has_value(const Node& n) const { return n.start_idx != n.end_idx; }
};

std::vector<Token> scan(const std::string& pattern);
// Fills ast (pattern, arena, root); throws SyntaxError with the reference's text.
void parse(const std::string& pattern, Ast& ast);

}  // namespace mrx
