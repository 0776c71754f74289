// Lexer + parser (see mrx_ast.hpp for the reference citations).
#include "mrx_ast.hpp"

namespace mrx {

std::vector<Token> scan(const std::string& re) {
  // src/regex/lexer.mojo:61-195
  std::vector<Token> out;
  out.reserve(re.size());
  const int n = (int)re.size();
  bool esc = false;
  for (int i = 0; i < n; ++i) {
    const int ch = (unsigned char)re[i];
    if (esc) {
      // lexer.mojo:78-100: \t -> TAB element, \s \d \w classes, \c -> element c
      switch (ch) {
        case 't': out.push_back({TK_ELEMENT, 9, i - 1}); break;
        case 's': out.push_back({TK_SPACE, ch, i - 1}); break;
        case 'd': out.push_back({TK_DIGIT, ch, i - 1}); break;
        case 'w': out.push_back({TK_WORD, ch, i - 1}); break;
        default: out.push_back({TK_ELEMENT, ch, i}); break;
      }
      esc = false;
      continue;
    }
    switch (ch) {
      case '\\': esc = true; continue;
      case '.': out.push_back({TK_WILDCARD, ch, i}); break;
      case '(': out.push_back({TK_LPAREN, ch, i}); break;
      case ')': out.push_back({TK_RPAREN, ch, i}); break;
      case '[': out.push_back({TK_LBRACKET, ch, i}); break;
      case ']': out.push_back({TK_RBRACKET, ch, i}); break;
      case '-': out.push_back({TK_DASH, ch, i}); break;
      case '$': out.push_back({TK_END, ch, i}); break;
      case '?': out.push_back({TK_QMARK, ch, i}); break;
      case '*': out.push_back({TK_ASTERISK, ch, i}); break;
      case '+': out.push_back({TK_PLUS, ch, i}); break;
      case '|': out.push_back({TK_VBAR, ch, i}); break;
      case '}': out.push_back({TK_RCURLY, ch, i}); break;
      case '^': out.push_back({i == 0 ? TK_START : TK_CIRCUMFLEX, ch, i}); break;
      case '{': {
        // lexer.mojo:129-153: only digits, ',' and '}' may follow
        out.push_back({TK_LCURLY, ch, i});
        ++i;
        for (; i < n; ++i) {
          const int c2 = (unsigned char)re[i];
          if (c2 == ',') out.push_back({TK_COMMA, c2, i});
          else if (c2 >= '0' && c2 <= '9') out.push_back({TK_ELEMENT, c2, i});
          else if (c2 == '}') { out.push_back({TK_RCURLY, c2, i}); break; }
          else throw SyntaxError("Bad token at index " + std::to_string(i) + ".{");
        }
        break;
      }
      default: out.push_back({TK_ELEMENT, ch, i}); break;
    }
  }
  return out;
}

namespace {

struct Parser {
  Ast& ast;
  int group_counter = 0;
  const int plen;
  explicit Parser(Ast& a) : ast(a), plen((int)a.pattern.size()) {}

  static Node leaf(NodeType t, int s, int e, bool positive = true) {
    Node n;
    n.type = t; n.start_idx = s; n.end_idx = e; n.min = 1; n.max = 1; n.positive = positive;
    return n;
  }
  static Node group(std::vector<uint16_t> kids, int s, int e, bool cap, int gid) {
    Node n;
    n.type = N_GROUP; n.start_idx = s; n.end_idx = e; n.capturing = cap; n.kids = std::move(kids);
    n.min = 1; n.max = 1; n.group_id = gid;
    return n;
  }

  // parser.mojo:56-114; tokens[i+1] exists (checked by the caller)
  static void quantifier(int& i, Node& el, const std::vector<Token>& t) {
    const int nt = (int)t.size();
    const TokType nx = t[i + 1].type;
    if (nx == TK_ASTERISK) { el.min = 0; el.max = -1; ++i; }
    else if (nx == TK_PLUS) { el.min = 1; el.max = -1; ++i; }
    else if (nx == TK_QMARK) { el.min = 0; el.max = 1; ++i; }
    else if (nx == TK_LCURLY) {
      i += 2;
      long mn = 0, mx = 0;
      bool has_mn = false, has_mx = false;
      while (i < nt && t[i].type == TK_ELEMENT) {
        const int d = t[i].ch;
        if (d < '0' || d > '9') throw SyntaxError("Invalid digit in quantifier");
        mn = mn * 10 + (d - '0'); has_mn = true; ++i;
      }
      el.min = has_mn ? (int)mn : 0;
      if (i < nt && t[i].type == TK_COMMA) {
        ++i;
        while (i < nt && t[i].type == TK_ELEMENT) {
          const int d = t[i].ch;
          if (d < '0' || d > '9') throw SyntaxError("Invalid digit in quantifier");
          mx = mx * 10 + (d - '0'); has_mx = true; ++i;
        }
        el.max = has_mx ? (int)mx : -1;
      } else {
        el.max = el.min;
      }
      if (i < nt && t[i].type == TK_RCURLY) ++i;
      --i;
    }
  }

  // parser.mojo:130-464
  Node parse_list(const std::vector<Token>& t) {
    const int nt = (int)t.size();
    if (nt == 0) return group({}, 0, 0, true, 0);

    // first top-level '|' splits the list (right-leaning binary OR tree)
    int depth = 0;
    for (int k = 0; k < nt; ++k) {
      if (t[k].type == TK_LPAREN) ++depth;
      else if (t[k].type == TK_RPAREN) --depth;
      else if (t[k].type == TK_VBAR && depth == 0) {
        std::vector<Token> lt(t.begin(), t.begin() + k), rt(t.begin() + k + 1, t.end());
        Node l = lt.empty() ? group({}, 0, 0, true, 0) : parse_list(lt);
        Node r = rt.empty() ? group({}, 0, 0, true, 0) : parse_list(rt);
        const int li = ast.add(l);
        const int ri = ast.add(r);
        Node o;
        o.type = N_OR; o.start_idx = 0; o.end_idx = plen; o.min = 1; o.max = 1;
        o.kids = {(uint16_t)li, (uint16_t)ri};
        return o;
      }
    }

    // parser.mojo:216-237
    int bd = 0, pd = 0;
    for (const Token& tk : t) {
      if (tk.type == TK_LBRACKET) ++bd;
      else if (tk.type == TK_RBRACKET) {
        if (--bd < 0)
          throw SyntaxError("Unescaped closing bracket ']' at position " + std::to_string(tk.pos));
      } else if (tk.type == TK_LPAREN) ++pd;
      else if (tk.type == TK_RPAREN) {
        if (--pd < 0)
          throw SyntaxError("Unescaped closing parenthesis ')' at position " +
                            std::to_string(tk.pos));
      }
    }

    std::vector<Node> elems;
    for (int i = 0; i < nt; ++i) {
      const Token& tk = t[i];
      switch (tk.type) {
        case TK_ELEMENT:
        case TK_DASH: {  // a dash outside brackets is a literal '-'
          Node e = leaf(N_ELEMENT, tk.pos, tk.pos + 1);
          if (i + 1 < nt) quantifier(i, e, t);
          elems.push_back(e);
          break;
        }
        case TK_WILDCARD: {
          Node e = leaf(N_WILDCARD, tk.pos, tk.pos + 1);
          if (i + 1 < nt) quantifier(i, e, t);
          elems.push_back(e);
          break;
        }
        case TK_SPACE:
        case TK_DIGIT:
        case TK_WORD: {
          const NodeType k = tk.type == TK_SPACE ? N_SPACE : tk.type == TK_DIGIT ? N_DIGIT : N_WORD;
          Node e = leaf(k, tk.pos, tk.pos + 2);
          if (i + 1 < nt) quantifier(i, e, t);
          elems.push_back(e);
          break;
        }
        case TK_START: elems.push_back(leaf(N_START, tk.pos, tk.pos + 1)); break;
        case TK_END: elems.push_back(leaf(N_END, tk.pos, tk.pos + 1)); break;
        case TK_LBRACKET: {
          // parser.mojo:313-354: the RANGE node keeps the raw "[...]" slice
          const int bs = tk.pos;
          ++i;
          bool positive = true;
          if (i < nt && t[i].type == TK_CIRCUMFLEX) { positive = false; ++i; }
          while (i < nt && t[i].type != TK_RBRACKET) {
            if (i + 2 < nt && t[i + 1].type == TK_DASH && t[i + 2].type == TK_ELEMENT) i += 3;
            else ++i;
          }
          if (i >= nt) throw SyntaxError("Missing closing ']'.");
          Node e = leaf(N_RANGE, bs, t[i].pos + 1, positive);
          if (i + 1 < nt) quantifier(i, e, t);
          elems.push_back(e);
          break;
        }
        case TK_LPAREN: {
          // parser.mojo:365-444
          const int ps = tk.pos;
          ++i;
          bool cap = true;
          int content = ps + 1;
          if (i + 1 < nt && t[i].type == TK_QMARK && t[i + 1].type == TK_ELEMENT &&
              t[i + 1].ch == ':') {
            cap = false; i += 2; content = ps + 3;
          }
          std::vector<Token> inner;
          int pc = 1;
          while (i < nt && pc > 0) {
            if (t[i].type == TK_LPAREN) ++pc;
            else if (t[i].type == TK_RPAREN) { if (--pc == 0) break; }
            inner.push_back(t[i]);
            ++i;
          }
          if (pc > 0) throw SyntaxError("Missing closing parenthesis ')'.");
          const int pe = t[i].pos;
          int gid = -1;
          if (cap) gid = ++group_counter;
          Node g = parse_list(inner);
          if (g.type == N_GROUP) {
            g.capturing = cap; g.group_id = gid; g.start_idx = content; g.end_idx = pe;
          } else {
            const int ci = ast.add(g);
            g = group({(uint16_t)ci}, content, pe, cap, gid);
          }
          if (i + 1 < nt) quantifier(i, g, t);
          elems.push_back(g);
          break;
        }
        default: break;  // stray quantifiers, ',', '}', '^' ... are skipped silently
      }
    }
    std::vector<uint16_t> kids;
    kids.reserve(elems.size());
    for (const Node& e : elems) kids.push_back((uint16_t)ast.add(e));
    return group(std::move(kids), 0, plen, true, 0);
  }
};

}  // namespace

void parse(const std::string& pattern, Ast& ast) {
  // parser.mojo:467-510
  ast.pattern = pattern;
  ast.arena.clear();
  Parser p(ast);
  Node body = p.parse_list(scan(pattern));
  const int id = ast.add(body);
  Node root;
  root.type = N_RE; root.start_idx = 0; root.end_idx = (int)pattern.size();
  root.kids = {(uint16_t)id};
  ast.root = root;
}

}  // namespace mrx
