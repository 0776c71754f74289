// Host-side engine tables: the DFAEngine-equivalent table set and the PikeVM
// program with its eagerly determinised LazyDFA.
//
// Reference behaviour mirrored:
//   DFA shape dispatcher + compilers   src/regex/dfa.mojo:308-1803, 2385-3589
//   byte-class matcher metadata         src/regex/simd_ops.mojo:63-134, 261-420
//   PikeVM bytecode                     src/regex/pikevm.mojo:39-333
//   first-byte filter / LazyDFA         src/regex/pikevm.mojo:367-416, 664-987
// The LazyDFA of the reference memoises PikeVM state sets on demand; here the
// reachable sets are enumerated once at compile time (same sets, same
// leftmost-start / longest-end results; see DESIGN.md "LazyDFA").
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "mrx_ast.hpp"

namespace mrx {

struct DfaCompileError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct ClassMatcher {
  std::array<uint8_t, 256> lookup{};  // 0/1
  int num_ranges = 0;                  // 0 => reference uses the nibble-table scan
  std::array<uint8_t, 16> lo_tbl{}, hi_tbl{};
  void build(const std::string& char_class);
  void finish();  // detect ranges + nibble tables from lookup
  bool nibble_hit(int c) const { return (lo_tbl[c & 15] & hi_tbl[(c >> 4) & 15]) != 0; }
};

struct DfaEngine {
  std::vector<std::array<int16_t, 256>> trans;  // -1 = no transition
  std::vector<uint8_t> accepting;
  bool has_start_anchor = false, has_end_anchor = false;
  bool is_pure_literal = false;
  bool has_matcher = false, scan_eligible = false;
  ClassMatcher matcher;
  std::string literal;
  std::string shape;  // which dispatcher branch fired

  int nstates() const { return (int)trans.size(); }
  int add_state(bool acc = false) {
    std::array<int16_t, 256> row;
    row.fill(-1);
    trans.push_back(row);
    accepting.push_back(acc ? 1 : 0);
    return (int)trans.size() - 1;
  }
};

// throws DfaCompileError where the reference's compile_dfa_pattern raises
void compile_dfa_pattern(const Ast& a, DfaEngine& out);

// ---- PikeVM ------------------------------------------------------------------
enum Op : uint8_t {
  OP_BYTE, OP_RANGE, OP_CLASS, OP_ANY, OP_SPLIT, OP_JUMP, OP_MATCH, OP_START_ANCHOR,
  OP_END_ANCHOR
};
struct Inst {
  Op op;
  int a0, a1;
};
struct Program {
  std::vector<Inst> insts;
  std::vector<std::array<uint8_t, 256>> classes;
  bool has_end_anchor() const {
    for (const auto& i : insts)
      if (i.op == OP_END_ANCHOR) return true;
    return false;
  }
};
constexpr int kPikeMaxStates = 512;  // pikevm.mojo:342

void compile_program(const Ast& a, Program& p);

struct LazyTables {
  bool supported = false;       // program <= 512 instructions
  bool has_filter = false;      // first-byte filter usable (pikevm.mojo:416)
  std::array<uint8_t, 256> first_byte{};
  std::array<uint8_t, 16> lo_tbl{}, hi_tbl{};
  bool start_dead = false;      // start closure empty (LAZY_DFA_DEAD)
  // determinised automaton: state 0 = start set
  std::vector<std::array<int32_t, 256>> trans;  // -1 dead
  std::vector<uint8_t> is_match;
  bool too_large = false;       // determinisation exceeded the state budget
  // '$' programs: the transition as computed while the text's last byte is consumed ('$' holds in its closure)
  bool has_end_variant = false;
  std::vector<std::array<int32_t, 256>> trans_end;
};
void build_lazy(const Program& p, LazyTables& out, int max_dfa_states);

// OnePass NFA (src/regex/onepass.mojo:180-355): subset construction that rejects programs in
// which one byte can fire two positions with different follow-up closures.  The reference
// consults it only for NFAMatcher.match_first on '$' programs (matcher.mojo:310-313, 378-379).
struct OnePassTables {
  bool ok = false;                              // compiled (one-pass, <= 512 states)
  bool has_start_anchor = false, has_end_anchor = false;
  std::vector<std::array<int16_t, 256>> trans;  // -1 dead
  std::vector<uint8_t> is_match, is_end_match;
};
void build_onepass(const Program& p, OnePassTables& out);

// Bit-parallel form of the same PikeVM program (the "bitset NFA"): one bit per
// non-epsilon instruction ("position": BYTE / CLASS / ANY / RANGE / MATCH).  A state set
// is kBitsetWords x 64 bits; one byte step is
//     next = OR over positions i in (set & byte_mask[byte]) of follow[i]
// with follow[i] = epsilon closure of pc_i + 1 (pikevm.mojo:604-648, not at text start).
// It yields exactly the sets LazyDFA._compute_transition determinises (pikevm.mojo:870-942),
// without enumerating them -- so it also serves programs whose DFA exceeds any budget.
constexpr int kBitsetWords = 4;  // up to 256 positions
struct BitsetNfa {
  bool ok = false;  // program fits (<= 256 positions) and has no '$'
  int npos = 0, nw = 0;
  std::vector<int> pos_pc;
  std::array<uint64_t, kBitsetWords> start{}, match{};
  std::vector<std::array<uint64_t, kBitsetWords>> follow;     // [npos]
  std::array<std::array<uint64_t, kBitsetWords>, 256> byte_mask{};
};
void build_bitset(const Program& p, BitsetNfa& out);

// ---- the backtracking matcher as a flat program (capture groups) ----------------------
// NFAEngine._match_node (nfa.mojo:657-1731) is a recursive walk of the AST.  For ASTs without
// alternation and without quantified groups its control flow is a linear program over the
// leaves with three kinds of events -- which is what the GPU interprets (bt_match_at in
// mrx_device.hpp):
//   LEAF   one byte class with {min, max}.  Not the last child of its sequence and quantified:
//          a CHOICE (nfa.mojo:1231-1311, counts from min(max, run) down to min, the rest of the
//          sequence decides); otherwise one greedy run (_apply_quantifier, nfa.mojo:1375-1443).
//   OPEN / CLOSE of a group.  CLOSE drops the choices made inside the group (a group's sequence
//          returns its first success and is never re-entered, nfa.mojo:1082-1103) and records the
//          group's span when it captures (appended, never rolled back: the last one wins,
//          matcher.mojo:1797-1802).
//   START / END anchors.
// Every leaf carries three membership tables because the reference tests bytes three ways: the
// leaf matcher's own first-byte test (nfa.mojo:757-995), ASTNode.is_match_char (ast.mojo:415-462;
// choices and short runs) and the cached "SIMD" matchers of long runs (nfa.mojo:1446-1647), and
// they differ (\s: the nibble tables also hold NUL and ")*+,-"; [a-z0-9]-style classes in long
// runs are searched as plain strings).
//   ALT / ALT_END / ALT_CLOSE: `A|B` (NFAEngine._match_or, nfa.mojo:1019-1055: the left branch's first local
//          success is returned, else the right branch's; neither is re-entered afterwards).  ALT marks the
//          choice stack (min = first item of B, max = first item behind the alternation), ALT_END closes A
//          (drops A's choices and the mark, jumps to max), ALT_CLOSE closes B.
//   LOOP / LOOP_END: a quantified group that is the LAST child of its sequence
//          (_match_group_with_quantifier, nfa.mojo:1105-1156): the body is matched greedily up to max times,
//          each repetition's first local success is kept, the first failing repetition ends the loop, at
//          least min repetitions are required; every repetition records the span from the group's FIRST
//          start.  (A quantified group elsewhere in a sequence goes through _match_with_backtracking, which
//          asks ASTNode.is_match_char of the group node -- false -- so it matches zero times when min == 0
//          and not at all otherwise: nothing is emitted, or FAIL.)
//   FAIL:  never matches.
enum BtKind : uint8_t { BT_LEAF = 0, BT_OPEN = 1, BT_CLOSE = 2, BT_START = 3, BT_END = 4,
                        BT_ALT = 5, BT_ALT_END = 6, BT_ALT_CLOSE = 7, BT_LOOP = 8, BT_LOOP_END = 9, BT_FAIL = 10 };
enum BtFlags : uint8_t {
  BTF_LAST = 1,        // leaf: last child of its sequence
  BTF_ZERO_OK = 2,     // \d, \w: zero repetitions on a non-matching byte / at the end when min == 0
  BTF_SIMD_TYPE = 4,   // \s \d \w [..]: long runs use the cached matcher's set
  BTF_RANGE_LONG = 8,  // [..] whose text is longer than 8 bytes: always the cached matcher's set
  BTF_CAPTURING = 16,  // OPEN / CLOSE
  BTF_QUANT = 32       // min != 1 or max != 1
};
struct BtItem {   // 12 bytes, device layout
  uint8_t kind, flags;
  int8_t gid;     // OPEN / CLOSE: group id (0 = the implicit root group)
  uint8_t tbl;    // LEAF: its tables are tbl * 3 + {0 first byte, 1 is_match_char, 2 cached matcher}
  int32_t min, max;
};
struct BtProg {
  bool ok = false;
  std::string why_not;
  std::vector<BtItem> items;
  std::vector<std::array<uint8_t, 32>> tables;   // 256-bit membership bitmaps
  int ngroups = 0;                                  // highest capturing group id (1..9 are addressable)
  int max_depth = 0;
};
void build_bt(const Ast& a, BtProg& out);

}  // namespace mrx
