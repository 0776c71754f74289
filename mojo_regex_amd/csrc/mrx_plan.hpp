// HostPlan: everything mrx_compile() derives from a pattern -- the reference's
// routing decision (HybridMatcher.__init__, src/regex/matcher.mojo:566-693;
// NFAMatcher, :273-431; NFAEngine flags, src/regex/nfa.mojo:86-143;
// CompiledRegex fixed-width groups, matcher.mojo:1002-1035) plus the flat,
// byte-class-compressed table blob the kernels stage into LDS.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "mrx_analysis.hpp"
#include "mrx_engines.hpp"

namespace mrx {

// ---- device-visible plan (POD, passed to kernels by value) --------------------
enum PlanKind : int32_t {
  PLAN_DFA = 0,   // DFAEngine semantics (src/regex/dfa.mojo:1815-2253)
  PLAN_LAZY = 1,  // LazyDFA semantics   (src/regex/pikevm.mojo:754-867)
  PLAN_ANY = 2    // pattern ".*"        (matcher.mojo:740-745, 762-766, 809-813)
};

enum PlanFlags : uint32_t {
  PF_START_ANCHOR = 1u << 0,
  PF_END_ANCHOR = 1u << 1,
  PF_PURE_LITERAL = 1u << 2,    // DFAEngine.is_pure_literal
  PF_HAS_MATCHER = 1u << 3,     // DFAEngine._has_simd_matcher / LazyDFA has_filter
  PF_SCAN_ELIGIBLE = 1u << 4,   // DFAEngine._simd_scan_eligible
  PF_START_ACCEPTING = 1u << 5,
  PF_EXACT_LITERAL = 1u << 6,   // HybridMatcher.is_exact_literal && !has_anchors
  PF_PREFILTER = 1u << 7,       // HybridMatcher.prefilter && !has_anchors
  PF_START_DEAD = 1u << 8,      // LazyDFA start state is LAZY_DFA_DEAD
  PF_STREAMABLE = 1u << 9,      // findall can run on the single-pass streaming kernel
  PF_BITSET = 1u << 10,         // PLAN_LAZY walks run on the bitset NFA instead of the DFA table
  PF_STEP_SEARCH = 1u << 13,    // match_next may run on the windowed stepper (plain table walk)
  PF_STEP_REQ = 1u << 14,       // findall / count: the required-byte route on the windowed stepper
  PF_STEP_BIG = 1u << 15,       // stepper plan whose byte-indexed table does not fit LDS: only the
                                // wavefront-per-text kernel (class-indexed table) runs it
  PF_STEPPABLE = 1u << 12,      // plain restart-per-position route (no anchors, literals, shortcuts, empty
                                // matches): findall / search may run on the flattened lane-per-text kernel
  PF_BSTEP = 1u << 16,          // PF_BITSET plan of at most 64 positions on the windowed stepper's plain route:
                                // the state is the set of live positions, a step is mask + follow-table lookups
                                // in LDS (k_wstep<., 0, 1>); PF_STEP_SEARCH / PF_STEPPABLE are set with it
  PF_BT_FIRST = 1u << 17,       // match_first / is_match: NFAMatcher falls through to NFAEngine.match_first
                                // (matcher.mojo:380) -- the backtracking matcher, run as its flat program
  PF_BT_SEARCH = 1u << 18,      // match_next / match_all: NFAEngine.match_next / match_all (matcher.mojo:419, 431)
  PF_STEP_EMPTY = 1u << 19,     // findall / count of a table plan whose start state accepts and that has no
                                // first-byte matcher (every position yields a match, possibly empty): the
                                // windowed stepper's plain route in its EMPTY form (k_wstep<., 0, 0, 1>)
  PF_MWALK = 1u << 20,          // findall / count / search of a plain-route table plan in ONE left-to-right pass over
                                // several simultaneous walks (DevPlan::off_mw_*; k_mwalk): what the single-walk proof
                                // of the streaming kernel rejects ("accepting state continues into a non-accepting
                                // one", "a later start survives ...") without the stepper's re-scans
  PF_MWALK_REQ = 1u << 21,      // findall / count of a required-byte plan (PF_STEP_REQ: HybridMatcher._match_all_required_byte,
                                // matcher.mojo:864-898) in one pass on the same kernel (DevPlan::off_mwr_*)
  PF_BACKSET = 1u << 22,        // plain-route stepper plan with a BACKWARD table (DevPlan::off_bk_*): a right-to-left pass
                                // marks the positions at which a match begins, the stepper then only starts walks
                                // that succeed (k_backscan + k_wstep<., 0, 0, 0, 1>)
  PF_LAZY_END = 1u << 23,       // '$' program on the LazyDFA search: transition rows nstates/2.. are the "computed at the
                                // text's end" variants, chosen per text as the lazy cache would have them (generic kernels only)
  PF_MW_EMPTY = 1u << 24,       // PF_STEP_EMPTY plan whose walks never read beyond their last accepting position (every
                                // reachable state accepts): findall / count in one pass on k_mwalk, one walk, up to two
                                // reports per byte -- the match that a byte ends, and the empty match at that byte when
                                // no walk can begin on it (DevPlan::off_mw_*; build_emptywalk())
  PF_MW_TRIES = 1u << 25,       // plain-route table plan that fails the multi-walk proofs but whose walks read at most seven
                                // bytes beyond their last accepting position: findall / count in one pass with the tries the
                                // reference may come back to kept beside the oldest walk (build_emptywalk2(., empty = false);
                                // DevPlan::off_mw_* with mw_k == -3; k_mwalk<., 2, 0, 3>); search = the first report of the same walk.
  PF_STREAM_SEARCH = 1u << 11   // search / sub / captures may use the streaming kernel too (findall and
                                // count may whenever PF_STREAMABLE is set): not with a memchr prefilter,
                                // which only match_next consults (matcher.mojo:784-796)
};

constexpr int kMaxTemplateSegs = 32;

struct DevPlan {
  int32_t kind;
  uint32_t flags;
  int32_t nstates, ncls;
  int32_t lit_len;        // engine literal (pure literal) or exact literal length
  int32_t pre_len;        // MemchrPrefilter literal length (PF_PREFILTER)
  int32_t required_byte;  // -1 if none (matcher.mojo:684-693)
  // byte offsets into the table blob
  int32_t off_cls, off_first, off_trans, off_lit, off_pre, blob_bytes;
  // fixed-width capture groups (matcher.mojo:1002-1035)
  int32_t fixed_total, fixed_ngroups, fixed_concat;
  int32_t fixed_off[10], fixed_w[10];
  // streaming (single-pass search automaton), see mrx_kernels.hip
  int32_t st_nstates;     // live states of the search automaton (0 = idle)
  int32_t st_kind;        // 0 none, 1 = byte-column form (<= 4 states), 2 = class-table form,
                          // 3 = wide byte-column form (<= 8 states)
  int32_t st_fixed_len;   // > 0: every match is [end - st_fixed_len, end) (exact-literal KMP automaton)
  int32_t off_stcol;      // kind 1: u16 column table [256] (4 states x 4 bit); kind 3: u64 [256] (8 x 8 bit)
  uint32_t st_accept_mask;
  // kind 1, "code columns" (off_stcol32 >= 0; round 3): u16 column table [256] (idle + at most two states).  State q owns the 5-bit field at
  // bit offset st_code_off(q) (the idle state at 0); the field holds the bit offset of the next state's field,
  // so one step is `s = col[byte] >> s` (the hardware takes the shift count modulo 32).  Offsets are chosen
  // with offset mod 4 = the state's 2-bit CODE: bit 1 = "accepting", bit 0 = "a walk's first state(s)", and the
  // plan qualifies only if for every transition EMIT == acc(old) && !acc(new) and NEWSTART == first(new) &&
  // !first(old) -- so the kernel records the low two bits of s per byte (`alignbit`) and derives the event word
  // of a 16-byte group from that code word and the one before it with two instructions.  st_acc32: bit o set
  // when the state whose field sits at offset o accepts (the end-of-text rule).
  int32_t off_stcol32;
  uint32_t st_acc32;
  // kind 2: cls[256] u8, trans[st_nstates][1 << st_cshift] u16 = (next << st_cshift) << 2 | EMIT << 1 | NEWSTART,
  // accept[st_nstates] u8
  int32_t off_stg_cls, off_stg_trans, off_stg_acc, st_cshift, stg_bytes;
  // kind 2 with a reset byte and a small class count: pair[st_nstates][1 << 2 st_cshift] u32 -- two bytes
  // per dependent lookup: (row offset of the state after both bytes) << 4 | flags of byte 1 << 2 | flags
  // of byte 0, indexed by row offset + (class of byte 0 << st_cshift | class of byte 1); -1: none
  int32_t off_stg_pair;
  // multi-walk automaton (PF_MWALK; mw_ncfg == 0: none).  The restart-per-position search as ONE pass: the walks
  // begun at every candidate byte since the last match run side by side, oldest first.  A configuration is
  // (the oldest walk has accepted, the DFA states of the live walks in age order); cls[256] u8 gives the byte's
  // class, tab[config][1 << mw_cshift] u32 the step: bits 16.. = row offset of the next configuration, bit 0 =
  // EMIT (the oldest walk ended behind its last accepting position: the match is [start of slot 0, that position)),
  // bit 1 = the oldest walk accepts behind this byte, bit 10 = "the oldest walk has accepted" in the new
  // configuration (the end-of-text rule), bits 2-4 / 5-6 / 7-8 / 9 = where the start register of walk slot
  // 0 / 1 / 2 / 3 comes from: v <= 3 - j: old slot j + v, v == 4 - j: this byte (a walk begins here).
  int32_t off_mw_cls, off_mw_tab, mw_ncfg, mw_cshift, mw_bytes, mw_k;   // mw_k: the most walks any configuration holds
  // the same table form for the required-byte route (PF_MWALK_REQ): walks begin where a run of first-class bytes
  // begins, count from the required byte that ends their run ("hits"), report only behind it; see build_reqwalk()
  int32_t off_mwr_cls, mwr_ncfg, mwr_cshift, mwr_bytes, mwr_k;
  // backward table (PF_BACKSET): cls[256] u8 | tab[bk_nsub][1 << bk_cshift] u16.  The state of the right-to-left scan
  // behind position s is the SET of DFA states from which the text from s on leads to an accepting state (the
  // sets are numbered on the host); entry = next set << 1 | "a match begins at this byte" (the start state's
  // transition on the byte lands in the set, and the byte may start a walk); bk_start = the set at the text's end.
  int32_t off_bk_cls, bk_nsub, bk_cshift, bk_bytes, bk_start;
  // synchronising bytes of the search automaton: sync[b] != 0 when byte b takes EVERY state to the
  // same state with the same start (idle, or a new start at b) -- after such a byte the walk does not
  // depend on what came before, so a long text can be cut there (st_nsync = how many, 0 = no table)
  int32_t off_st_sync, st_nsync;
  // a byte that takes every state to idle without starting a walk and emits exactly from the accepting
  // states: bytes of a frame that lie outside the text may be replaced by it (-1: no such byte)
  int32_t st_reset_byte;
  // anchored automaton for match_first on the streaming kernel (fa_bytes == 0: none); same layout
  // as kind 2 but entry = (next << fa_cshift) << 2 | ACCEPT(next) << 1, last row = dead state
  int32_t off_fa_cls, off_fa_trans, fa_cshift, fa_bytes, fa_nstates, fa_start_acc;
  // OnePass '$' fixup (onepass.mojo:480-484): u8 per state (dead row included), consulted when the
  // walk reaches the end of the text alive; -1 = the automaton has no such flags
  int32_t off_fa_end;
  // the anchored automaton is "one or more bytes of a class" (start -C-> s, s -C-> s, s accepts): u8[256]
  // membership table; match_first is then the length of the class run at 0 (k_first_run); -1: not so
  int32_t off_fa_run;
  // fa_kind: 2 = class table (above), 1 / 3 = byte-column forms as for the search automaton
  // (u16 / u64 columns at off_fa_col; the dead state is the last field, entry = next | ACC << 1)
  int32_t fa_kind, off_fa_col;
  // bitset NFA (PF_BITSET): cls[256] u8, byte masks u64[bs_ncls][bs_nw], follow u64[bs_npos][bs_nw]
  int32_t bs_nw, bs_npos, bs_ncls, off_bs_cls, off_bs_mask, off_bs_follow;
  uint64_t bs_start[4], bs_match[4];
  int32_t bs_fixed_len, bs_pad_;   // > 0: every match of the program has this length (k_bscan modes 2-4 need no second pass)
  // the backtracking matcher as a flat program (BtProg, mrx_engines.hpp; bt_nitems == 0: none): BtItem
  // [bt_nitems], membership bitmaps u8[32] x 3 per leaf, and NFAEngine's literal prefilter facts
  // (nfa.mojo:86-143): bt_flags bit 0 = has_literal_optimization, 1 = is_prefix_literal, 2 = starts_with_dotstar,
  // 3 = ends_with_dotstar
  int32_t off_bt_items, bt_nitems, off_bt_tbl, bt_ngroups, off_bt_lit, bt_lit_len, bt_flags, bt_pattern_len;
};

// empty-match plans whose walks read beyond their match: the one-pass table (build_emptywalk2(), mrx_plan.cpp)
struct EwEntry { uint32_t x = 0; uint32_t r[7] = {0, 0, 0, 0, 0, 0, 0}; };   // control word | up to 28 (a, len) reports (mrx_plan.cpp)
struct EmptyWalk2 {
  std::array<uint8_t, 256> cls{};
  int ncls = 0, cshift = 0, ncfg = 0;
  bool empty = true;          // the plan has empty matches (a try at every position, the last one at len)
  std::vector<EwEntry> tab;   // [ncfg][1 << cshift]
  std::vector<EwEntry> end;   // [ncfg]
};
std::vector<std::pair<int, int>> emptywalk2_run(const EmptyWalk2& ew, const uint8_t* text, int len);

// Leaves of a chain program in order (BtProg without ALT / LOOP / anchors), every leaf's three membership tables equal,
// min >= 1; a leaf with a variable count is followed by a leaf whose set is disjoint from its own.  gopen / gclose: the
// number of leaves in front of a capturing group's OPEN / CLOSE (-1: no such group).  mask[byte] bit i: leaf i takes it.
constexpr int kChainLeaves = 16;
struct ChainGroups {
  bool ok = false;
  std::string why;
  int nleaf = 0;
  int lmin[kChainLeaves] = {0}, lmax[kChainLeaves] = {0};
  int gopen[10], gclose[10];
  std::array<uint16_t, 256> mask{};
};
struct HostPlan {
  EmptyWalk2 ew2;   // PF_STEP_EMPTY plans that are not every_state_accepts: the general one-pass table, when it exists
  bool ew2_ok = false;
  bool ew2_tries = false;   // ew2 is the table of a plan WITHOUT empty matches (multi-walk proofs failed)
  std::string ew2_why;
  bool empty_all_accepting = false;   // PF_STEP_EMPTY plans: every state a walk can reach accepts (a walk never overshoots)
  std::string pattern;
  // routing facts (for mrx_engine_type / mrx_stats / mrx_describe)
  Complexity complexity = CX_SIMPLE;
  bool wildcard_any = false;
  bool use_dfa = false;
  bool use_pure_dfa = false;
  bool force_nfa = false;  // MRX_COMPILE_LAZYDFA_SEMANTICS
  bool exact_literal = false, literal_has_anchors = false;
  std::string best_literal;
  bool has_prefilter = false;
  std::string prefilter_literal;
  int required_byte = -1;
  std::string engine_type, stats;
  DfaEngine dfa;
  Program program;
  LazyTables lazy;
  BitsetNfa bitset;
  OnePassTables onepass;
  BtProg bt;                   // NFAEngine's recursive matcher as a flat program (general capture groups)
  std::string nfa_literal;     // NFAEngine.literal_prefix (nfa.mojo:108-125)
  bool first_onepass = false;  // match_first runs the OnePass tables (NFA-routed '$' pattern)
  bool force_bitset = false;  // MRX_COMPILE_BITSET_NFA
  bool nfa_engine = false;    // MRX_COMPILE_NFA_ENGINE
  bool dfa_engine = false;    // MRX_COMPILE_DFA_ENGINE
  bool nfa_has_literal_opt = false, nfa_starts_dotstar = false, nfa_ends_dotstar = false;
  // per-operation support: empty string = supported, else the reason
  std::string why_no_match_first, why_no_search;
  // fixed-width group form
  int fixed_total = -1, fixed_ngroups = 0;
  bool fixed_concat = false;
  bool fixed_pure = false;   // the pattern is nothing but (\d{N}) groups (see build_plan)
  int fixed_off[10] = {0}, fixed_w[10] = {0};
  // general capture groups of a deterministic chain whose matches are the table walk's (build_plan: "chain groups"):
  // regex.sub with \1..\9 takes the spans of the plain search and finds the group boundaries as runs of the leaves' classes
  ChainGroups chain;
  // device payload
  DevPlan dev{};
  std::vector<uint8_t> blob;
  std::string streamable_why_not;
  std::string mwalk_why_not, mwalk_req_why_not, backset_why_not;
  std::string first_stream_why_not;
};

// Throws SyntaxError for patterns the reference's parser raises on.
// force_nfa: route as if DFAEngine compilation had failed (matcher.mojo:666-672), i.e. the
// NFAMatcher / LazyDFA path -- the "LazyDFA semantics" switch of SURVEY.md 8(c).
// force_bitset: LazyDFA-routed patterns walk the bitset NFA even when the determinised table fits.
// nfa_engine: the Engine is NFAEngine itself (nfa.mojo:66-143), no hybrid router in front.
// dfa_engine: the Engine is compile_dfa_pattern's DFAEngine (dfa.mojo:2385-2496), no hybrid router in front.
void build_plan(const std::string& pattern, HostPlan& out, bool force_nfa = false,
                bool force_bitset = false, bool nfa_engine = false, bool dfa_engine = false);
std::string describe_plan(const HostPlan& p);

// replacement template (matcher.mojo:1436-1482)
struct ReplSeg {
  int32_t group_ref, start, length;
};
bool repl_has_group_refs(const std::string& repl);
std::vector<ReplSeg> parse_repl_template(const std::string& repl);

}  // namespace mrx
