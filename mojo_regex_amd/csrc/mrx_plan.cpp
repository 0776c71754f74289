// Routing + device payload construction.  See mrx_plan.hpp.
#include "mrx_plan.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <set>
#include <sstream>

namespace mrx {
namespace {

// _is_simple_pattern_skip_prefilter, matcher.mojo:447-532
bool skip_prefilter(const std::string& p) {
  const int n = (int)p.size();
  if (n <= 4) return true;
  bool q = false, alt = false, anch = false, wild = false;
  for (int i = 0; i < n; ++i) {
    const char c = p[i];
    if (c == '*' || c == '+' || c == '?') q = true;
    else if (c == '|') alt = true;
    else if (c == '^' && i == 0) anch = true;
    else if (c == '$' && i == n - 1) anch = true;
    else if (c == '.') wild = true;
  }
  if (q && !alt && !wild) return true;
  if (alt && n <= 10 && !wild) {
    const int bars = (int)std::count(p.begin(), p.end(), '|');
    if (bars >= n / 3) return true;
  }
  if (anch && n <= 10) return true;
  if (wild && n >= 8) return false;
  if (alt && n >= 12) return false;
  if (n >= 15) return false;
  return true;
}

bool ast_has_anchors(const Ast& a, const Node& n) {  // matcher.mojo:168-178
  if (n.type == N_START || n.type == N_END) return true;
  if (n.type == N_GROUP || n.type == N_RE)
    for (int i = 0; i < a.nkids(n); ++i)
      if (ast_has_anchors(a, a.child(n, i))) return true;
  return false;
}

int rare_required_byte(const Ast& a, const std::array<uint8_t, 256>& lookup) {
  // matcher.mojo:122-165
  const Node* n = &a.root;
  if (n->type == N_RE && a.nkids(*n) == 1) n = &a.child(*n, 0);
  if (n->type != N_GROUP) return -1;
  for (int i = 0; i < a.nkids(*n); ++i) {
    const Node& c = a.child(*n, i);
    if (c.type != N_ELEMENT || c.min < 1) continue;
    auto v = a.value(c);
    if (v.size() != 1) continue;
    const int b = (unsigned char)v[0];
    if (lookup[b] == 0) return b;
  }
  return -1;
}

// _detect_fixed_width_groups, matcher.mojo:1485-1586
bool detect_fixed_width(const std::string& p, std::vector<int>& segs) {
  const int n = (int)p.size();
  int i = 0, lit = 0;
  segs.clear();
  while (i < n) {
    if (p[i] == '(') {
      if (lit > 0) { segs.push_back(-lit); lit = 0; }
      if (i + 1 < n && p[i + 1] == '?') return false;
      ++i;
      if (i + 1 >= n || p[i] != '\\' || p[i + 1] != 'd') return false;
      i += 2;
      if (i < n && p[i] == '{') {
        ++i;
        const int ns = i;
        while (i < n && p[i] >= '0' && p[i] <= '9') ++i;
        if (i == ns || i >= n || p[i] != '}') return false;
        long w = 0;
        for (int j = ns; j < i; ++j) w = w * 10 + (p[j] - '0');
        ++i;
        segs.push_back((int)w);
      } else if (i < n && p[i] == ')') {
        segs.push_back(1);
      } else {
        return false;
      }
      if (i >= n || p[i] != ')') return false;
      ++i;
    } else if (p[i] == '|' || p[i] == '[') {
      return false;
    } else {
      if (p[i] == '\\' && i + 1 < n) { ++lit; i += 2; }
      else { ++lit; ++i; }
    }
  }
  for (int s : segs)
    if (s > 0) return true;
  return false;
}

// ---- single-pass ("streaming") search automaton --------------------------------
// findall's restart-per-position loop (dfa.mojo:2074-2128) visits candidate starts
// in increasing order.  The streaming kernel replaces it by ONE left-to-right pass
// that, when a walk dies at byte q, restarts at q itself.  That is equivalent to
// the reference loop iff (a) a successful walk always dies right at its last
// accepting position and (b) when a walk started at p fails at q, every walk
// started in (p, q) fails too.  Both are decided here on the actual table by
// exploring pairs (state of the earlier walk, state of a later walk).
struct SearchAutomaton {
  int n = 0;                                  // states, 0 = idle/start
  std::vector<std::array<int, 256>> next;     // -1 dead
  std::vector<uint8_t> acc;
  std::array<uint8_t, 256> allowed{};         // bytes a walk may start on
};

bool check_streamable(const SearchAutomaton& s, std::string& why) {
  if (s.acc[0]) { why = "start state accepts (empty matches)"; return false; }
  for (int q = 0; q < s.n; ++q)
    for (int c = 0; c < 256; ++c)
      if (s.next[q][c] == 0) { why = "transition back into the start state"; return false; }
  // (a) accept-closed
  for (int q = 0; q < s.n; ++q)
    if (s.acc[q])
      for (int c = 0; c < 256; ++c) {
        const int t = s.next[q][c];
        if (t >= 0 && !s.acc[t]) { why = "accepting state continues into a non-accepting one"; return false; }
      }
  // reachable states
  std::vector<uint8_t> reach(s.n, 0);
  std::vector<int> st;
  for (int c = 0; c < 256; ++c)
    if (s.allowed[c] && s.next[0][c] > 0 && !reach[s.next[0][c]]) {
      reach[s.next[0][c]] = 1; st.push_back(s.next[0][c]);
    }
  while (!st.empty()) {
    const int q = st.back(); st.pop_back();
    for (int c = 0; c < 256; ++c) {
      const int t = s.next[q][c];
      if (t > 0 && !reach[t]) { reach[t] = 1; st.push_back(t); }
    }
  }
  // (b) pair exploration: x = earlier (still failing) walk, y = later walk
  // y == -2: the later walk starts on the next byte ("fresh")
  std::set<std::pair<int, int>> seen;
  std::vector<std::pair<int, int>> work;
  for (int x = 1; x < s.n; ++x)
    if (reach[x] && !s.acc[x]) work.push_back({x, -2});
  while (!work.empty()) {
    auto [x, y] = work.back(); work.pop_back();
    if (!seen.insert({x, y}).second) continue;
    for (int c = 0; c < 256; ++c) {
      const int xn = s.next[x][c];
      const bool fresh = (y == -2);
      const int yn = fresh ? (s.allowed[c] ? s.next[0][c] : -1) : s.next[y][c];
      if (xn >= 0 && s.acc[xn]) continue;  // earlier walk succeeds: not a failed walk
      if (yn < 0) continue;                // later walk dies here
      if (xn < 0) {
        if (fresh) continue;               // later walk starts at the death byte: that IS the restart
        why = "a later start survives the byte that kills the earlier walk";
        return false;
      }
      if (xn == yn) continue;              // merged
      if (s.acc[yn]) { why = "a later start accepts while the earlier walk is still undecided"; return false; }
      work.push_back({xn, yn});
    }
  }
  return true;
}

// Entry matrix of the search automaton over live states: next<<2 | EMIT<<1 | NEWSTART
std::vector<std::array<uint16_t, 256>> stream_entries(const SearchAutomaton& s,
                                                      const std::vector<int>& remap, int nlive) {
  std::vector<std::array<uint16_t, 256>> E(nlive);
  for (int q = 0; q < s.n; ++q) {
    if (remap[q] < 0) continue;
    for (int c = 0; c < 256; ++c) {
      int t = (q == 0) ? (s.allowed[c] ? s.next[0][c] : -1) : s.next[q][c];
      bool emit = false, newstart = false;
      if (q == 0) {
        newstart = t > 0;
      } else if (t < 0) {  // the walk dies: emit if it had accepted, restart on this byte
        emit = s.acc[q] != 0;
        t = s.allowed[c] ? s.next[0][c] : -1;
        newstart = t > 0;
      }
      const int tn = t > 0 ? remap[t] : 0;
      E[remap[q]][c] = (uint16_t)((tn << 2) | (emit ? 2 : 0) | (newstart ? 1 : 0));
    }
  }
  return E;
}

// ---- several walks in one pass ("multi-walk" automaton, DevPlan::off_mw_*) ---------------------------------------
// What check_streamable() rejects still has a one-pass form when the walks that the reference starts one after the
// other (findall's `pos + 1` / `pos = end` loop, dfa.mojo:2096-2130, pikevm.mojo:755-817) are run SIDE BY SIDE: the
// list of walks begun at candidate bytes since the last reported match, oldest first.  Per byte every walk takes
// its table step; a dead walk leaves the list; a walk begins on the byte if it may (first-byte filter) -- unless a
// live walk is already in that state (same future, the older one wins: leftmost); two walks in one state merge the
// same way.  When the OLDEST walk accepts, every younger one began inside its match and is dropped (the reference
// resumes at the match end); when the oldest walk dies after having accepted the match [its start, its last
// accepting position) is reported and the next-oldest walk -- begun at or behind that position, by construction --
// takes over; when it dies without having accepted, likewise without a report (the reference's `pos + 1`).
// Not covered, and left to the stepper: a younger walk that accepts while an older one is still undecided (its
// match would have to be remembered until the older walk is settled), more than four walks at a time, tables
// beyond the LDS budget.
struct MultiWalk {
  std::array<uint8_t, 256> cls{};
  int ncls = 0, cshift = 0, ncfg = 0, kmax = 0;
  std::vector<uint32_t> tab;   // [ncfg][1 << cshift]
};

bool build_multiwalk(const SearchAutomaton& s, MultiWalk& mw, std::string& why) {
  constexpr int K = 4;
  if (s.acc[0]) { why = "start state accepts (empty matches)"; return false; }
  // byte classes: identical columns over all states and the same "may start a walk"
  std::vector<int> rep;
  {
    std::map<std::vector<int>, int> seen;
    for (int c = 0; c < 256; ++c) {
      std::vector<int> col(s.n + 1);
      for (int q = 0; q < s.n; ++q) col[q] = s.next[q][c];
      col[s.n] = s.allowed[c];
      auto it = seen.find(col);
      if (it == seen.end()) { it = seen.emplace(col, (int)rep.size()).first; rep.push_back(c); }
      mw.cls[c] = (uint8_t)it->second;
    }
  }
  mw.ncls = (int)rep.size();
  mw.cshift = 0;
  while ((1 << mw.cshift) < mw.ncls) ++mw.cshift;
  const int ncp = 1 << mw.cshift;
  using Key = std::pair<int, std::vector<int>>;   // (oldest has accepted, states oldest first)
  std::map<Key, int> ids;
  std::vector<Key> cfgs;
  auto id_of = [&](const Key& k) {
    auto it = ids.find(k);
    if (it == ids.end()) { it = ids.emplace(k, (int)cfgs.size()).first; cfgs.push_back(k); }
    return it->second;
  };
  id_of({0, {}});
  std::vector<std::array<uint32_t, 64>> rows;   // per config, per class (ncls <= 64 checked below)
  if (mw.ncls > 64) { why = "multi-walk: more than 64 byte classes"; return false; }
  for (size_t ci = 0; ci < cfgs.size(); ++ci) {
    if ((int64_t)cfgs.size() * ncp > 8192) { why = "multi-walk: configuration table beyond the LDS budget"; return false; }
    const Key cur = cfgs[ci];   // (copy: cfgs grows)
    std::array<uint32_t, 64> row{};
    for (int k = 0; k < mw.ncls; ++k) {
      const int c = rep[k];
      const int n_old = (int)cur.second.size();
      // advance; provenance = old slot index, or -1 for the walk that begins on this byte
      std::vector<std::pair<int, int>> lst;   // (state, provenance)
      bool emit = false;
      int A = cur.first;
      for (int i = 0; i < n_old; ++i) {
        const int t = s.next[cur.second[i]][c];
        if (t < 0) {
          if (i == 0) { emit = A != 0; A = 0; }
          continue;
        }
        if (t == 0) { why = "transition back into the start state"; return false; }
        lst.push_back({t, i});
      }
      if (n_old > 0 && (lst.empty() || lst[0].second != 0)) A = 0;   // the walk the flag belonged to is gone
      const int b = s.allowed[c] ? s.next[0][c] : -1;
      if (b == 0) { why = "transition back into the start state"; return false; }
      if (b > 0) lst.push_back({b, -1});
      // merge: the older walk in a state wins
      std::vector<std::pair<int, int>> m;
      for (const auto& x : lst) {
        bool dup = false;
        for (const auto& y : m) dup = dup || y.first == x.first;
        if (!dup) m.push_back(x);
      }
      bool acc_now = false;
      if (!m.empty() && s.acc[m[0].first]) { acc_now = true; A = 1; m.resize(1); }   // younger walks lie inside the match
      for (size_t j = 1; j < m.size(); ++j)
        if (s.acc[m[j].first]) { why = "a later start accepts while the earlier walk is still undecided"; return false; }
      if ((int)m.size() > K) { why = "multi-walk: more than four walks at a time"; return false; }
      mw.kmax = std::max(mw.kmax, (int)m.size());
      Key nk{m.empty() ? 0 : A, {}};
      for (const auto& x : m) nk.second.push_back(x.first);
      const int nid = id_of(nk);
      uint32_t e = ((uint32_t)nid << mw.cshift) << 16;
      if (emit) e |= 1u;
      if (acc_now) e |= 2u;
      if (nk.first) e |= 1u << 10;
      static const int shift[4] = {2, 5, 7, 9};
      for (int j = 0; j < (int)m.size(); ++j) {
        const int code = m[j].second < 0 ? K - j : m[j].second - j;   // provenance >= j: walks only ever leave the list
        e |= (uint32_t)code << shift[j];
      }
      if ((((uint64_t)nid << mw.cshift) >> 16) != 0) { why = "multi-walk: configuration table beyond the LDS budget"; return false; }
      row[k] = e;
    }
    rows.push_back(row);
  }
  mw.ncfg = (int)cfgs.size();
  if ((int64_t)mw.ncfg * ncp > 8192) { why = "multi-walk: configuration table beyond the LDS budget"; return false; }
  mw.tab.assign((size_t)mw.ncfg * ncp, 0);
  for (int ci = 0; ci < mw.ncfg; ++ci)
    for (int k = 0; k < mw.ncls; ++k) mw.tab[(size_t)ci * ncp + k] = rows[ci][k];
  return true;
}

// Empty matches in one pass (PF_MW_EMPTY).  The reference's loop (dfa.mojo:2118-2130 / pikevm.mojo:805-817): try at pos --
// the start state accepts, so every try matches --, report (pos, end), resume at end, or one byte on after an empty
// match; the last try is at pos == len.  When EVERY state a walk can reach accepts, a walk dies ON the byte behind its
// match: nothing is read twice, one walk at a time is the whole search, and a byte sees at most two reports -- the match
// it ends (start, pos) and, when no walk can begin on it either, the empty match (pos, pos).  In k_mwalk's entry format:
// configuration = the walk's state (0: nothing consumed yet); bit 0 = the oldest walk ends behind its last accepting
// position (report (start, last): every step accepts, so last is this byte's position), bit 1 = accepts now, bits 2-4 =
// 4: the start register takes this byte's position, bit 10 = the walk has consumed something (reported at the text's
// end), bit 11 = an empty match at this byte.  The empty match at len is the kernel's (PF_MW_EMPTY).
bool build_emptywalk(const SearchAutomaton& s, MultiWalk& mw, std::string& why) {
  std::vector<int> rep;
  {
    std::map<std::vector<int>, int> seen;
    for (int c = 0; c < 256; ++c) {
      std::vector<int> col(s.n);
      for (int q = 0; q < s.n; ++q) col[q] = s.next[q][c];
      auto it = seen.find(col);
      if (it == seen.end()) { it = seen.emplace(col, (int)rep.size()).first; rep.push_back(c); }
      mw.cls[c] = (uint8_t)it->second;
    }
  }
  mw.ncls = (int)rep.size();
  if (mw.ncls > 64) { why = "empty-match walk: more than 64 byte classes"; return false; }
  mw.cshift = 0;
  while ((1 << mw.cshift) < mw.ncls) ++mw.cshift;
  const int ncp = 1 << mw.cshift;
  // configurations: 0 = nothing consumed yet (the walk stands in the start state at the search position), then the
  // states a walk can be in after a byte (the start state among them when the table loops back into it)
  std::vector<int> id(s.n, -1), order;
  auto cfg_of = [&](int t) {
    if (id[t] < 0) { id[t] = 1 + (int)order.size(); order.push_back(t); }
    return id[t];
  };
  for (int k = 0; k < mw.ncls; ++k)
    if (s.next[0][rep[k]] >= 0) cfg_of(s.next[0][rep[k]]);
  for (size_t i = 0; i < order.size(); ++i)
    for (int k = 0; k < mw.ncls; ++k) {
      const int t = s.next[order[i]][rep[k]];
      if (t >= 0) cfg_of(t);
    }
  mw.ncfg = 1 + (int)order.size();
  if ((int64_t)mw.ncfg * ncp > 8192 || ((((uint64_t)mw.ncfg) << mw.cshift) >> 16) != 0) {
    why = "empty-match walk: table beyond the LDS budget";
    return false;
  }
  mw.kmax = 1;
  mw.tab.assign((size_t)mw.ncfg * ncp, 0);
  for (int ci = 0; ci < mw.ncfg; ++ci) {
    const bool fresh = ci == 0;
    const int q = fresh ? 0 : order[ci - 1];
    for (int k = 0; k < mw.ncls; ++k) {
      const int c = rep[k];
      uint32_t e = 0;
      int t = s.next[q][c];
      if (t < 0) {               // the walk dies on this byte
        if (!fresh) {            // ... behind a match: reported, and the search resumes ON this byte
          e |= 1u;
          t = s.next[0][c];
        }
        if (t < 0) e |= 1u << 11;   // no walk begins on it either: the empty match here, the search resumes one byte on
      }
      int nid = 0;
      if (t >= 0) {
        nid = id[t];
        e |= 2u | (1u << 10);                    // consumed and accepting
        if (fresh || (e & 1u)) e |= 4u << 2;     // a walk began on this byte
      }
      e |= ((uint32_t)nid << mw.cshift) << 16;
      mw.tab[(size_t)ci * ncp + k] = e;
    }
  }
  return true;
}

// The general form (PF_MW_EMPTY with DevPlan::mw_k == -2): walks that read up to fourteen bytes beyond their last accepting
// position (`(ab)*`, `(foo)?x*`).  When such a walk dies at byte p the reference resumes at its match's end (or one byte
// behind its start after an empty match) and tries again from there -- over bytes this pass has already seen.  So the
// tries that MAY be asked for run beside the oldest walk W0: one per position from W0's resume point R up to p, each
// alive (its state, how far its match reaches) or dead (how far its match reached).  A configuration is W0's state and
// that list (at most fourteen entries); W0's start and match end are the lane's registers.  Per byte: every walk steps, a
// walk begins on the byte; when W0 accepts, the list is dropped (every entry began inside its match); when W0 dies, its
// match is reported and the list is chased as the reference's loop would: a dead entry is reported and skipped past, the
// first live entry reached takes over as W0 (the entries in front of ITS resume point are dropped), and when the chase
// runs off the list the next byte starts afresh.  Entry, 32 bytes (EwEntry): x bit 0 report W0 (start, last registers),
// bit 1 W0 accepts here, bits 2-5 a + 1 / bits 6-9 len of the entry that takes over (start = pos - a, last = start +
// len; 0: none), bits 10-14 how many dead entries are reported, bits 16-31 the next configuration; r[0..6]: their (a, len)
// pairs, eight bits each, in report order (read only when there are any).  end[config]: the same for the end of the text
// (every walk dies; a relative to len); the empty match at len is the kernel's.  Up to fourteen entries behind W0.
// empty = false: the same search for plans WITHOUT empty matches (the plain restart-per-position route of a table plan
// that fails the multi-walk proofs, e.g. `foo|[a-z]{3}\d|[ab]`: a later start accepts while the earlier walk is still
// undecided -- here that later try simply waits in the list with its match).  A try that leaves no match (it dies before
// it accepts, or its byte may not start a walk) is a dead entry of length 0 that is skipped WITHOUT a report, W0 is
// reported only once it has accepted (part of the configuration), and there is no try at len.
bool build_emptywalk2(const SearchAutomaton& s, EmptyWalk2& ew, std::string& why, bool empty = true) {
  constexpr int kSlots = 14, kReports = 28;
  ew.empty = empty;
  std::vector<int> rep;
  {
    std::map<std::vector<int>, int> seen;
    for (int c = 0; c < 256; ++c) {
      std::vector<int> col(s.n + 1);
      for (int q = 0; q < s.n; ++q) col[q] = s.next[q][c];
      col[s.n] = empty ? 1 : s.allowed[c];
      auto it = seen.find(col);
      if (it == seen.end()) { it = seen.emplace(col, (int)rep.size()).first; rep.push_back(c); }
      ew.cls[c] = (uint8_t)it->second;
    }
  }
  ew.ncls = (int)rep.size();
  if (ew.ncls > 64) { why = "empty-match walk: more than 64 byte classes"; return false; }
  ew.cshift = 0;
  while ((1 << ew.cshift) < ew.ncls) ++ew.cshift;
  const int ncp = 1 << ew.cshift;
  struct Slot { int state; int rel; };   // state < 0: dead; rel: match length so far
  using Key = std::vector<int>;          // {q0, W0 has a match to report, state, rel, state, rel, ...}; {-1}: nothing consumed yet (fresh)
  std::map<Key, int> ids;
  std::vector<Key> cfgs;
  auto id_of = [&](const Key& k) {
    auto it = ids.find(k);
    if (it == ids.end()) { it = ids.emplace(k, (int)cfgs.size()).first; cfgs.push_back(k); }
    return it->second;
  };
  id_of(Key{-1});
  // the chase behind W0's death: list[idx] is the try at position R + idx; reports go to `out` as (a, len) with a = base - position
  auto chase = [&](const std::vector<Slot>& list, int base_minus_R, std::vector<std::pair<int, int>>& out, int* take_idx) {
    *take_idx = -1;
    int idx = 0;
    while (idx < (int)list.size()) {
      if (list[idx].state >= 0) { *take_idx = idx; return; }
      if (empty || list[idx].rel > 0) out.push_back({base_minus_R - idx, list[idx].rel});   // (a try without a match: skipped silently)
      idx += list[idx].rel > 0 ? list[idx].rel : 1;
    }
  };
  // Entries the chase can never reach are all alike: the first entry is where the chase begins; behind a dead entry it goes on
  // at its match's end (one byte on when it left none); an entry inside a match that can only grow (a dead one's, or a
  // live one's so far) is never asked for.  They become "dead, no match" -- fewer configurations, same reports.
  auto canon = [&](std::vector<Slot>& list, size_t from) {
    size_t idx = from;
    while (idx < list.size()) {
      const Slot e = list[idx];
      for (size_t j = idx + 1; j < idx + (size_t)e.rel && j < list.size(); ++j) list[j] = Slot{-1, 0};
      if (e.state >= 0) break;   // (how far a live entry's match will reach is not known yet)
      idx += e.rel > 0 ? (size_t)e.rel : 1;
    }
  };
  std::vector<std::vector<EwEntry>> rows;
  std::vector<std::vector<int>> nexts;   // [configuration][class]: the next configuration
  for (size_t ci = 0; ci < cfgs.size(); ++ci) {
    // 32-byte entries, 41 KB at most, and no more than the plan's blob has room for (60 KiB for everything the generic kernels
    // stage: checked where the table is stored).  Small tables gain 3-10 x over marks + stepper; tables of 16-41 KB run at
    // about half their speed (two workgroups per CU) and win two times out of three (tools/r04_tries_cap.py); on the
    // reference's phone patterns by pieces the form is slower (flexible_phone 269 -> 195 GB/s).  Plans without empty matches
    // are therefore MEASURED per handle against marks + stepper (FindallJob::tries_route_tuner).
    // (the configurations are explored first and MERGED afterwards -- below -- so the limit here is on the exploration)
    if (cfgs.size() > 20000) { why = "pending-tries walk: more than 20000 configurations explored"; return false; }
    const Key cur = cfgs[ci];
    const bool fresh = cur[0] < 0;
    const int q0 = fresh ? 0 : cur[0];
    const bool w0acc = !fresh && cur[1] != 0;
    std::vector<Slot> slots;
    for (size_t j = 2; j + 1 < cur.size(); j += 2) slots.push_back({cur[j], cur[j + 1]});
    const int m = (int)slots.size();   // tries at positions R .. p - 1, R = p - m
    std::vector<EwEntry> row(ncp + 1);
    std::vector<int> nrow(ew.ncls, 0);
    for (int k = 0; k <= ew.ncls; ++k) {
      const bool at_end = k == ew.ncls;   // the virtual step behind the last byte: every walk dies, none begins
      const int c = at_end ? 0 : rep[k];
      EwEntry e;
      std::vector<Slot> list;
      for (int j = 0; j < m; ++j) {
        Slot sl = slots[j];
        if (sl.state >= 0) {
          const int t = at_end ? -1 : s.next[sl.state][c];
          if (t < 0) sl.state = -1;
          else { sl.state = t; if (s.acc[t]) sl.rel = (m - j) + 1; }   // its start is p - (m - j): the match now ends behind p
        }
        list.push_back(sl);
      }
      if (!at_end) {
        const int tn = (empty || s.allowed[c]) ? s.next[0][c] : -1;
        list.push_back(tn < 0 ? Slot{-1, 0} : Slot{tn, s.acc[tn] ? 1 : 0});
      }
      const int t0 = at_end ? -1 : s.next[q0][c];
      Key nk;
      if (fresh) {
        // the try at p is the newest list entry itself
        const Slot me = list.back();
        if (at_end) { nk = Key{-1}; }
        else if (me.state < 0) { if (empty) e.x |= 1u << 10; nk = Key{-1}; }   // the empty match at p: one report, (a, len) = (0, 0)
        else {
          e.x |= 2u * (s.acc[me.state] ? 1 : 0);
          e.x |= (uint32_t)(0 + 1) << 2;                       // takes over: start = p ...
          e.x |= (uint32_t)(s.acc[me.state] ? 1 : 0) << 6;     // ... last = start + rel
          nk = Key{me.state, (empty || s.acc[me.state]) ? 1 : 0};   // (no entries: its resume point is p + 1 either way)
        }
      } else if (t0 >= 0 && s.acc[t0]) {
        e.x |= 2u;
        nk = Key{t0, 1};
      } else if (t0 >= 0) {
        if ((int)list.size() > kSlots) { why = "pending-tries walk: a walk reads more than fourteen bytes beyond its last accepting position"; return false; }
        nk = Key{t0, w0acc ? 1 : 0};
        canon(list, 0);
        for (const Slot& sl : list) { nk.push_back(sl.state); nk.push_back(sl.rel); }
      } else {
        if (w0acc) e.x |= 1u;   // W0's match
        std::vector<std::pair<int, int>> out;
        int take = -1;
        // positions: list[idx] is at R + idx with R = p - m; relative to base (p, or len = p at the end): a = base - R - idx = m - idx
        chase(list, m, out, &take);
        if ((int)out.size() > kReports) { why = "pending-tries walk: more than 28 reports on one byte"; return false; }
        e.x |= (uint32_t)out.size() << 10;   // (five bits)
        for (size_t r = 0; r < out.size(); ++r)
          e.r[r / 4] |= (uint32_t)((out[r].first & 15) | ((out[r].second & 15) << 4)) << (8 * (r % 4));
        if (take >= 0) {
          const Slot w = list[take];
          const int a = m - take;
          e.x |= (uint32_t)(a + 1) << 2;
          e.x |= (uint32_t)w.rel << 6;
          nk = Key{w.state, (empty || w.rel > 0) ? 1 : 0};
          // its resume point: behind its match, or one byte behind its start; the entries from there on stay
          const int keep_from = take + (w.rel > 0 ? w.rel : 1);
          canon(list, (size_t)keep_from);
          for (int j = keep_from; j < (int)list.size(); ++j) { nk.push_back(list[j].state); nk.push_back(list[j].rel); }
          if ((int)(nk.size() - 2) / 2 > kSlots) { why = "pending-tries walk: a walk reads more than fourteen bytes beyond its last accepting position"; return false; }
        } else nk = Key{-1};
      }
      if (at_end) { row[ncp] = e; continue; }
      nrow[k] = id_of(nk);   // (the next configuration is packed into the entry once the configurations are merged)
      row[k] = e;
    }
    rows.push_back(row);
    nexts.push_back(nrow);
  }
  // Merge configurations that behave alike (the table is a Mealy machine: partition refinement on what an entry does
  // and which block it leads to).  Lists that differ only in tries that never matter again -- the usual case once a list
  // holds a dozen entries -- collapse; the reports are unchanged by construction.
  const int nc = (int)cfgs.size();
  std::vector<int> block(nc, 0);
  int nblocks = 1;
  for (;;) {
    std::map<std::vector<uint32_t>, int> sig_ids;
    std::vector<int> nb(nc);
    for (int ci = 0; ci < nc; ++ci) {
      std::vector<uint32_t> sig;
      sig.reserve((size_t)(ew.ncls + 1) * 9);
      for (int k = 0; k <= ew.ncls; ++k) {
        const EwEntry& e = rows[ci][k == ew.ncls ? ncp : k];
        sig.push_back(e.x);
        for (int w = 0; w < 7; ++w) sig.push_back(e.r[w]);
        sig.push_back(k == ew.ncls ? 0u : (uint32_t)block[nexts[ci][k]]);
      }
      auto it = sig_ids.find(sig);
      if (it == sig_ids.end()) it = sig_ids.emplace(std::move(sig), (int)sig_ids.size()).first;
      nb[ci] = it->second;
    }
    const int n2 = (int)sig_ids.size();
    block.swap(nb);
    if (n2 == nblocks) break;
    nblocks = n2;
  }
  // block of the fresh configuration first (row 0 is where a text begins)
  std::vector<int> order(nblocks, -1), repr(nblocks, -1);
  int next_id = 0;
  order[block[0]] = next_id++;
  for (int ci = 0; ci < nc; ++ci) {
    if (order[block[ci]] < 0) order[block[ci]] = next_id++;
    if (repr[order[block[ci]]] < 0) repr[order[block[ci]]] = ci;
  }
  static const int64_t cap_entries = getenv("MRX_TRIES_CAP_ENTRIES") ? atoll(getenv("MRX_TRIES_CAP_ENTRIES")) : 1300;   // (A/B runs)
  // 32-byte entries, 41 KB at most, and no more than the plan's blob has room for (60 KiB for everything the generic kernels
  // stage: checked where the table is stored).  Small tables gain 3-10 x over marks + stepper; tables of 16-41 KB run at
  // about half their speed (two workgroups per CU) and win two times out of three (tools/r04_tries_cap.py); on the
  // reference's phone patterns by pieces the form is slower (flexible_phone 269 -> 195 GB/s).  Plans without empty matches
  // are therefore MEASURED per handle against marks + stepper (ab_tuner_begin).
  if ((int64_t)nblocks * (ncp + 1) > cap_entries) { why = "pending-tries walk: configuration table beyond 41 KB"; return false; }
  if ((((uint64_t)nblocks << ew.cshift) >> 16) != 0) { why = "pending-tries walk: configuration table beyond the LDS budget"; return false; }
  ew.ncfg = nblocks;
  ew.tab.assign((size_t)ew.ncfg * ncp, EwEntry{});
  ew.end.assign(ew.ncfg, EwEntry{});
  for (int b = 0; b < nblocks; ++b) {
    const int ci = repr[b];
    for (int k = 0; k < ew.ncls; ++k) {
      EwEntry e = rows[ci][k];
      e.x |= (((uint32_t)order[block[nexts[ci][k]]] << ew.cshift) & 0xFFFFu) << 16;
      ew.tab[(size_t)b * ncp + k] = e;
    }
    ew.end[b] = rows[ci][ncp];
  }
  return true;
}
// ---- where do matches begin?  (backward table, DevPlan::off_bk_*) -----------------------------------------------------
// The restart-per-position search spends its time on walks that fail.  Whether the walk from position s succeeds
// does not depend on the search's history -- only on the text from s on -- so it can be known for EVERY s before the
// search runs: scan the text right to left keeping B(s) = { q : some prefix of text[s:] leads q into an accepting
// state }, B(len) = accepting states, B(s) = accepting states + { q : next[q][text[s]] in B(s + 1) }; the walk from
// s succeeds iff text[s] may start a walk and next[start][text[s]] lies in B(s + 1).  The sets are numbered here
// (subset construction over the reversed table), the device keeps a set number and one table lookup per byte.
// With the marks the forward search starts a walk only where one succeeds: `pos + 1` restarts disappear, what is
// left of the reference's loop is "next mark at or behind pos, longest walk from it, pos = its end".
struct BackSet {
  std::array<uint8_t, 256> cls{};
  int ncls = 0, cshift = 0, nsub = 0, start = 0;
  std::vector<uint16_t> tab;
};

bool build_backset(const SearchAutomaton& s, BackSet& bk, std::string& why) {
  if (s.acc[0]) { why = "start state accepts (empty matches)"; return false; }
  if (s.n > 256) { why = "backward table: more than 256 states"; return false; }
  std::vector<int> rep;
  {
    std::map<std::vector<int>, int> seen;
    for (int c = 0; c < 256; ++c) {
      std::vector<int> col(s.n + 1);
      for (int q = 0; q < s.n; ++q) col[q] = s.next[q][c];
      col[s.n] = s.allowed[c];
      auto it = seen.find(col);
      if (it == seen.end()) { it = seen.emplace(col, (int)rep.size()).first; rep.push_back(c); }
      bk.cls[c] = (uint8_t)it->second;
    }
  }
  bk.ncls = (int)rep.size();
  bk.cshift = 0;
  while ((1 << bk.cshift) < bk.ncls) ++bk.cshift;
  const int ncp = 1 << bk.cshift;
  using Set = std::vector<uint8_t>;   // membership per DFA state
  std::map<Set, int> ids;
  std::vector<Set> sets;
  auto id_of = [&](const Set& b) {
    auto it = ids.find(b);
    if (it == ids.end()) { it = ids.emplace(b, (int)sets.size()).first; sets.push_back(b); }
    return it->second;
  };
  Set accs(s.n, 0);
  for (int q = 0; q < s.n; ++q) accs[q] = s.acc[q] ? 1 : 0;
  bk.start = id_of(accs);
  std::vector<std::vector<uint16_t>> rows;
  for (size_t si = 0; si < sets.size(); ++si) {
    if ((int64_t)sets.size() * ncp > 16384 || sets.size() > 32000) { why = "backward table beyond the LDS budget"; return false; }
    const Set cur = sets[si];
    std::vector<uint16_t> row(ncp, 0);
    for (int k = 0; k < bk.ncls; ++k) {
      const int c = rep[k];
      Set nb = accs;
      for (int q = 0; q < s.n; ++q) {
        const int t = s.next[q][c];
        if (t >= 0 && cur[t]) nb[q] = 1;
      }
      const int t0 = s.allowed[c] ? s.next[0][c] : -1;
      const int mark = (t0 >= 0 && cur[t0]) ? 1 : 0;
      row[k] = (uint16_t)((id_of(nb) << 1) | mark);
    }
    rows.push_back(row);
  }
  bk.nsub = (int)sets.size();
  if ((int64_t)bk.nsub * ncp > 16384) { why = "backward table beyond the LDS budget"; return false; }
  bk.tab.assign((size_t)bk.nsub * ncp, 0);
  for (int si = 0; si < bk.nsub; ++si)
    for (int k = 0; k < ncp; ++k) bk.tab[(size_t)si * ncp + k] = rows[si][k];
  return true;
}

// The required-byte route (HybridMatcher._match_all_required_byte, matcher.mojo:864-898) in the same table form:
//   pos = 0; while pos < len: hit = next required byte at or behind pos; start = hit backed up over first-class
//   bytes; m = DFAEngine.match_first(text, start); if m and m.end > hit: report, pos = m.end; else pos = hit + 1
// The walks that can ever be asked for begin where a run of first-class bytes begins (or on a required byte that
// follows none), so ONE speculative walk accompanies every run (the last slot of the list); the required byte that
// ends the run ACTIVATES it (it joins the list of pending walks, oldest first, keeping its start register -- a
// change of role, not of place), any other byte behind the run discards it.  Only what a walk accepts from its
// hit on counts (m.end > hit).  Among the pending walks the rules are those of build_multiwalk(): the oldest one
// that has accepted behind its hit makes every younger pending walk void the moment it accepts (their hits lie
// inside its match, the reference's `pos = m.end` skips them) and reports when it dies; a pending walk that dies
// without having accepted behind its hit just leaves (`pos = hit + 1`).  The quirk that a start may lie inside the
// previous match (the back-up is not bounded by pos) needs nothing: the speculative walk began there anyway.
// Left to the stepper: a younger pending walk that accepts while an older one is undecided, more than four slots.
bool build_reqwalk(const SearchAutomaton& s, int needle, const std::array<uint8_t, 256>& first, MultiWalk& mw,
                   std::string& why) {
  constexpr int K = 4;
  if (s.acc[0]) { why = "start state accepts"; return false; }
  if (needle < 0 || first[needle]) { why = "required byte inside the first class"; return false; }
  std::vector<int> rep;
  {
    std::map<std::vector<int>, int> seen;
    for (int c = 0; c < 256; ++c) {
      std::vector<int> col(s.n + 2);
      for (int q = 0; q < s.n; ++q) col[q] = s.next[q][c];
      col[s.n] = first[c];
      col[s.n + 1] = c == needle;
      auto it = seen.find(col);
      if (it == seen.end()) { it = seen.emplace(col, (int)rep.size()).first; rep.push_back(c); }
      mw.cls[c] = (uint8_t)it->second;
    }
  }
  mw.ncls = (int)rep.size();
  if (mw.ncls > 64) { why = "multi-walk: more than 64 byte classes"; return false; }
  mw.cshift = 0;
  while ((1 << mw.cshift) < mw.ncls) ++mw.cshift;
  const int ncp = 1 << mw.cshift;
  // (A: the oldest pending walk has accepted behind its hit; P: the previous byte was a first-class byte;
  //  pending walks oldest first; the run's speculative walk, -1 = none / dead)
  struct Cfg { int A, P; std::vector<int> act; int spec; };
  auto key = [](const Cfg& c) { std::vector<int> k{c.A, c.P, c.spec}; k.insert(k.end(), c.act.begin(), c.act.end()); return k; };
  std::map<std::vector<int>, int> ids;
  std::vector<Cfg> cfgs;
  auto id_of = [&](const Cfg& c) {
    auto k = key(c);
    auto it = ids.find(k);
    if (it == ids.end()) { it = ids.emplace(k, (int)cfgs.size()).first; cfgs.push_back(c); }
    return it->second;
  };
  id_of(Cfg{0, 0, {}, -1});
  std::vector<std::array<uint32_t, 64>> rows;
  for (size_t ci = 0; ci < cfgs.size(); ++ci) {
    if ((int64_t)cfgs.size() * ncp > 8192) { why = "multi-walk: configuration table beyond the LDS budget"; return false; }
    const Cfg cur = cfgs[ci];
    std::array<uint32_t, 64> row{};
    const int n_act = (int)cur.act.size();
    const int spec_slot = n_act;   // old slot of the speculative walk
    for (int k = 0; k < mw.ncls; ++k) {
      const int b = rep[k];
      const bool isF = first[b] != 0, isN = b == needle;
      std::vector<std::pair<int, int>> lst;   // pending walks: (state, provenance: old slot, -1 = begins on this byte)
      bool emit = false;
      int A = cur.A;
      for (int i = 0; i < n_act; ++i) {
        const int t = s.next[cur.act[i]][b];
        if (t < 0) {
          if (i == 0) { emit = A != 0; A = 0; }
          continue;
        }
        if (t == 0) { why = "transition back into the start state"; return false; }
        lst.push_back({t, i});
      }
      if (n_act > 0 && (lst.empty() || lst[0].second != 0)) A = 0;
      const int spec_t = cur.spec >= 0 ? s.next[cur.spec][b] : -1;
      const int fresh = s.next[0][b];   // a walk that begins on this byte
      if (spec_t == 0 || fresh == 0) { why = "transition back into the start state"; return false; }
      int nspec = -1, nspec_prov = 0, nP = 0;
      if (isF) {
        nP = 1;
        if (!cur.P) { nspec = fresh; nspec_prov = -1; }
        else { nspec = spec_t; nspec_prov = spec_slot; }
      } else if (isN) {
        if (cur.P) { if (spec_t >= 0) lst.push_back({spec_t, spec_slot}); }
        else if (fresh >= 0) lst.push_back({fresh, -1});   // no run in front of the hit: the walk begins on it
      }
      std::vector<std::pair<int, int>> m;
      for (const auto& x : lst) {
        bool dup = false;
        for (const auto& y : m) dup = dup || y.first == x.first;
        if (!dup) m.push_back(x);
      }
      bool acc_now = false;
      if (!m.empty() && s.acc[m[0].first]) { acc_now = true; A = 1; m.resize(1); }
      for (size_t j = 1; j < m.size(); ++j)
        if (s.acc[m[j].first]) { why = "a later hit's walk accepts while an earlier one is still undecided"; return false; }
      const int total = (int)m.size() + (nspec >= 0 ? 1 : 0);
      if (total > K) { why = "multi-walk: more than four walks at a time"; return false; }
      mw.kmax = std::max(mw.kmax, total);
      Cfg nc{m.empty() ? 0 : A, nP, {}, nspec};
      for (const auto& x : m) nc.act.push_back(x.first);
      const int nid = id_of(nc);
      if ((((uint64_t)nid << mw.cshift) >> 16) != 0) { why = "multi-walk: configuration table beyond the LDS budget"; return false; }
      uint32_t e = ((uint32_t)nid << mw.cshift) << 16;
      if (emit) e |= 1u;
      if (acc_now) e |= 2u;
      if (nc.A) e |= 1u << 10;
      static const int shift[4] = {2, 5, 7, 9};
      int j = 0;
      for (; j < (int)m.size(); ++j) e |= (uint32_t)(m[j].second < 0 ? K - j : m[j].second - j) << shift[j];
      if (nspec >= 0) e |= (uint32_t)(nspec_prov < 0 ? K - j : nspec_prov - j) << shift[j];
      row[k] = e;
    }
    rows.push_back(row);
  }
  mw.ncfg = (int)cfgs.size();
  mw.tab.assign((size_t)mw.ncfg * ncp, 0);
  for (int ci = 0; ci < mw.ncfg; ++ci)
    for (int k = 0; k < mw.ncls; ++k) mw.tab[(size_t)ci * ncp + k] = rows[ci][k];
  return true;
}

void put(std::vector<uint8_t>& blob, const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  blob.insert(blob.end(), b, b + n);
}
void align(std::vector<uint8_t>& blob, size_t a) {
  while (blob.size() % a) blob.push_back(0);
}

}  // namespace

// The table run on the host (tests: against the oracle, before any kernel sees it): findall of one text.
std::vector<std::pair<int, int>> emptywalk2_run(const EmptyWalk2& ew, const uint8_t* text, int len) {
  std::vector<std::pair<int, int>> out;
  uint32_t row = 0;
  int s0 = 0, last = 0;
  auto apply = [&](const EwEntry& e, int base) {
    if (e.x & 1u) out.push_back({s0, last});
    const int nrep = (int)((e.x >> 10) & 31);
    for (int r = 0; r < nrep; ++r) {
      const int f = (int)((e.r[r / 4] >> (8 * (r % 4))) & 255);
      const int st = base - (f & 15);
      out.push_back({st, st + (f >> 4)});
    }
    const int ta = (int)((e.x >> 2) & 15);
    if (ta) { s0 = base - (ta - 1); last = s0 + (int)((e.x >> 6) & 15); }
    else if (e.x & 2u) last = base + 1;
  };
  for (int p = 0; p < len; ++p) {
    const EwEntry& e = ew.tab[(size_t)row + ew.cls[text[p]]];
    apply(e, p);
    row = e.x >> 16;
  }
  apply(ew.end[row >> ew.cshift], len);
  if (ew.empty) out.push_back({len, len});
  return out;
}


bool repl_has_group_refs(const std::string& r) {  // matcher.mojo:1472-1482
  for (size_t i = 0; i + 1 < r.size(); ++i)
    if (r[i] == '\\' && r[i + 1] >= '1' && r[i + 1] <= '9') return true;
  return false;
}

std::vector<ReplSeg> parse_repl_template(const std::string& r) {  // matcher.mojo:1436-1469
  std::vector<ReplSeg> segs;
  const int n = (int)r.size();
  int i = 0, lit_start = 0;
  while (i < n) {
    if (r[i] == '\\' && i + 1 < n) {
      const char nc = r[i + 1];
      if (nc >= '1' && nc <= '9') {
        if (i > lit_start) segs.push_back({0, lit_start, i - lit_start});
        segs.push_back({nc - '0', 0, 0});
        i += 2;
        lit_start = i;
        continue;
      }
    }
    ++i;
  }
  if (lit_start < n) segs.push_back({0, lit_start, n - lit_start});
  return segs;
}

void build_plan(const std::string& pattern, HostPlan& hp, bool force_nfa, bool force_bitset, bool nfa_engine,
                bool dfa_engine) {
  hp = HostPlan();
  hp.force_nfa = force_nfa;
  hp.force_bitset = force_bitset;
  hp.nfa_engine = nfa_engine;
  hp.dfa_engine = dfa_engine;
  hp.pattern = pattern;
  // nfa_engine: the Engine is NFAEngine itself (engine.mojo:4-37, nfa.mojo:66-143), as regex.nfa's module
  // functions build it (nfa.mojo:1733-1769): no HybridMatcher in front, so none of its shortcuts ('.*',
  // exact literal, memchr prefilter, required byte, fixed-width groups), no DFA, LazyDFA or OnePass.
  // dfa_engine: the Engine is the DFAEngine that compile_dfa_pattern(parse(pattern)) returns, as the comptime
  // API (comptime_regex.mojo:59-87, 176-233) and the reference's tests/test_dfa.mojo use it: again no router.
  hp.wildcard_any = (pattern == ".*") && !nfa_engine && !dfa_engine;  // matcher.mojo:435-444, 573-591

  // fixed-width capture groups, CompiledRegex._try_precompute_fixed_sub (:1002-1035)
  {
    std::vector<int> segs;
    if (!nfa_engine && !dfa_engine && detect_fixed_width(pattern, segs)) {
      int ng = 0, total = 0;
      bool lits = false, ok = true;
      int off[10] = {0}, w[10] = {0};
      for (int s : segs) {
        if (s > 0) {
          if (++ng > 9) { ok = false; break; }
          off[ng] = total; w[ng] = s; total += s;
        } else { lits = true; total += -s; }
      }
      if (ok && ng > 0) {
        hp.fixed_total = total; hp.fixed_ngroups = ng; hp.fixed_concat = !lits;
        // nothing but (\d{N}) / (\d) groups, end to end: only then is the whole-text shortcut of
        // _sub_impl_with_repl (matcher.mojo:1726-1744: a text of exactly this many bytes is rewritten iff it is
        // all digits, no engine asked) the same as matching.  Literals behind the last group are not counted
        // upstream ('(\\d)*a' has fixed_total 1 and "concat"), so such patterns take the shortcut on texts they
        // would match differently.
        {
          size_t i = 0;
          bool pure = !lits;
          while (pure && i < pattern.size()) {
            if (pattern.compare(i, 3, "(\\d") != 0) { pure = false; break; }
            i += 3;
            if (i < pattern.size() && pattern[i] == '{') {
              const size_t close = pattern.find('}', i);
              if (close == std::string::npos) { pure = false; break; }
              i = close + 1;
            }
            if (i >= pattern.size() || pattern[i] != ')') { pure = false; break; }
            ++i;
          }
          hp.fixed_pure = pure;
        }
        std::memcpy(hp.fixed_off, off, sizeof off);
        std::memcpy(hp.fixed_w, w, sizeof w);
      }
    }
  }

  Ast ast;
  if (!hp.wildcard_any) {
    parse(pattern, ast);  // may throw SyntaxError
    hp.complexity = classify(ast);
    hp.use_pure_dfa = should_use_pure_dfa(ast);
    const bool analyze = !skip_prefilter(pattern) && !hp.use_pure_dfa && !nfa_engine && !dfa_engine;
    if (analyze) {  // matcher.mojo:609-654
      LiteralSet ls = extract_literals(ast);
      const bool anchors = ast_has_anchors(ast, ast.root);
      const LiteralInfo* best = ls.best_literal();
      bool exact = false;
      if (best && !best->literal.empty()) {
        hp.best_literal = best->literal;
        const bool ops = pattern.find_first_of("*+?.|([{") != std::string::npos;
        exact = best->is_required && has_literal_prefix(ast) && !anchors && !ops;
      }
      hp.literal_has_anchors = anchors;
      hp.exact_literal = exact;
      if (pattern.find('|') == std::string::npos && hp.best_literal.size() >= 2) {
        hp.has_prefilter = true;  // create_optimized_prefilter, matcher.mojo:99-108
        hp.prefilter_literal = hp.best_literal;
      }
    }
    // NFAMatcher (always built), matcher.mojo:301-322 + NFAEngine flags, nfa.mojo:86-143
    {
      LiteralSet ls = extract_literals(ast);
      if (const LiteralInfo* b = ls.best_literal()) {
        if (b->is_prefix && b->is_required && b->literal.size() >= 1) hp.nfa_has_literal_opt = true;
        else if (b->is_required && b->literal.size() >= 3) hp.nfa_has_literal_opt = true;
        if (hp.nfa_has_literal_opt) hp.nfa_literal = b->literal;
      }
      auto ends = [&](const char* s) {
        const size_t k = std::strlen(s);
        return pattern.size() >= k && pattern.compare(pattern.size() - k, k, s) == 0;
      };
      hp.nfa_ends_dotstar = ends(".*") && !ends("\\.*");
      if (pattern.compare(0, 2, ".*") == 0) {
        hp.nfa_starts_dotstar = true;
        if (pattern.size() > 2 && (pattern[2] == '?' || pattern[2] == '*' || pattern[2] == '+'))
          hp.nfa_starts_dotstar = false;
      }
      build_bt(ast, hp.bt);
      compile_program(ast, hp.program);
      build_lazy(hp.program, hp.lazy, /*max_dfa_states=*/4096);
      build_bitset(hp.program, hp.bitset);
      // NFAMatcher.__init__, matcher.mojo:310-313: OnePass only for programs with '$'
      if (hp.lazy.supported && hp.program.has_end_anchor()) build_onepass(hp.program, hp.onepass);
    }
    if (dfa_engine) {   // whatever the classifier says; a pattern no shape compiler takes has no DFAEngine
      compile_dfa_pattern(ast, hp.dfa);   // DfaCompileError -> MRX_E_UNSUPPORTED with its message
      hp.use_dfa = true;
    } else
    if (hp.complexity == CX_SIMPLE && !force_nfa && !nfa_engine) {  // matcher.mojo:664-675
      try {
        compile_dfa_pattern(ast, hp.dfa);
        hp.use_dfa = true;
      } catch (const DfaCompileError&) {
        hp.use_dfa = false;
      }
    }
    if (hp.use_dfa && !dfa_engine && !hp.literal_has_anchors && hp.dfa.has_matcher)
      hp.required_byte = rare_required_byte(ast, hp.dfa.matcher.lookup);
  }

  // get_engine_type / get_stats, matcher.mojo:900-918, 1139-1163
  hp.engine_type = hp.use_dfa ? "DFA" : "NFA";
  if (nfa_engine) hp.engine_type = "NFAEngine";
  if (dfa_engine) hp.engine_type = "DFAEngine";
  if (hp.exact_literal && !hp.literal_has_anchors) hp.engine_type += "+ExactLiteral";
  else if (hp.has_prefilter && !hp.literal_has_anchors) hp.engine_type += "+Prefilter";
  static const char* cxn[] = {"SIMPLE", "MEDIUM", "COMPLEX"};
  hp.stats = "Pattern: '" + pattern + "', Engine: " + hp.engine_type +
             ", Complexity: " + cxn[hp.complexity];

  // ---- which operations stay on the hot path ----------------------------------
  // a LazyDFA whose determinisation exceeds the budget still runs, on the bitset NFA
  const bool use_bitset = hp.lazy.supported && !hp.lazy.start_dead && hp.bitset.ok &&
                          (hp.lazy.too_large || hp.force_bitset);
  const bool too_large = hp.lazy.too_large && !use_bitset;
  const bool lazy_ok = hp.lazy.supported && !too_large;
  const bool nfa_end = hp.program.has_end_anchor();
  bool lazy_end = false;   // '$' program on the LazyDFA search, per-text transition cache (below)
  bool bt_first = false, bt_search = false;   // operations the reference sends to NFAEngine's backtracking matcher
  if (nfa_engine) {
    // NFAEngine.match_first / match_next / match_all (nfa.mojo:169-498) and nothing else
    if (hp.bt.ok) bt_first = bt_search = true;
    else
      hp.why_no_match_first = hp.why_no_search =
          "NFAEngine selected as the engine; its flat-program form does not cover: " + hp.bt.why_not;
  } else if (!hp.wildcard_any && !hp.use_dfa) {
    // NFAMatcher.match_first, matcher.mojo:361-380
    if (lazy_ok && nfa_end && hp.onepass.ok)
      hp.first_onepass = true;  // matcher.mojo:378-379
    else if (!(lazy_ok && !nfa_end)) {
      // the reference falls through to NFAEngine.match_first (matcher.mojo:380): served by the flat
      // program of the backtracking matcher when the pattern has one
      if (!too_large && hp.bt.ok) bt_first = true;
      else
      hp.why_no_match_first = too_large
          ? "LazyDFA determinisation exceeds the state budget"
          : "reference routes match_first to the backtracking NFA ('$' in an NFA-routed pattern that is "
            "not one-pass); its flat-program form does not cover: " + hp.bt.why_not;
    }
    // NFAMatcher.match_next / match_all, matcher.mojo:383-431
    const bool fast_absent = !hp.nfa_has_literal_opt && !hp.nfa_starts_dotstar && !hp.nfa_ends_dotstar;
    if (!(hp.lazy.supported && fast_absent)) {
      if (hp.bt.ok) bt_search = true;   // NFAEngine.match_next / match_all (matcher.mojo:419, 431)
      else
      hp.why_no_search = "reference routes search/findall to the backtracking NFA "
                         "(literal prefilter or leading/trailing .* fast path); its flat-program form does "
                         "not cover: " + hp.bt.why_not;
    }
    else if (nfa_end) {
      // Upstream the answer depends on what EARLIER calls left in the transition cache (pikevm.mojo:697-700).  A
      // batch has no call order: every text is answered as a freshly compiled pattern would answer it (the cache
      // empty when the text's call begins, carried through the call's walks as upstream) -- the generic kernels
      // keep, per text, which (state, last byte) pairs were first computed where (walk_lazy_end, mrx_device.hpp).
      // Needs the determinised states of both transition variants to fit one 64-bit mask.
      if (!use_bitset && !too_large && hp.lazy.has_end_variant && hp.lazy.trans.size() <= 64) lazy_end = true;
      hp.why_no_search = lazy_end ? "(internal: '$' on the LazyDFA search, cleared at the end of build_plan)"
                                  : "LazyDFA search with '$': the per-text transition cache is tracked for at most 64 "
                                    "determinised states (pikevm.mojo:697-700)";
    }
    else if (too_large)
      hp.why_no_search = "LazyDFA determinisation exceeds the state budget and the program has more "
                         "than 256 positions";
  }
  if (hp.use_dfa && hp.dfa.has_matcher && hp.dfa.matcher.num_ranges == 0 && !hp.dfa.scan_eligible) {
    // nibble-table false positives (simd_ops.mojo:63-134) are SIMD-width dependent
    // upstream; they are result-neutral unless a false-positive byte can start a
    // walk or the start state accepts.
    const ClassMatcher& m = hp.dfa.matcher;
    for (int x = 0; x < 256; ++x)
      if (!m.lookup[x] && m.nibble_hit(x) && (hp.dfa.accepting[0] || hp.dfa.trans[0][x] != -1)) {
        hp.why_no_search = "first-class nibble-table false positives change results "
                           "(depends on the reference build's SIMD width)";
        break;
      }
  }

  // the exact-literal bypass answers search/findall/sub before any engine is consulted
  // (matcher.mojo:768-781, 815-847)
  if (hp.exact_literal && !hp.literal_has_anchors) hp.why_no_search.clear();

  // ---- device payload ------------------------------------------------------------
  DevPlan& d = hp.dev;
  d = DevPlan();
  d.required_byte = -1;
  d.fixed_total = hp.fixed_total; d.fixed_ngroups = hp.fixed_ngroups;
  d.fixed_concat = hp.fixed_concat ? 1 : 0;
  std::memcpy(d.fixed_off, hp.fixed_off, sizeof d.fixed_off);
  std::memcpy(d.fixed_w, hp.fixed_w, sizeof d.fixed_w);
  std::vector<std::array<int, 256>> T;  // transition rows (int, -1 dead)
  std::vector<uint8_t> acc;
  std::array<uint8_t, 256> first{};
  std::string lit, exact_lit, pre;  // engine literal, exact-literal bypass, prefilter literal
  if (hp.wildcard_any) {
    d.kind = PLAN_ANY;
  } else if (hp.use_dfa) {
    d.kind = PLAN_DFA;
    const DfaEngine& e = hp.dfa;
    for (int s = 0; s < e.nstates(); ++s) {
      std::array<int, 256> row;
      for (int c = 0; c < 256; ++c) row[c] = e.trans[s][c];
      T.push_back(row);
      acc.push_back(e.accepting[s]);
    }
    if (e.has_start_anchor) d.flags |= PF_START_ANCHOR;
    if (e.has_end_anchor) d.flags |= PF_END_ANCHOR;
    if (e.is_pure_literal) { d.flags |= PF_PURE_LITERAL; lit = e.literal; }
    if (e.has_matcher) { d.flags |= PF_HAS_MATCHER; first = e.matcher.lookup; }
    if (e.scan_eligible) d.flags |= PF_SCAN_ELIGIBLE;
    if (!acc.empty() && acc[0]) d.flags |= PF_START_ACCEPTING;
    if (hp.exact_literal && !hp.literal_has_anchors) { d.flags |= PF_EXACT_LITERAL; exact_lit = hp.best_literal; }
    else if (hp.has_prefilter && !hp.literal_has_anchors) { d.flags |= PF_PREFILTER; pre = hp.prefilter_literal; }
    d.required_byte = hp.required_byte;
  } else {
    d.kind = PLAN_LAZY;
    if (bt_first) d.flags |= PF_BT_FIRST;
    if (bt_search) d.flags |= PF_BT_SEARCH;
    const LazyTables& z = hp.lazy;
    if (z.start_dead || !lazy_ok) d.flags |= PF_START_DEAD;
    if (use_bitset) d.flags |= PF_BITSET;
    for (size_t s = 0; s < z.trans.size() && !use_bitset; ++s) {
      std::array<int, 256> row;
      for (int c = 0; c < 256; ++c) row[c] = z.trans[s][c];
      T.push_back(row);
      acc.push_back(z.is_match[s]);
    }
    if (lazy_end) {   // rows nstates/2 ..: the transitions as computed while the text's last byte is consumed
      for (size_t s = 0; s < z.trans_end.size(); ++s) {
        std::array<int, 256> row;
        for (int c = 0; c < 256; ++c) row[c] = z.trans_end[s][c];
        T.push_back(row);
        acc.push_back(z.is_match[s]);
      }
      d.flags |= PF_LAZY_END;
    }
    if (use_bitset) {  // the table is not used; only "does the start set accept" survives
      std::array<int, 256> row; row.fill(-1); T.push_back(row);
      acc.push_back(!z.is_match.empty() && z.is_match[0]);
    }
    if (T.empty()) { std::array<int, 256> row; row.fill(-1); T.push_back(row); acc.push_back(0); }
    if (z.has_filter) { d.flags |= PF_HAS_MATCHER; first = z.first_byte; }
    if (acc[0]) d.flags |= PF_START_ACCEPTING;
    if (hp.exact_literal && !hp.literal_has_anchors) { d.flags |= PF_EXACT_LITERAL; exact_lit = hp.best_literal; }
    else if (hp.has_prefilter && !hp.literal_has_anchors) { d.flags |= PF_PREFILTER; pre = hp.prefilter_literal; }
  }
  if (T.empty()) { std::array<int, 256> row; row.fill(-1); T.push_back(row); acc.push_back(0); }

  // byte classes: bytes with identical columns share a class
  std::array<uint8_t, 256> cls{};
  int ncls = 0;
  {
    std::map<std::vector<int>, int> seen;
    for (int c = 0; c < 256; ++c) {
      std::vector<int> col(T.size());
      for (size_t s = 0; s < T.size(); ++s) col[s] = T[s][c];
      auto it = seen.find(col);
      if (it == seen.end()) it = seen.emplace(col, ncls++).first;
      cls[c] = (uint8_t)it->second;
    }
  }
  d.nstates = (int)T.size();
  d.ncls = ncls;
  // the exact-literal bypass (matcher.mojo:768-781, 815-847) replaces the engine
  // entirely, so its literal can share the slot of the engine literal
  if (d.flags & PF_EXACT_LITERAL) lit = exact_lit;
  d.lit_len = (int)lit.size();
  d.pre_len = (int)pre.size();
  hp.blob.clear();
  d.off_cls = 0; put(hp.blob, cls.data(), 256);
  d.off_first = 256; put(hp.blob, first.data(), 256);
  d.off_trans = (int)hp.blob.size();
  {
    // u16 entries: bit 15 = target accepts, low 15 bits = target, 0xFFFF = dead
    std::vector<uint16_t> tr((size_t)d.nstates * ncls, 0xFFFF);
    for (int c = 0; c < 256; ++c)
      for (int s = 0; s < d.nstates; ++s) {
        const int t = T[s][c];
        if (t >= 0) tr[(size_t)s * ncls + cls[c]] = (uint16_t)(t | (acc[t] ? 0x8000 : 0));
      }
    put(hp.blob, tr.data(), tr.size() * 2);
  }
  d.off_lit = (int)hp.blob.size();
  put(hp.blob, lit.data(), lit.size());
  d.off_pre = (int)hp.blob.size();
  put(hp.blob, pre.data(), pre.size());
  align(hp.blob, 8);
  if (d.flags & PF_BITSET) {
    const BitsetNfa& b = hp.bitset;
    d.bs_nw = b.nw; d.bs_npos = b.npos;
    for (int w = 0; w < kBitsetWords; ++w) { d.bs_start[w] = b.start[w]; d.bs_match[w] = b.match[w]; }
    std::array<uint8_t, 256> bcls{};
    std::vector<std::array<uint64_t, kBitsetWords>> masks;
    for (int c = 0; c < 256; ++c) {
      size_t k = 0;
      while (k < masks.size() && masks[k] != b.byte_mask[c]) ++k;
      if (k == masks.size()) masks.push_back(b.byte_mask[c]);
      bcls[c] = (uint8_t)k;   // at most 256 distinct columns
    }
    d.bs_ncls = (int)masks.size();
    d.off_bs_cls = (int)hp.blob.size(); put(hp.blob, bcls.data(), 256);
    d.off_bs_mask = (int)hp.blob.size();
    for (const auto& m : masks) put(hp.blob, m.data(), 8 * b.nw);
    d.off_bs_follow = (int)hp.blob.size();
    for (const auto& f : b.follow) put(hp.blob, f.data(), 8 * b.nw);
    // Do all matches have one length?  Layer the positions by the number of bytes consumed before them: a start
    // position is at depth 0, every follower of a position at depth k at k + 1; one position at two depths (any
    // loop, any optional part) or MATCH at two depths means no.  With one length L the restart-per-position loop
    // (pikevm.mojo:755-867 under matcher.mojo:1341-1354) takes a match end e iff e - L is not inside the match
    // taken before it, which the union pass can decide as it goes (k_bscan modes 2-4).
    d.bs_fixed_len = 0;
    if (b.nw == 1) {
      std::vector<int> depth(b.npos, -1);
      std::vector<int> queue;
      bool fixed = (b.start[0] & b.match[0]) == 0;
      for (int q = 0; q < b.npos; ++q)
        if ((b.start[0] >> q) & 1) { depth[q] = 0; queue.push_back(q); }
      int L = -1;
      for (size_t qi = 0; qi < queue.size() && fixed; ++qi) {
        const int q = queue[qi];
        if ((b.match[0] >> q) & 1) {
          if (L >= 0 && L != depth[q]) fixed = false;
          L = depth[q];
          continue;   // MATCH consumes nothing
        }
        for (int r = 0; r < b.npos && fixed; ++r)
          if ((b.follow[q][0] >> r) & 1) {
            if (depth[r] < 0) { depth[r] = depth[q] + 1; queue.push_back(r); }
            else if (depth[r] != depth[q] + 1) fixed = false;
          }
      }
      if (fixed && L >= 1 && L < 32768) d.bs_fixed_len = L;
    }
  }
  if (d.nstates >= 0x7FFF) {
    hp.why_no_match_first = hp.why_no_search = "more than 32766 DFA states";
  }
  d.bt_nitems = 0; d.off_bt_items = d.off_bt_tbl = d.off_bt_lit = -1; d.bt_ngroups = 0; d.bt_lit_len = 0; d.bt_flags = 0;
  d.bt_pattern_len = (int)pattern.size();
  if (hp.bt.ok) {
    align(hp.blob, 4);
    d.bt_nitems = (int)hp.bt.items.size();
    d.bt_ngroups = hp.bt.ngroups;
    d.off_bt_items = (int)hp.blob.size();
    put(hp.blob, hp.bt.items.data(), hp.bt.items.size() * sizeof(BtItem));
    d.off_bt_tbl = (int)hp.blob.size();
    for (const auto& t : hp.bt.tables) put(hp.blob, t.data(), 32);
    d.off_bt_lit = (int)hp.blob.size();
    d.bt_lit_len = (int)hp.nfa_literal.size();
    put(hp.blob, hp.nfa_literal.data(), hp.nfa_literal.size());
    if (hp.nfa_has_literal_opt) d.bt_flags |= 1;
    if (hp.nfa_starts_dotstar) d.bt_flags |= 4;
    if (hp.nfa_ends_dotstar) d.bt_flags |= 8;
    if (!hp.nfa_literal.empty() && pattern.compare(0, hp.nfa_literal.size(), hp.nfa_literal) == 0) d.bt_flags |= 2;
    // bit 5: a deterministic chain.  No ALT / LOOP / FAIL items, and behind every quantified leaf that can leave a
    // choice (not the last child of its sequence) comes -- past group boundaries -- an anchor, or a leaf that must
    // consume a byte (min >= 1) none of whose three membership tables shares a byte with the quantified leaf's
    // is_match_char table: the leaf behind then fails on every byte the quantified leaf could give back, so a
    // shorter count never rescues the sequence and the first failure is final (k_match<., 2> etc. run without
    // the choice stack).
    {
      bool chain = true;
      const auto& items = hp.bt.items;
      const auto& tbls = hp.bt.tables;
      for (size_t i = 0; i < items.size() && chain; ++i) {
        const BtItem& it = items[i];
        if (it.kind >= BT_ALT) { chain = false; break; }
        if (it.kind != BT_LEAF || !(it.flags & BTF_QUANT) || (it.flags & BTF_LAST)) continue;
        if (it.min == it.max) continue;   // a fixed count leaves no choice
        size_t j = i + 1;
        while (j < items.size() && (items[j].kind == BT_OPEN || items[j].kind == BT_CLOSE)) ++j;
        if (j >= items.size()) { chain = false; break; }   // (cannot happen: a non-last leaf has a sibling behind it)
        const BtItem& nx = items[j];
        if (nx.kind == BT_START || nx.kind == BT_END) continue;
        if (nx.kind != BT_LEAF || nx.min < 1) { chain = false; break; }
        for (int b = 0; b < 32 && chain; ++b) {
          const uint8_t mine = tbls[it.tbl * 3 + 1][b];
          const uint8_t theirs = tbls[nx.tbl * 3 + 0][b] | tbls[nx.tbl * 3 + 1][b] | tbls[nx.tbl * 3 + 2][b];
          if (mine & theirs) chain = false;
        }
      }
      if (chain) d.bt_flags |= 32;
    }
    // "chain groups": are NFAEngine.match_next_with_groups' matches (nfa.mojo:500-574: every start in turn, the greedy
    // count of every leaf, the first failure final) the matches of the plain table walk?  Decided on the tables: the
    // chain's own automaton -- state (leaf, bytes it has taken) -- against the engine's, pair by pair.
    {
      ChainGroups& cg = hp.chain;
      cg = ChainGroups();
      for (int g = 0; g < 10; ++g) cg.gopen[g] = cg.gclose[g] = -1;
      const auto& items = hp.bt.items;
      const auto& tbls = hp.bt.tables;
      std::string why;
      if (!(d.bt_flags & 32)) why = "not a deterministic chain";
      else if (d.bt_flags & 1) why = "NFAEngine literal prefilter";
      else if ((d.kind != PLAN_DFA && d.kind != PLAN_LAZY) ||
               (d.flags & (PF_START_ANCHOR | PF_END_ANCHOR | PF_PREFILTER | PF_EXACT_LITERAL | PF_PURE_LITERAL | PF_SCAN_ELIGIBLE |
                           PF_BITSET | PF_LAZY_END | PF_START_DEAD)))
        why = "the search is not a plain walk of a DFAEngine / LazyDFA table";
      std::vector<int> ltbl;
      for (size_t i = 0; i < items.size() && why.empty(); ++i) {
        const BtItem& it = items[i];
        if (it.kind == BT_OPEN || it.kind == BT_CLOSE) {
          if ((it.flags & BTF_CAPTURING) && it.gid >= 1 && it.gid <= 9)
            (it.kind == BT_OPEN ? cg.gopen : cg.gclose)[it.gid] = cg.nleaf;
          continue;
        }
        if (it.kind != BT_LEAF) { why = "anchor in the chain"; break; }
        if (cg.nleaf >= kChainLeaves) { why = "more than 16 leaves"; break; }
        if (it.min < 1 || (it.max != -1 && it.max < it.min) || it.max > 4000 || it.min > 4000) { why = "leaf that may take no byte"; break; }
        if (tbls[it.tbl * 3] != tbls[it.tbl * 3 + 1] || tbls[it.tbl * 3] != tbls[it.tbl * 3 + 2]) { why = "a leaf's three membership tests differ"; break; }
        cg.lmin[cg.nleaf] = it.min; cg.lmax[cg.nleaf] = it.max;
        ltbl.push_back(it.tbl * 3);
        ++cg.nleaf;
      }
      auto in = [&](int leaf, int c) { return (tbls[ltbl[leaf]][c >> 3] >> (c & 7)) & 1; };
      if (why.empty() && cg.nleaf == 0) why = "no leaf";
      for (int l = 0; l + 1 < cg.nleaf && why.empty(); ++l)
        if (cg.lmin[l] != cg.lmax[l])
          for (int c = 0; c < 256; ++c)
            if (in(l, c) && in(l + 1, c)) { why = "a leaf with a variable count shares a byte with the leaf behind it"; break; }
      for (int g = 1; g <= 9 && why.empty(); ++g)
        if ((cg.gopen[g] < 0) != (cg.gclose[g] < 0)) why = "group without both ends";
      if (why.empty()) {
        // states of the chain: base[l] + c, c = bytes leaf l has taken (capped at min for an unbounded leaf); -1 dead
        std::vector<int> base(cg.nleaf + 1, 0);
        for (int l = 0; l < cg.nleaf; ++l) base[l + 1] = base[l] + (cg.lmax[l] == -1 ? cg.lmin[l] : cg.lmax[l]) + 1;
        if (base[cg.nleaf] > 20000) why = "counted leaves: more than 20000 chain states";
        auto step = [&](int st, int c) {
          int l = 0;
          while (base[l + 1] <= st) ++l;
          const int k = st - base[l];
          const bool stay = cg.lmax[l] == -1 || k < cg.lmax[l];
          if (in(l, c) && stay) return base[l] + (cg.lmax[l] == -1 ? std::min(k + 1, cg.lmin[l]) : k + 1);
          if (k >= cg.lmin[l] && l + 1 < cg.nleaf && in(l + 1, c)) return base[l + 1] + 1;
          return -1;
        };
        auto accepts = [&](int st) { return st >= base[cg.nleaf - 1] + cg.lmin[cg.nleaf - 1]; };
        std::set<std::pair<int, int>> seen;
        std::vector<std::pair<int, int>> todo{{0, 0}};
        seen.insert({0, 0});
        while (!todo.empty() && why.empty()) {
          const auto [a, q] = todo.back(); todo.pop_back();
          const bool acc_a = a >= 0 && accepts(a), acc_q = q >= 0 && acc[q];
          if (acc_a != acc_q) { why = "the chain and the engine's table accept different texts"; break; }
          if (a < 0 && q < 0) continue;
          if (seen.size() > 200000) { why = "chain against table: too many pairs"; break; }
          for (int c = 0; c < 256; ++c) {
            const std::pair<int, int> nx{a < 0 ? -1 : step(a, c), q < 0 ? -1 : T[q][c]};
            if (nx.first < 0 && nx.second < 0) continue;
            if (seen.insert(nx).second) todo.push_back(nx);
          }
        }
      }
      if (why.empty()) {
        for (int c = 0; c < 256; ++c) {
          uint16_t m = 0;
          for (int l = 0; l < cg.nleaf; ++l) if (in(l, c)) m |= (uint16_t)(1u << l);
          cg.mask[c] = m;
        }
        cg.ok = true;
      }
      cg.why = why;
    }
    align(hp.blob, 8);
  }

  // ---- streaming automaton (findall only) -----------------------------------------
  d.off_stcol = -1;
  d.off_stcol32 = -1;
  d.st_acc32 = 0;
  d.off_mw_cls = d.off_mw_tab = -1;
  d.mw_ncfg = d.mw_cshift = d.mw_bytes = d.mw_k = 0;
  d.off_mwr_cls = -1;
  d.mwr_ncfg = d.mwr_cshift = d.mwr_bytes = d.mwr_k = 0;
  d.off_bk_cls = -1;
  d.bk_nsub = d.bk_cshift = d.bk_bytes = d.bk_start = 0;
  d.off_st_sync = -1;
  d.off_stg_pair = -1;
  d.st_nsync = 0;
  d.st_reset_byte = -1;
  hp.streamable_why_not.clear();
  d.st_kind = 0;
  d.st_fixed_len = 0;
  // Tables of a search automaton given as an entry matrix over its live states
  // (E[q][byte] = next << 2 | EMIT << 1 | NEWSTART) and their accept flags; picks the form.
  auto emit_stream_tables = [&](const std::vector<std::array<uint16_t, 256>>& E,
                                const std::vector<uint8_t>& live_acc) {
    const int nlive = (int)E.size();
    const uint32_t fl = PF_STREAMABLE | ((d.flags & PF_PREFILTER) ? 0u : (uint32_t)PF_STREAM_SEARCH);
    d.st_nstates = nlive;
    {
      std::array<uint8_t, 256> sync{};
      int nsync = 0;
      for (int c = 0; c < 256; ++c) {
        bool same = true, all_new = true;
        for (int q = 0; q < nlive; ++q) {
          same = same && (E[q][c] >> 2) == (E[0][c] >> 2);
          all_new = all_new && (E[q][c] & 1);
        }
        // same target for every state, and the start of the walk that is then alive is known: there
        // is none (idle), it begins at this byte from every state, or starts are not tracked (KMP)
        sync[c] = same && ((E[0][c] >> 2) == 0 || all_new || d.st_fixed_len > 0);
        nsync += sync[c];
      }
      d.st_reset_byte = -1;
      for (int c = 255; c >= 0 && d.st_reset_byte < 0; --c) {
        bool ok = true;
        for (int q = 0; q < nlive && ok; ++q)
          ok = (E[q][c] >> 2) == 0 && !(E[q][c] & 1) && (((E[q][c] >> 1) & 1) == (live_acc[q] ? 1 : 0));
        if (ok) d.st_reset_byte = c;
      }
      d.st_nsync = nsync;
      if (nsync) {
        d.off_st_sync = (int)hp.blob.size();
        put(hp.blob, sync.data(), 256);
      }
    }
    d.st_accept_mask = 0;
    for (int q = 0; q < nlive && q < 32; ++q)
      if (live_acc[q]) d.st_accept_mask |= 1u << q;
    // class-table form (and its two-bytes-per-lookup companion); set_kind = false: only as the pair
    // table's carrier for a plan whose own form is the wide byte columns
    auto emit_class_tables = [&](bool set_kind) {
      // class-table form: any number of live states that fits u16 entries and LDS
      std::array<uint8_t, 256> scls{};
      int sncls = 0;
      {
        std::map<std::vector<uint16_t>, int> seen;
        for (int c = 0; c < 256; ++c) {
          std::vector<uint16_t> col(nlive);
          for (int q = 0; q < nlive; ++q) col[q] = E[q][c];
          auto it = seen.find(col);
          if (it == seen.end()) it = seen.emplace(col, sncls++).first;
          scls[c] = (uint8_t)it->second;
        }
      }
      int cshift = 0;
      while ((1 << cshift) < sncls) ++cshift;
      const int ncp = 1 << cshift;
      if ((int64_t)nlive * ncp > 8192) {
        if (set_kind) hp.streamable_why_not = "search automaton too large for the streaming kernel's LDS table";
        return;
      }
      std::vector<uint16_t> tr((size_t)nlive * ncp, 0);
      for (int c = 0; c < 256; ++c)
        for (int q = 0; q < nlive; ++q) {
          const uint16_t e = E[q][c];
          tr[(size_t)q * ncp + scls[c]] = (uint16_t)((((e >> 2) << cshift) << 2) | (e & 3));
        }
      if (set_kind) { d.flags |= fl; d.st_kind = 2; }
      d.st_cshift = cshift;
      align(hp.blob, 16);
      const int begin = (int)hp.blob.size();
      d.off_stg_cls = begin;
      put(hp.blob, scls.data(), 256);
      d.off_stg_trans = (int)hp.blob.size();
      put(hp.blob, tr.data(), tr.size() * 2);
      d.off_stg_acc = (int)hp.blob.size();
      put(hp.blob, live_acc.data(), live_acc.size());
      // two bytes per lookup: the chain of dependent LDS reads is what bounds this form
      const int ncp2 = ncp * ncp;
      if (d.st_reset_byte >= 0 && (int64_t)nlive * ncp2 <= 4096) {
        // a representative byte of every class
        std::vector<int> rep(ncp, -1);
        for (int c = 0; c < 256; ++c)
          if (rep[scls[c]] < 0) rep[scls[c]] = c;
        std::vector<uint32_t> pair((size_t)nlive * ncp2, 0);
        for (int q = 0; q < nlive; ++q)
          for (int c0 = 0; c0 < ncp; ++c0)
            for (int c1 = 0; c1 < ncp; ++c1) {
              if (rep[c0] < 0 || rep[c1] < 0) continue;   // padding classes are never looked up
              const uint16_t e0 = E[q][rep[c0]];
              const uint16_t e1 = E[e0 >> 2][rep[c1]];
              pair[(size_t)q * ncp2 + ((size_t)c0 << cshift) + c1] =
                  ((uint32_t)((e1 >> 2) * ncp2) << 4) | ((uint32_t)(e1 & 3) << 2) | (uint32_t)(e0 & 3);
            }
        align(hp.blob, 4);
        d.off_stg_pair = (int)hp.blob.size();
        put(hp.blob, pair.data(), pair.size() * 4);
      }
      align(hp.blob, 16);
      d.stg_bytes = (int)hp.blob.size() - begin;
    };
    if (nlive <= 4) {
      // byte-column form: col[byte] = 4 entries x 4 bit
      std::vector<uint16_t> cols(256, 0);
      for (int c = 0; c < 256; ++c)
        for (int q = 0; q < nlive; ++q) cols[c] |= (uint16_t)((E[q][c] & 0xF) << (4 * q));
      d.flags |= fl;
      d.st_kind = 1;
      align(hp.blob, 4);
      d.off_stcol = (int)hp.blob.size();
      put(hp.blob, cols.data(), 512);
      // code columns (see DevPlan::off_stcol32): the events must be functions of the states' 2-bit codes
      if (d.st_reset_byte >= 0 && !live_acc[0]) {
        int first[4] = {0, 0, 0, 0};
        for (int q = 0; q < nlive; ++q)
          for (int c = 0; c < 256; ++c)
            if (E[q][c] & 1) first[E[q][c] >> 2] = 1;
        bool ok = first[0] == 0;
        for (int q = 0; q < nlive && ok; ++q)
          for (int c = 0; c < 256 && ok; ++c) {
            const int nx = E[q][c] >> 2, em = (E[q][c] >> 1) & 1, ns = E[q][c] & 1;
            ok = nx < nlive && em == ((live_acc[q] && !live_acc[nx]) ? 1 : 0) && ns == ((first[nx] && !first[q]) ? 1 : 0);
          }
        int off[4] = {0, 0, 0, 0}, cur = 0;
        for (int q = 0; q < nlive && ok; ++q) {
          const int code = (live_acc[q] ? 2 : 0) | first[q];
          int o = cur;
          while ((o & 3) != code) ++o;
          off[q] = o;
          cur = o + 5;
          // u16 entries (idle + two states): a u32 table was measured to add a fifth to the LDS bank conflicts
          // (bank = byte mod 32 instead of byte / 2 mod 32), which cost the count kernel 8 %
          ok = cur <= 16;
        }
        if (ok) {
          std::vector<uint16_t> c16(256, 0);
          for (int c = 0; c < 256; ++c)
            for (int q = 0; q < nlive; ++q) c16[c] |= (uint16_t)(off[E[q][c] >> 2] << off[q]);
          for (int q = 0; q < nlive; ++q)
            if (live_acc[q]) d.st_acc32 |= 1u << off[q];
          align(hp.blob, 4);
          d.off_stcol32 = (int)hp.blob.size();
          put(hp.blob, c16.data(), 512);
        }
      }
    } else if (nlive <= 8) {
      // wide byte-column form: 8 states x 8-bit fields in a u64 column,
      // field(q) = next << 3 | EMIT << 1 | NEWSTART, so "field & 0x38" is the next shift amount
      std::vector<uint64_t> cols64(256, 0);
      for (int c = 0; c < 256; ++c)
        for (int q = 0; q < nlive; ++q)
          cols64[c] |= (uint64_t)((((E[q][c] >> 2) << 3) | (E[q][c] & 3)) & 0xFF) << (8 * q);
      d.flags |= fl;
      d.st_kind = 3;
      align(hp.blob, 16);
      d.off_stcol = (int)hp.blob.size();
      put(hp.blob, cols64.data(), 2048);
      // with a reset byte and few classes the two-bytes-per-lookup table is faster still (5 TB/s
      // against 4 on configs 1 / 5); the columns stay for pieces-free plans without it
      if (d.st_reset_byte >= 0) emit_class_tables(false);
    } else {
      emit_class_tables(true);
    }
  };
  if (d.kind == PLAN_ANY) hp.streamable_why_not = "'.*' shortcut";
  else if (d.flags & PF_EXACT_LITERAL) {
    // HybridMatcher's exact-literal bypass (matcher.mojo:768-781, 815-847): search = first
    // occurrence, findall = EVERY occurrence, overlapping ones included (start = pos + 1).  That
    // is the KMP automaton of the literal: state = length of the matched prefix, the full state L
    // emits on the byte that follows it (or at the end of the text) and continues through its
    // failure link; a match is [end - L, end), so no start tracking is needed.
    const std::string& lit_ = exact_lit;
    const int Ln = (int)lit_.size();
    if (Ln == 0 || Ln > 4000) hp.streamable_why_not = "exact literal too long for the streaming tables";
    else {
      std::vector<std::array<int, 256>> delta(Ln + 1);
      std::vector<int> fail(Ln + 1, 0);
      for (int c = 0; c < 256; ++c) delta[0][c] = ((unsigned char)lit_[0] == c) ? 1 : 0;
      for (int q = 1; q <= Ln; ++q) {
        // fail[q] = state reached by the text lit[1..q) ; standard KMP automaton construction
        fail[q] = (q == 1) ? 0 : delta[fail[q - 1]][(unsigned char)lit_[q - 1]];
        for (int c = 0; c < 256; ++c)
          delta[q][c] = (q < Ln && (unsigned char)lit_[q] == c) ? q + 1 : delta[fail[q]][c];
      }
      std::vector<std::array<uint16_t, 256>> E(Ln + 1);
      std::vector<uint8_t> live_acc(Ln + 1, 0);
      live_acc[Ln] = 1;
      for (int q = 0; q <= Ln; ++q)
        for (int c = 0; c < 256; ++c) E[q][c] = (uint16_t)((delta[q][c] << 2) | (q == Ln ? 2 : 0));
      d.st_fixed_len = Ln;
      emit_stream_tables(E, live_acc);
      if (!(d.flags & PF_STREAMABLE)) d.st_fixed_len = 0;
    }
  }
  else if (d.flags & PF_START_DEAD) hp.streamable_why_not = "dead start state";
  else if (d.flags & PF_BT_SEARCH) hp.streamable_why_not = "backtracking matcher route (NFAEngine.match_all)";
  else if (d.flags & PF_BITSET) hp.streamable_why_not = "bitset NFA walk (no determinised table)";
  else if (d.flags & (PF_START_ANCHOR | PF_END_ANCHOR)) hp.streamable_why_not = "anchored";
  else if (d.required_byte >= 0) {
    hp.streamable_why_not = "required-byte findall path";
    // one-pass forms on k_mwalk: the plain search (match_next never takes the required-byte route) and findall's
    // required-byte route itself
    if (hp.why_no_search.empty() && d.kind == PLAN_DFA && (d.flags & PF_HAS_MATCHER) && !first[d.required_byte] &&
        !(d.flags & (PF_PURE_LITERAL | PF_EXACT_LITERAL | PF_SCAN_ELIGIBLE | PF_START_ACCEPTING))) {
      SearchAutomaton sa;
      sa.n = d.nstates;
      sa.next = T;
      sa.acc = acc;
      sa.allowed = first;
      auto store = [&](const MultiWalk& mw, int32_t& off_cls, int32_t& ncfg, int32_t& cshift, int32_t& bytes, int32_t& kk) {
        align(hp.blob, 16);
        off_cls = (int)hp.blob.size();
        put(hp.blob, mw.cls.data(), 256);
        put(hp.blob, mw.tab.data(), mw.tab.size() * 4);
        ncfg = mw.ncfg; cshift = mw.cshift; bytes = 256 + (int)mw.tab.size() * 4; kk = mw.kmax;
        align(hp.blob, 16);
      };
      MultiWalk plain, req;
      std::string w1, w2;
      {   // marks for the plain route (search never takes the required-byte route, nor does sub's match_next loop)
        BackSet bk;
        std::string bwhy;
        if (build_backset(sa, bk, bwhy)) {
          align(hp.blob, 16);
          d.off_bk_cls = (int)hp.blob.size();
          put(hp.blob, bk.cls.data(), 256);
          put(hp.blob, bk.tab.data(), bk.tab.size() * 2);
          d.bk_nsub = bk.nsub; d.bk_cshift = bk.cshift; d.bk_start = bk.start;
          d.bk_bytes = 256 + (int)bk.tab.size() * 2;
          d.flags |= PF_BACKSET;
          align(hp.blob, 16);
        } else hp.backset_why_not = bwhy;
      }
      if (build_multiwalk(sa, plain, w1)) {
        store(plain, d.off_mw_cls, d.mw_ncfg, d.mw_cshift, d.mw_bytes, d.mw_k);
        d.off_mw_tab = d.off_mw_cls + 256;
        d.flags |= PF_MWALK;
      } else hp.mwalk_why_not = w1;
      if (build_reqwalk(sa, d.required_byte, first, req, w2)) {
        store(req, d.off_mwr_cls, d.mwr_ncfg, d.mwr_cshift, d.mwr_bytes, d.mwr_k);
        d.flags |= PF_MWALK_REQ;
      } else hp.mwalk_req_why_not = w2;
    }
  }
  else if (!hp.why_no_search.empty()) hp.streamable_why_not = hp.why_no_search;
  else {
    SearchAutomaton sa;
    if (d.flags & PF_SCAN_ELIGIBLE) {
      // dfa.mojo:2075-2094: maximal runs of first-class bytes, whatever the table says
      sa.n = 2;
      sa.next.assign(2, {});
      for (auto& r : sa.next) r.fill(-1);
      sa.acc = {0, 1};
      for (int c = 0; c < 256; ++c)
        if (first[c]) { sa.next[0][c] = 1; sa.next[1][c] = 1; }
      sa.allowed = first;
    } else {
      // DFAEngine table walk (also the chain of a pure literal: simd_search == leftmost
      // start whose walk reaches the end) or LazyDFA walk; candidate starts are limited
      // by the first-class / first-byte filter when the engine has one
      sa.n = d.nstates;
      sa.next = T;
      sa.acc = acc;
      if (d.flags & PF_HAS_MATCHER) sa.allowed = first;
      else sa.allowed.fill(1);
    }
    std::string why;
    if (!check_streamable(sa, why) && (d.flags & PF_PURE_LITERAL) && !lit.empty() && lit.size() <= 4000) {
      // A literal whose prefix is also a suffix ("555-": a later start survives the byte that kills
      // the walk).  DFAEngine's pure-literal loops are a substring search that resumes at the END of
      // each occurrence (dfa.mojo:2053-2073, simd_search), i.e. the KMP automaton of the literal with
      // the full state falling back to the start state; a match is [end - L, end).
      const int Ln = (int)lit.size();
      std::vector<std::array<int, 256>> delta(Ln + 1);
      std::vector<int> fail(Ln + 1, 0);
      for (int c = 0; c < 256; ++c) delta[0][c] = ((unsigned char)lit[0] == c) ? 1 : 0;
      for (int q = 1; q <= Ln; ++q) {
        fail[q] = (q == 1) ? 0 : delta[fail[q - 1]][(unsigned char)lit[q - 1]];
        for (int c = 0; c < 256; ++c)
          delta[q][c] = (q < Ln && (unsigned char)lit[q] == c) ? q + 1 : q == Ln ? delta[0][c] : delta[fail[q]][c];
      }
      std::vector<std::array<uint16_t, 256>> E(Ln + 1);
      std::vector<uint8_t> live_acc(Ln + 1, 0);
      live_acc[Ln] = 1;
      for (int q = 0; q <= Ln; ++q)
        for (int c = 0; c < 256; ++c) E[q][c] = (uint16_t)((delta[q][c] << 2) | (q == Ln ? 2 : 0));
      d.st_fixed_len = Ln;
      emit_stream_tables(E, live_acc);
      if (!(d.flags & PF_STREAMABLE)) d.st_fixed_len = 0;
    } else if (!why.empty()) {
      hp.streamable_why_not = why;
      // several walks side by side (k_mwalk) where one walk does not do; plain route only (no required byte)
      if (d.required_byte < 0 && !(d.flags & (PF_PURE_LITERAL | PF_EXACT_LITERAL | PF_SCAN_ELIGIBLE))) {
        {   // marks of the positions where a match begins, for the stepper (whatever the multi-walk builder says)
          BackSet bk;
          std::string bwhy;
          if (build_backset(sa, bk, bwhy)) {
            align(hp.blob, 16);
            d.off_bk_cls = (int)hp.blob.size();
            put(hp.blob, bk.cls.data(), 256);
            put(hp.blob, bk.tab.data(), bk.tab.size() * 2);
            d.bk_nsub = bk.nsub; d.bk_cshift = bk.cshift; d.bk_start = bk.start;
            d.bk_bytes = 256 + (int)bk.tab.size() * 2;
            d.flags |= PF_BACKSET;
            align(hp.blob, 16);
          } else {
            hp.backset_why_not = bwhy;
          }
        }
        MultiWalk mw;
        std::string mwhy;
        if (build_multiwalk(sa, mw, mwhy)) {
          align(hp.blob, 16);
          d.off_mw_cls = (int)hp.blob.size();
          put(hp.blob, mw.cls.data(), 256);
          d.off_mw_tab = (int)hp.blob.size();
          put(hp.blob, mw.tab.data(), mw.tab.size() * 4);
          d.mw_ncfg = mw.ncfg;
          d.mw_cshift = mw.cshift;
          d.mw_k = mw.kmax;
          d.mw_bytes = 256 + (int)mw.tab.size() * 4;
          d.flags |= PF_MWALK;
          align(hp.blob, 16);
        } else {
          hp.mwalk_why_not = mwhy;
          // ... then the tries that may be asked for beside the oldest walk (build_emptywalk2, empty = false): host copy
          if (!sa.acc[0]) {
            hp.ew2_ok = build_emptywalk2(sa, hp.ew2, hp.ew2_why, false);
            hp.ew2_tries = true;
            if (hp.ew2_ok && hp.blob.size() + 8192 + (hp.ew2.tab.size() + hp.ew2.end.size()) * sizeof(EwEntry) > 60 * 1024) {
              hp.ew2_ok = false;   // (8 KiB kept for what is appended behind this point: stepper table, sync bytes)
              hp.ew2_why = "pending-tries walk: no room left in the plan's 60 KiB";
            }
            if (hp.ew2_ok) {
              align(hp.blob, 16);
              d.off_mw_cls = (int)hp.blob.size();
              put(hp.blob, hp.ew2.cls.data(), 256);
              d.off_mw_tab = (int)hp.blob.size();
              put(hp.blob, hp.ew2.tab.data(), hp.ew2.tab.size() * sizeof(EwEntry));
              put(hp.blob, hp.ew2.end.data(), hp.ew2.end.size() * sizeof(EwEntry));
              d.mw_ncfg = hp.ew2.ncfg;
              d.mw_cshift = hp.ew2.cshift;
              d.mw_k = -3;   // (-3: k_mwalk<., 2, 0, 3>)
              d.mw_bytes = 256 + (int)((hp.ew2.tab.size() + hp.ew2.end.size()) * sizeof(EwEntry));
              d.flags |= PF_MW_TRIES;
              align(hp.blob, 16);
            }
          }
        }
      }
    } else {
      // number the states actually reachable
      std::vector<int> remap(sa.n, -1);
      int nlive = 0;
      remap[0] = nlive++;
      std::vector<int> st{0};
      while (!st.empty()) {
        const int q = st.back(); st.pop_back();
        for (int c = 0; c < 256; ++c) {
          const int t = (q == 0) ? (sa.allowed[c] ? sa.next[0][c] : -1) : sa.next[q][c];
          if (t > 0 && remap[t] < 0) { remap[t] = nlive++; st.push_back(t); }
        }
      }
      std::vector<uint8_t> live_acc(nlive, 0);
      for (int q = 0; q < sa.n; ++q)
        if (remap[q] >= 0) live_acc[remap[q]] = sa.acc[q];
      emit_stream_tables(stream_entries(sa, remap, nlive), live_acc);
    }
  }
  // ---- flattened generic kernel (k_step_*) -------------------------------------------------
  // The plain route of DFAEngine.match_all / match_next (dfa.mojo:2096-2130, 2200-2253) and of
  // LazyDFA (pikevm.mojo:754-817): optional skip to a first-class / first-byte candidate, table
  // walk, emit (start, last accept) and continue at the match end, or retry at start + 1.
  // Everything that takes another branch upstream keeps the literal restatement in mrx_device.hpp.
  if (lazy_end && (d.flags & PF_LAZY_END) && d.nstates <= 94 &&
      !(d.flags & (PF_EXACT_LITERAL | PF_PREFILTER | PF_START_ACCEPTING | PF_START_DEAD | PF_BITSET | PF_BT_SEARCH))) {
    // '$' on the LazyDFA search: the windowed stepper's plain route in its LZ form (both variants of every state's
    // row and of the idle row in one byte-indexed table); one lane per text only -- no pieces, no wavefront kernel
    d.flags |= PF_STEP_SEARCH | PF_STEPPABLE;
  } else
  if ((d.kind == PLAN_DFA || d.kind == PLAN_LAZY) && hp.why_no_search.empty() &&
      !(d.flags & (PF_START_ANCHOR | PF_END_ANCHOR | PF_PURE_LITERAL | PF_EXACT_LITERAL | PF_START_ACCEPTING |
                   PF_START_DEAD | PF_BITSET | PF_SCAN_ELIGIBLE | PF_BT_SEARCH)) &&
      (d.nstates <= 96 ||   // (nstates + 1) x 512 B byte-indexed table in LDS
       (d.required_byte < 0 && (int64_t)d.nstates * d.ncls < 16384))) {   // or nstates x ncls entries (k_req_wave BIG)
    if (d.nstates > 96) d.flags |= PF_STEP_BIG;
    d.flags |= PF_STEP_SEARCH;                  // match_next never takes the required-byte route
    if (d.required_byte < 0) d.flags |= PF_STEPPABLE;
    // findall with a rare required byte (_match_all_required_byte, matcher.mojo:864-898): memchr for
    // the byte, back up over first-class bytes, anchored walk from there, keep it if it passes the hit
    else if (d.kind == PLAN_DFA && (d.flags & PF_HAS_MATCHER) && !first[d.required_byte]) d.flags |= PF_STEP_REQ;
    // Bytes at which a long text may be cut for these routes (findall / count of a plan that does not stream):
    // no state has a transition on the byte -- every walk dies there and none begins -- and the required-byte
    // route neither stops at it nor backs up over it.  No match contains such a byte, so the pieces between
    // them are searched as texts of their own (k_virt_* with disjoint pieces, see findall of stepper plans).
    if (!(d.flags & (PF_STREAMABLE | PF_STEP_BIG)) && (d.flags & (PF_STEPPABLE | PF_STEP_REQ)) && d.st_nsync == 0) {
      std::array<uint8_t, 256> sync{};
      int nsync = 0;
      for (int c = 0; c < 256; ++c) {
        bool dead = true;
        for (size_t q = 0; q < T.size() && dead; ++q) dead = T[q][c] < 0;
        sync[c] = dead && !first[c] && c != d.required_byte;
        nsync += sync[c];
      }
      if (nsync) {
        d.st_nsync = nsync;
        d.off_st_sync = (int)hp.blob.size();
        put(hp.blob, sync.data(), 256);
        align(hp.blob, 16);
      }
    }
  }

  // Empty matches (dfa.mojo:2118-2130 / pikevm.mojo:805-817: "while pos <= len: try at pos; every try
  // matches; resume at the end, or one byte on after an empty match"): the start state accepts and no
  // first-byte matcher or filter picks the candidates.  findall / count only -- match_next is one try at 0.
  if ((d.kind == PLAN_DFA || d.kind == PLAN_LAZY) && hp.why_no_search.empty() && (d.flags & PF_START_ACCEPTING) &&
      !(d.flags & (PF_START_ANCHOR | PF_END_ANCHOR | PF_PURE_LITERAL | PF_EXACT_LITERAL | PF_PREFILTER | PF_HAS_MATCHER |
                   PF_START_DEAD | PF_BITSET | PF_SCAN_ELIGIBLE | PF_BT_SEARCH)) &&
      d.required_byte < 0 && d.nstates <= 96) {
    d.flags |= PF_STEP_EMPTY;
    // census for the one-pass form: does a walk ever read beyond its last accepting position?  (Never, when every
    // state a walk can reach accepts: it dies ON the byte behind its match.)
    std::vector<uint8_t> seen(d.nstates, 0);
    std::vector<int> st{0};
    seen[0] = 1;
    bool all_acc = true;
    while (!st.empty()) {
      const int q = st.back(); st.pop_back();
      if (!acc[q]) all_acc = false;
      for (int c = 0; c < 256; ++c) {
        const int t = T[q][c];
        if (t >= 0 && !seen[t]) { seen[t] = 1; st.push_back(t); }
      }
    }
    hp.empty_all_accepting = all_acc;
    if (!all_acc) {   // walks that read beyond their match: the general table (host copy for the tests' table run as well)
      SearchAutomaton sa;
      sa.n = d.nstates;
      sa.next = T;
      sa.acc = acc;
      sa.allowed.fill(1);
      hp.ew2_ok = build_emptywalk2(sa, hp.ew2, hp.ew2_why);
      if (hp.ew2_ok && hp.blob.size() + 512 + (hp.ew2.tab.size() + hp.ew2.end.size()) * sizeof(EwEntry) > 60 * 1024) {
        hp.ew2_ok = false;
        hp.ew2_why = "pending-tries walk: no room left in the plan's 60 KiB";
      }
      if (hp.ew2_ok) {
        align(hp.blob, 16);
        d.off_mw_cls = (int)hp.blob.size();
        put(hp.blob, hp.ew2.cls.data(), 256);
        d.off_mw_tab = (int)hp.blob.size();
        put(hp.blob, hp.ew2.tab.data(), hp.ew2.tab.size() * sizeof(EwEntry));
        put(hp.blob, hp.ew2.end.data(), hp.ew2.end.size() * sizeof(EwEntry));
        d.mw_ncfg = hp.ew2.ncfg;
        d.mw_cshift = hp.ew2.cshift;
        d.mw_k = -2;   // (-2: k_mwalk<., 2, 0, 2>, 128-bit entries)
        d.mw_bytes = 256 + (int)((hp.ew2.tab.size() + hp.ew2.end.size()) * sizeof(EwEntry));
        d.flags |= PF_MW_EMPTY;
        align(hp.blob, 16);
      }
    }
    if (all_acc) {
      SearchAutomaton sa;
      sa.n = d.nstates;
      sa.next = T;
      sa.acc = acc;
      sa.allowed.fill(1);
      MultiWalk mw;
      std::string why;
      if (build_emptywalk(sa, mw, why)) {
        align(hp.blob, 16);
        d.off_mw_cls = (int)hp.blob.size();
        put(hp.blob, mw.cls.data(), 256);
        d.off_mw_tab = (int)hp.blob.size();
        put(hp.blob, mw.tab.data(), mw.tab.size() * 4);
        d.mw_ncfg = mw.ncfg;
        d.mw_cshift = mw.cshift;
        d.mw_k = -1;   // (-1 = the empty-match walk: k_mwalk<., 2, 0, EMP>)
        d.mw_bytes = 256 + (int)mw.tab.size() * 4;
        d.flags |= PF_MW_EMPTY;
        align(hp.blob, 16);
      } else {
        hp.mwalk_why_not = why;
      }
    }
  }

  // The same plain route for a LazyDFA that is walked as a bitset NFA (pikevm.mojo:754-817 over the state
  // sets of pikevm.mojo:497-648): one 64-bit word of live positions per lane, no determinised table.
  if (d.kind == PLAN_LAZY && (d.flags & PF_BITSET) && hp.bitset.nw == 1 && hp.why_no_search.empty() &&
      !(d.flags & (PF_EXACT_LITERAL | PF_START_ACCEPTING | PF_START_DEAD | PF_BT_SEARCH)))
    d.flags |= PF_BSTEP | PF_STEP_SEARCH | PF_STEPPABLE;

  // ---- anchored automaton: regex.match_first as one forward pass ----------------------
  // match_first(text) = engine_match_first(text, 0) keeps no restart loop, so it is a plain
  // automaton run from byte 0 that remembers the last accepting position:
  //   pure literal       verify_match at 0 (simd_ops.mojo:937-960): the literal's chain
  //   _try_match_simd    (dfa.mojo:2133-2197) first-class run; falls through to the table walk
  //                      when the run is empty and the start state does not accept
  //   otherwise          the table walk (dfa.mojo:1979-2024 / pikevm.mojo:819-867)
  d.fa_bytes = 0; d.fa_nstates = 0; d.fa_start_acc = 0; d.off_fa_cls = d.off_fa_trans = -1; d.fa_cshift = 0;
  d.off_fa_end = -1;
  d.fa_kind = 0; d.off_fa_col = -1; d.off_fa_run = -1;
  hp.first_stream_why_not.clear();
  if (d.kind == PLAN_ANY) hp.first_stream_why_not = "'.*' shortcut";
  else if (!hp.why_no_match_first.empty()) hp.first_stream_why_not = hp.why_no_match_first;
  else if (d.flags & PF_BT_FIRST) hp.first_stream_why_not = "backtracking matcher route (NFAEngine.match_first)";
  else if (!hp.first_onepass && (d.flags & PF_START_DEAD)) hp.first_stream_why_not = "dead start state";
  else if (!hp.first_onepass && (d.flags & PF_BITSET)) hp.first_stream_why_not = "bitset NFA walk (no determinised table)";
  else if ((d.flags & PF_END_ANCHOR) &&
           (d.kind != PLAN_DFA || (d.flags & PF_PURE_LITERAL) ||
            ((d.flags & PF_HAS_MATCHER) && (d.flags & (PF_SCAN_ELIGIBLE | PF_START_ACCEPTING)))))
    // (with the _try_match_simd shortcut in play a '$' match is "the class run reaches the end, or else the
    // table walk does": two walks; pure literals return before the '$' check, dfa.mojo:1915-1925)
    hp.first_stream_why_not = "'$' needs the end-of-text check of both paths";
  else {
    std::vector<std::array<int, 256>> N;  // -1 = dead
    std::vector<uint8_t> A, E;            // E: OnePass end-of-text flags (empty otherwise)
    auto add = [&](bool a) { std::array<int, 256> r; r.fill(-1); N.push_back(r); A.push_back(a ? 1 : 0); return (int)N.size() - 1; };
    if (hp.first_onepass) {
      // OnePassNFA.match_first from 0 (onepass.mojo:440-488); '^' is vacuous at start == 0
      const OnePassTables& op = hp.onepass;
      for (size_t q = 0; q < op.trans.size(); ++q) {
        const int id = add(op.is_match[q] != 0);
        for (int c = 0; c < 256; ++c) N[id][c] = op.trans[q][c];
        E.push_back(op.is_end_match[q]);
      }
    } else if (d.flags & PF_PURE_LITERAL) {
      const std::string& L = hp.dfa.literal;
      add(L.empty());
      for (size_t k = 0; k < L.size(); ++k) { const int t = add(k + 1 == L.size()); N[t - 1][(unsigned char)L[k]] = t; }
    } else {
      const bool quirk = d.kind == PLAN_DFA && (d.flags & PF_HAS_MATCHER) &&
                         (d.flags & (PF_SCAN_ELIGIBLE | PF_START_ACCEPTING));
      const int base = quirk ? 2 : 0;   // quirk: 0 = start, 1 = inside the first-class run
      if (quirk) { add((d.flags & PF_START_ACCEPTING) != 0); add(true); }
      // '$' (dfa.mojo:2019-2024: the LAST accepting position of the greedy walk must be the end of the text,
      // i.e. the walk reaches the end alive and in an accepting state): no state accepts on the way, the
      // accepting flags become the end-of-text flags the OnePass tables use
      const bool dollar = (d.flags & PF_END_ANCHOR) != 0;
      for (size_t q = 0; q < T.size(); ++q) {
        const int id = add(!dollar && acc[q] != 0);
        for (int c = 0; c < 256; ++c) N[id][c] = T[q][c] < 0 ? -1 : T[q][c] + base;
        if (dollar) E.push_back(acc[q] != 0);
      }
      if (quirk)
        for (int c = 0; c < 256; ++c) {
          if (first[c]) { N[0][c] = 1; N[1][c] = 1; }
          else if (!(d.flags & PF_START_ACCEPTING)) N[0][c] = N[base][c];  // empty run: table walk from 0
        }
    }
    // reachable states only, then the dead state
    std::vector<int> remap(N.size(), -1), order;
    remap[0] = 0; order.push_back(0);
    for (size_t k = 0; k < order.size(); ++k)
      for (int c = 0; c < 256; ++c) {
        const int t = N[order[k]][c];
        if (t >= 0 && remap[t] < 0) { remap[t] = (int)order.size(); order.push_back(t); }
      }
    const int nl = (int)order.size();      // live states; dead = nl
    std::array<uint8_t, 256> fcls{};
    int fn = 0;
    {
      std::map<std::vector<int>, int> seen;
      for (int c = 0; c < 256; ++c) {
        std::vector<int> col(nl);
        for (int q = 0; q < nl; ++q) { const int t = N[order[q]][c]; col[q] = t < 0 ? -1 : remap[t]; }
        auto it = seen.find(col);
        if (it == seen.end()) it = seen.emplace(col, fn++).first;
        fcls[c] = (uint8_t)it->second;
      }
    }
    int cshift = 0;
    while ((1 << cshift) < fn) ++cshift;
    const int ncp = 1 << cshift;
    if (E.empty() && nl + 1 <= 8) {
      // byte-column forms (no dependent LDS read per byte): field(q) = next | ACC(next) << 1 in the
      // layouts of the search automaton, the dead state is field nl
      const bool narrow = nl + 1 <= 4;
      std::vector<uint64_t> cols(256, 0);
      for (int c = 0; c < 256; ++c) {
        for (int q = 0; q <= nl; ++q) {
          int t = nl;   // dead
          bool a = false;
          if (q < nl && N[order[q]][c] >= 0) { t = remap[N[order[q]][c]]; a = A[N[order[q]][c]] != 0; }
          const uint64_t f = narrow ? (uint64_t)((t << 2) | (a ? 2 : 0)) : (uint64_t)((t << 3) | (a ? 2 : 0));
          cols[c] |= f << ((narrow ? 4 : 8) * q);
        }
      }
      align(hp.blob, 16);
      const int begin = (int)hp.blob.size();
      d.off_fa_col = begin;
      if (narrow) {
        std::vector<uint16_t> c16(256);
        for (int c = 0; c < 256; ++c) c16[c] = (uint16_t)cols[c];
        put(hp.blob, c16.data(), 512);
      } else {
        put(hp.blob, cols.data(), 2048);
      }
      align(hp.blob, 16);
      d.fa_bytes = (int)hp.blob.size() - begin;
      d.fa_kind = narrow ? 1 : 3;
      d.fa_nstates = nl; d.fa_start_acc = A[0];
      if (nl == 2 && !A[order[0]] && A[order[1]]) {
        // start -C-> s, s -C-> s over one byte class C, nothing else: match_first = the run of C at 0
        bool run = true;
        std::array<uint8_t, 256> in{};
        for (int c = 0; c < 256 && run; ++c) {
          const int t0 = N[order[0]][c], t1 = N[order[1]][c];
          in[c] = t0 >= 0;
          run = (t0 >= 0) == (t1 >= 0) && (t0 < 0 || (remap[t0] == 1 && remap[t1] == 1));
        }
        if (run) {
          d.off_fa_run = (int)hp.blob.size();
          put(hp.blob, in.data(), 256);
          align(hp.blob, 16);
        }
      }
    } else if ((int64_t)(nl + 1) * ncp > 8192) {
      hp.first_stream_why_not = "anchored automaton too large for the streaming kernel's LDS table";
    } else {
      d.fa_kind = 2;
      std::vector<uint16_t> tr((size_t)(nl + 1) * ncp, (uint16_t)(((nl << cshift) << 2)));  // default: dead
      for (int c = 0; c < 256; ++c)
        for (int q = 0; q < nl; ++q) {
          const int t = N[order[q]][c];
          if (t >= 0) tr[(size_t)q * ncp + fcls[c]] = (uint16_t)((((remap[t]) << cshift) << 2) | (A[t] ? 2 : 0));
        }
      align(hp.blob, 16);
      const int begin = (int)hp.blob.size();
      d.off_fa_cls = begin; put(hp.blob, fcls.data(), 256);
      d.off_fa_trans = (int)hp.blob.size(); put(hp.blob, tr.data(), tr.size() * 2);
      if (!E.empty()) {
        std::vector<uint8_t> ends(nl + 1, 0);
        for (int q = 0; q < nl; ++q) ends[q] = E[order[q]];
        d.off_fa_end = (int)hp.blob.size(); put(hp.blob, ends.data(), ends.size());
      }
      align(hp.blob, 16);
      d.fa_bytes = (int)hp.blob.size() - begin;
      d.fa_cshift = cshift; d.fa_nstates = nl; d.fa_start_acc = A[0];
    }
  }
  if (hp.first_onepass && d.fa_bytes == 0)  // no generic kernel walks the OnePass tables
    hp.why_no_match_first = "OnePass table too large for the streaming kernel's LDS table";
  align(hp.blob, 16);
  d.blob_bytes = (int)hp.blob.size();
  // '$' on the LazyDFA search: none of the derived tables above exists for it (they were refused with the search);
  // the generic lane-per-text kernels serve it
  if (lazy_end && (d.flags & PF_LAZY_END)) {
    hp.why_no_search.clear();
    if (hp.streamable_why_not.empty() || hp.streamable_why_not[0] == '(')
      hp.streamable_why_not = "'$' on the LazyDFA search: per-text transition cache (generic kernels)";
  }
}

std::string describe_plan(const HostPlan& hp) {
  std::ostringstream o;
  const DevPlan& d = hp.dev;
  o << "pattern=" << hp.pattern << "\n";
  o << "engine_type=" << hp.engine_type << "\n";
  static const char* cxn[] = {"SIMPLE", "MEDIUM", "COMPLEX"};
  o << "complexity=" << cxn[hp.complexity] << "\n";
  if (hp.force_nfa) o << "option.lazydfa_semantics=1\n";
  if (hp.force_bitset) o << "option.bitset_nfa=1\n";
  if (hp.nfa_engine) o << "option.nfa_engine=1\n";
  if (hp.dfa_engine) o << "option.dfa_engine=1\n";
  o << "use_dfa=" << hp.use_dfa << " wildcard_any=" << hp.wildcard_any
    << " use_pure_dfa=" << hp.use_pure_dfa << "\n";
  o << "exact_literal=" << hp.exact_literal << " literal_has_anchors=" << hp.literal_has_anchors
    << " prefilter=" << hp.has_prefilter << " required_byte=" << hp.required_byte << "\n";
  auto hex = [&](const std::string& s) {
    static const char* dg = "0123456789abcdef";
    std::string h;
    for (unsigned char c : s) { h.push_back(dg[c >> 4]); h.push_back(dg[c & 15]); }
    return h;
  };
  o << "best_literal=" << hex(hp.best_literal) << "\n";
  if (hp.use_dfa) {
    const DfaEngine& e = hp.dfa;
    o << "dfa.shape=" << e.shape << "\n";
    o << "dfa.nstates=" << e.nstates() << "\n";
    o << "dfa.flags start_anchor=" << e.has_start_anchor << " end_anchor=" << e.has_end_anchor
      << " pure_literal=" << e.is_pure_literal << " has_matcher=" << e.has_matcher
      << " scan_eligible=" << e.scan_eligible << "\n";
    o << "dfa.literal=" << hex(e.literal) << "\n";
    o << "dfa.accepting=";
    for (int s = 0; s < e.nstates(); ++s) o << (int)e.accepting[s];
    o << "\n";
    if (e.has_matcher) {
      o << "dfa.matcher.num_ranges=" << e.matcher.num_ranges << "\n";
      o << "dfa.matcher.lookup=";
      for (int c = 0; c < 256; ++c) o << (int)e.matcher.lookup[c];
      o << "\n";
    }
    for (int s = 0; s < e.nstates(); ++s) {
      o << "dfa.row" << s << "=";
      // run-length form: c0-c1:target
      int c = 0;
      bool firstr = true;
      while (c < 256) {
        const int t = e.trans[s][c];
        int c2 = c;
        while (c2 + 1 < 256 && e.trans[s][c2 + 1] == t) ++c2;
        if (t != -1) {
          if (!firstr) o << ",";
          o << c << "-" << c2 << ":" << t;
          firstr = false;
        }
        c = c2 + 1;
      }
      o << "\n";
    }
  } else if (!hp.wildcard_any) {
    o << "nfa.program_len=" << hp.program.insts.size() << "\n";
    o << "nfa.program=";
    for (const Inst& in : hp.program.insts) o << (int)in.op << ":" << in.a0 << ":" << in.a1 << ";";
    o << "\n";
    o << "nfa.has_filter=" << hp.lazy.has_filter << " lazy_states=" << hp.lazy.trans.size()
      << " start_dead=" << hp.lazy.start_dead << " too_large=" << hp.lazy.too_large << "\n";
    o << "nfa.literal_opt=" << hp.nfa_has_literal_opt << " starts_dotstar=" << hp.nfa_starts_dotstar
      << " ends_dotstar=" << hp.nfa_ends_dotstar << "\n";
  }
  o << "fixed_groups=" << hp.fixed_ngroups << " fixed_total=" << hp.fixed_total
    << " fixed_concat=" << hp.fixed_concat << "\n";
  o << "support.match_first=" << (hp.why_no_match_first.empty() ? "yes" : hp.why_no_match_first) << "\n";
  o << "support.search=" << (hp.why_no_search.empty() ? "yes" : hp.why_no_search) << "\n";
  if (d.flags & PF_LAZY_END) o << "device.lazy_end_cache=yes states=" << d.nstates / 2 << "\n";
  o << "device.kind=" << d.kind << " nstates=" << d.nstates << " ncls=" << d.ncls
    << " flags=0x" << std::hex << d.flags << std::dec << " blob_bytes=" << d.blob_bytes << "\n";
  o << "device.streamable=" << ((d.flags & PF_STREAMABLE) ? "yes" : ("no: " + hp.streamable_why_not))
    << " st_nstates=" << d.st_nstates << " st_kind=" << d.st_kind
    << (((d.flags & PF_STREAMABLE) && !(d.flags & PF_STREAM_SEARCH)) ? " findall_only=1" : "")
    << " sync_bytes=" << d.st_nsync << " reset_byte=" << d.st_reset_byte << " code_columns=" << (d.off_stcol32 >= 0 ? 1 : 0)
    << " multiwalk=" << ((d.flags & PF_MWALK) ? "yes" : hp.mwalk_why_not.empty() ? "no" : "no: " + hp.mwalk_why_not)
    << " mw_configs=" << d.mw_ncfg << " mw_walks=" << d.mw_k
    << " multiwalk_req=" << ((d.flags & PF_MWALK_REQ) ? "yes" : hp.mwalk_req_why_not.empty() ? "no" : "no: " + hp.mwalk_req_why_not)
    << " mwr_configs=" << d.mwr_ncfg << " mwr_walks=" << d.mwr_k
    << " backset=" << ((d.flags & PF_BACKSET) ? "yes" : hp.backset_why_not.empty() ? "no" : "no: " + hp.backset_why_not)
    << " bk_sets=" << d.bk_nsub
    << (d.off_stg_pair >= 0 ? " pair_table=1" : "") << "\n";
  o << "device.steppable=" << ((d.flags & PF_STEPPABLE) ? "yes" : (d.flags & PF_STEP_REQ) ? "required-byte route" : "no")
    << " step_search=" << ((d.flags & PF_STEP_SEARCH) ? 1 : 0) << ((d.flags & PF_STEP_BIG) ? " big_table=1" : "")
    << ((d.flags & PF_BSTEP) ? " bitset=1" : "") << ((d.flags & PF_STEP_EMPTY) ? (hp.empty_all_accepting ? " empty_matches=1 every_state_accepts=1" : " empty_matches=1") : "")
    << ((d.flags & PF_MW_EMPTY) ? " empty_walk=1" : "")
    << (((d.flags & PF_STEP_EMPTY) && !hp.empty_all_accepting) ? (hp.ew2_ok ? " empty_walk2=yes configs=" + std::to_string(hp.ew2.ncfg) : " empty_walk2=no: " + hp.ew2_why) : std::string())
    << (hp.ew2_tries ? (hp.ew2_ok ? " tries_walk=yes configs=" + std::to_string(hp.ew2.ncfg) : " tries_walk=no: " + hp.ew2_why) : std::string()) << "\n";
  o << "device.first_stream=" << (d.fa_bytes ? "yes" : ("no: " + hp.first_stream_why_not))
    << " fa_nstates=" << d.fa_nstates << " fa_kind=" << d.fa_kind << (hp.first_onepass ? " onepass=yes" : "")
    << (d.off_fa_run >= 0 ? " class_run=1" : "") << "\n";
  o << "device.backtrack=" << (hp.bt.ok ? "yes" : ("no: " + (hp.bt.why_not.empty() ? std::string("'.*' shortcut") : hp.bt.why_not)));
  if (hp.bt.ok) o << " items=" << d.bt_nitems << " groups=" << d.bt_ngroups << " literal_opt=" << (d.bt_flags & 1)
                  << " prefix_literal=" << ((d.bt_flags >> 1) & 1) << " chain=" << ((d.bt_flags >> 5) & 1);
    o << " chain_groups=" << (hp.chain.ok ? "yes leaves=" + std::to_string(hp.chain.nleaf) : "no: " + hp.chain.why);
  o << "\n";
  if (hp.fixed_total >= 0)   // group templates of regex.sub: see HostPlan::fixed_pure
    o << "device.sub_groups=fixed pure=" << (hp.fixed_pure ? 1 : 0) << "\n";
  if (d.flags & PF_BITSET)
    o << "device.bitset=yes positions=" << d.bs_npos << " words=" << d.bs_nw << " byte_classes=" << d.bs_ncls << " fixed_len=" << d.bs_fixed_len << "\n";
  return o.str();
}

}  // namespace mrx
