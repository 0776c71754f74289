// HIP kernels (gfx950 / CDNA4) and the C ABI of libmrx_hip.so.
//
// Kernels
//   k_match<OP>        one lane per text: match_first / search / is_match / captures
//   k_findall<MODE>    one lane per text: count, or emit spans at a CSR offset
//   k_stream_findall   the streaming scan: texts at a fixed pitch, a wavefront
//                      stages 64 texts x CHUNK bytes through LDS with coalesced
//                      16-byte loads, each lane then walks its own text in lockstep
//                      through a register-indexed search automaton (no per-byte
//                      dependent memory access); match events go to per-text records
//   k_decode           event records -> CSR spans
//   k_sub<MODE>        output sizes / output bytes of sub()
//   k_scan_*           exclusive prefix sums (counts -> CSR offsets)
// All tables are staged from the plan blob into LDS once per workgroup.
#include <hip/hip_runtime.h>
#include <atomic>
#include <map>
#include <type_traits>
#include <unordered_map>
#include <chrono>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mrx.h"
#include "../../include/mrx_testing.h"
#include "mrx_device.hpp"
#include "mrx_internal.hpp"
#include "mrx_lookback.hpp"

using namespace mrx;

// ============================================================================
// device side
// ============================================================================
namespace {

constexpr int kBlock = 256;  // 4 wavefronts

struct Layout {  // where text i lives
  const uint8_t* data;
  const int64_t* offsets;  // CSR layout when non-null
  int64_t stride;          // fixed pitch otherwise
  const int32_t* lens;     // optional per-text length (fixed pitch)
  int32_t len;             // common length when lens == nullptr
  // Stepper kernels only (ragged batches with a few very long texts): k_wstep leaves texts of at least
  // `split` bytes to k_req_wave, which in turn skips the shorter ones; 0 = no split
  int32_t split = 0;
  // k_wstep / k_req_wave<STEP_SLOTS / STEP_EMIT> and k_slots_gather_wide: slot rows sized by the text
  // (len / 4 + 32 spans, see slot_row()) instead of kStepSlots, so that long texts rarely need the second walk
  int32_t wide_slots = 0;
  // Views (the `start` argument, mrx_*_at_*): with offsets != nullptr, text i is the vlen[i] bytes at
  // data + offsets[i] (offsets need not be contiguous); nullptr = plain CSR
  const int32_t* vlen = nullptr;
  const uint32_t* vskip = nullptr;   // with vlen: the per-text word k_stream_findall's VIRT form expects
  // k_mwalk<STEP_SLOTS / STEP_EMIT> only: text i is a piece of a longer text and its spans are reported as positions
  // of that text, vbase[i] + position in the piece (FindallJob::step_scan's pieces); nullptr = as found
  const int32_t* vbase = nullptr;
  // per text: first occurrence of the backtracking matcher's literal, last occurrence << 1 | has-newline
  // (k_litscan in front of the lane-per-text kernels, see bt_prepass()); nullptr = not computed
  const int2* pre = nullptr;
  // marks of the positions at which a match begins (k_backscan -> k_wstep<., 0, 0, 0, 1>): one bit per byte of the
  // text's FRAME (the 16-byte blocks that hold it: bit f = frame position f, the text begins at f = address & 15),
  // text i's words from bm_row(i) on -- a multiple of four, so that the four words of a 128-byte window are one
  // aligned 16-byte load; bm_cnt[i] = how many marks text i has
  const uint32_t* bm = nullptr;
  const int32_t* bm_cnt = nullptr;
  __host__ __device__ __forceinline__ int64_t bm_row(int64_t i) const {
    return 4 * (offsets ? (offsets[i] >> 7) + 2 * i : i * ((stride >> 7) + 2));
  }
  // first slot of text i's row and the row's capacity (wide rows)
  __device__ __forceinline__ int64_t slot_row(int64_t i, int* cap) const {
    if (offsets) {
      const int64_t a = (offsets[i] >> 2) + 32 * i, b = (offsets[i + 1] >> 2) + 32 * (i + 1);
      *cap = (int)(b - a);
      return a;
    }
    const int64_t row = (int64_t)((lens ? stride : (int64_t)len) >> 2) + 32;
    *cap = (int)row;
    return i * row;
  }
  __device__ __forceinline__ Text text(int64_t i) const {
    Text t = offsets ? Text(data + offsets[i], vlen ? vlen[i] : (int)(offsets[i + 1] - offsets[i]))
                     : Text(data + i * stride, lens ? lens[i] : len);
    if (pre) { const int2 v = pre[i]; t.pre_first = v.x; t.pre_last_nl = v.y; }
    return t;
  }
};

__device__ __forceinline__ Ctx stage_tables(const DevPlan& p, const uint8_t* __restrict__ blob,
                                            uint8_t* lds) {
  // cooperative 4-byte copy of the plan blob into LDS
  const uint32_t* src = (const uint32_t*)blob;
  uint32_t* dst = (uint32_t*)lds;
  const int words = p.blob_bytes >> 2;
  for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  Ctx c;
  c.p = p;
  c.cls = lds + p.off_cls;
  c.first = lds + p.off_first;
  c.trans = (const uint16_t*)(lds + p.off_trans);
  c.lit = lds + p.off_lit;
  c.pre = lds + p.off_pre;
  c.bs_cls = lds + p.off_bs_cls;
  c.bs_mask = (const uint64_t*)(lds + p.off_bs_mask);
  c.bs_follow = (const uint64_t*)(lds + p.off_bs_follow);
  c.bt_items = (const BtItem*)(lds + (p.off_bt_items >= 0 ? p.off_bt_items : 0));
  c.bt_tbl = lds + (p.off_bt_tbl >= 0 ? p.off_bt_tbl : 0);
  c.bt_lit = lds + (p.off_bt_lit >= 0 ? p.off_bt_lit : 0);
  return c;
}

enum { OP_MATCH_FIRST = 0, OP_SEARCH = 1, OP_IS_MATCH = 2, OP_CAPTURES = 3 };

// BT: the instantiation that carries the backtracking matcher (see engine_match_first in mrx_device.hpp)
template <int OP, int BT = 0>
__global__ __launch_bounds__(kBlock) void k_match(DevPlan p, const uint8_t* __restrict__ blob,
                                                  Layout lay, int64_t n, int32_t* __restrict__ out_s,
                                                  int32_t* __restrict__ out_e,
                                                  uint8_t* __restrict__ out_flag) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Ctx c = stage_tables(p, blob, lds);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const Text t = lay.text(i);
    int ms = -1, me = -1;
    if (OP == OP_MATCH_FIRST) {
      // regex.match_first, matcher.mojo:1396-1415: keep only matches starting at 0
      if (!(hybrid_match_first<BT>(c, t, 0, ms, me) && ms == 0)) ms = me = -1;
      out_s[i] = ms; out_e[i] = me;
    } else if (OP == OP_SEARCH) {
      if (!hybrid_match_next<BT>(c, t, 0, ms, me)) ms = me = -1;
      out_s[i] = ms; out_e[i] = me;
    } else if (OP == OP_IS_MATCH) {
      out_flag[i] = hybrid_is_match<BT>(c, t, 0) ? 1 : 0;
    } else {
      if constexpr (BT) if (p.fixed_total < 0) {
        // general groups: NFAEngine.match_next_with_groups (nfa.mojo:500-574) on the flat program
        const int g = p.bt_ngroups;
        int32_t* o = out_s + i * (int64_t)(g + 1) * 2;
        BtCaps caps;
        if (bt_match_next_with_groups<BT == 1>(c, t, 0, ms, me, caps)) {
          for (int k = 1; k <= g; ++k) { o[(k - 1) * 2] = caps.gs(k); o[(k - 1) * 2 + 1] = caps.ge(k); }
          o[g * 2] = ms; o[g * 2 + 1] = me;
        } else {
          for (int k = 0; k < (g + 1) * 2; ++k) o[k] = -1;
        }
        continue;
      }
      // search + fixed-width groups in NFAEngine._match_group order (nfa.mojo:1057-1103)
      const int g = p.fixed_ngroups;
      int32_t* o = out_s + i * (int64_t)(g + 1) * 2;
      if (hybrid_match_next<BT>(c, t, 0, ms, me)) {
        for (int k = 1; k <= g; ++k) {
          o[(k - 1) * 2] = ms + p.fixed_off[k];
          o[(k - 1) * 2 + 1] = ms + p.fixed_off[k] + p.fixed_w[k];
        }
        o[g * 2] = ms; o[g * 2 + 1] = me;
      } else {
        for (int k = 0; k < (g + 1) * 2; ++k) o[k] = -1;
      }
    }
  }
}

// captures from a search result: groups 1..g at their fixed offsets, then the whole match
// (NFAEngine._match_group order, nfa.mojo:1057-1103), -1 everywhere when the text has no match
__global__ void k_expand_captures(DevPlan p, int64_t n, const int32_t* __restrict__ start,
                                  const int32_t* __restrict__ end, int32_t* __restrict__ out) {
  const int g = p.fixed_ngroups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t* o = out + i * (int64_t)(g + 1) * 2;
    const int ms = start[i], me = end[i];
    if (ms >= 0) {
      for (int k = 1; k <= g; ++k) {
        o[(k - 1) * 2] = ms + p.fixed_off[k];
        o[(k - 1) * 2 + 1] = ms + p.fixed_off[k] + p.fixed_w[k];
      }
      o[g * 2] = ms; o[g * 2 + 1] = me;
    } else {
      for (int k = 0; k < (g + 1) * 2; ++k) o[k] = -1;
    }
  }
}

// DFAEngine.is_match with a first-byte matcher (dfa.mojo:1832-1843): "is the first byte in the first
// element's class, or does the start state accept" -- one byte per text, nothing is walked
__global__ __launch_bounds__(kBlock) void k_is_match_byte(Layout lay, int64_t n, const uint8_t* __restrict__ first,
                                                          int start_accepts, uint8_t* __restrict__ flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const Text t = lay.text(i);
    flag[i] = (uint8_t)((t.len > 0 && first[t.ptr[0]]) || start_accepts);
  }
}
// '^'-anchored DFA plans: match_all is one match_next, which only tries position 0 (dfa.mojo:2046-2050,
// 1887-1891) -- so findall / count are the anchored automaton's run: one span or none per text
__global__ void k_first_to_counts(int64_t n, const int32_t* __restrict__ start, int32_t* __restrict__ counts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    counts[i] = start[i] >= 0 ? 1 : 0;
}
__global__ void k_first_to_spans(int64_t n, const int32_t* __restrict__ start, const int32_t* __restrict__ end,
                                 const int64_t* __restrict__ prefix, int32_t* __restrict__ spans, int64_t span_cap) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (start[i] >= 0 && prefix[i] < span_cap) *(int2*)(spans + 2 * prefix[i]) = make_int2(start[i], end[i]);
}
__global__ void k_span_to_flag(int64_t n, const int32_t* __restrict__ start, uint8_t* __restrict__ flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    flag[i] = start[i] >= 0 ? 1 : 0;
}

enum { FA_COUNT = 0, FA_EMIT = 1 };
// STEP_SLOTS: count, and park the first kStepSlots spans of every text in a per-text slot row, so
// that findall usually needs ONE walk over the batch (k_slots_gather moves the rows to their CSR
// place); STEP_EMIT then only re-walks the texts that have more matches than slots.
// STEP_ANY (k_mwalk only): counts[i] = 1 when text i holds a match, else 0; a text is left at its first match.
enum { STEP_COUNT = 0, STEP_EMIT = 1, STEP_SEARCH = 2, STEP_SLOTS = 3, STEP_ANY = 4 };
constexpr int kStepSlots = 32;

// Windowed stepper for PF_STEPPABLE plans.  Same results as for_each_match() / hybrid_match_next()
// on those plans, but (1) ONE loop whose every iteration looks at one byte of every lane's text --
// skipping to a candidate start and walking the table are two short arms of the same iteration
// instead of two inner loops that make the lanes of a wavefront take turns -- and (2) the text
// bytes come from an LDS tile that the wavefront fills with coalesced 128-byte rows (one per text,
// frame form as in the streaming kernel), one window after the other.  A lane whose restart
// position falls behind the window reads those few bytes from global memory; a lane that has run
// past the window waits for the next one.  The per-step cost drops from ~160 instructions
// (register window, 64-bit addressing and bounds logic of the generic Text; rocprof: 92 VALU + 67
// SALU instructions per byte step on config 4) to ~30.
constexpr int kWsWaves = 4;
constexpr uint32_t kWsDead = 0xFFFFu, kWsAcc = 0x8000u, kWsStart = 0x4000u, kWsFirst = 0x2000u;
__host__ __device__ inline size_t wstep_table_bytes(int nstates) { return (size_t)(nstates + 1) * 512; }

// ROUTE 1 = HybridMatcher._match_all_required_byte (matcher.mojo:864-898) as the same kind of
// flattened machine: SCAN forward for the required byte while tracking where the current run of
// first-class bytes began (rsb); on a hit jump back to that run start and WALK the table from
// there (DFAEngine.match_first, anchored); keep the match if it ends past the hit and resume at
// its end, else resume at hit + 1.  (The required byte is not in the first class, so the run
// start at hit + 1 is hit + 1; the run start at a match end is remembered when `last` moves.)
// BITS = 1 (PF_BSTEP, ROUTE 0 only): the bitset-NFA form of the same loop for LazyDFA plans that have no
// determinised table (PikeVM semantics, pikevm.mojo:497-648 / 754-867).  The state of a walk is the SET of
// live program positions (one 64-bit word per lane; the empty set = "looking for a start"), and a step is
//     x = set & mask[byte];   next = OR over the 8-bit chunks j of x of  follow8[j][chunk j of x]
// with mask[b] = positions whose instruction consumes byte b and follow8[j][v] = union of the closures
// after the positions 8 j + k, k in v (both built in LDS per workgroup from the plan's per-position
// follow sets): one mask read and ceil(positions / 8) table reads per byte, whatever the number of live
// positions and without any data-dependent loop -- the lanes of a wavefront stay in step.
__host__ __device__ inline size_t bstep_table_bytes(int npos) { return 2048 + (size_t)((npos + 7) / 8) * 2048 + 256; }

// EMPTY = 1 (plain route of a table plan whose start state accepts and that has no first-byte matcher:
// DFAEngine.match_all's last loop, dfa.mojo:2118-2130; LazyDFA.match_all without a filter, pikevm.mojo:805-817):
// every position is tried and every try matches -- the longest walk from it, or the empty match when no walk
// starts there -- and the search resumes at the match end, or one byte on after an empty match; the position
// behind the last byte is tried as well.
// BM = 1 (plain route of a table plan with a backward table, PF_BACKSET): the positions at which a match begins are
// marked (Layout::bm, written by k_backscan), so a lane that is looking for a start jumps to the next mark -- up to 32
// bytes per step -- and every walk it begins succeeds.
// CLSIDX = 1 (with BM, tables of more than 96 states -- PF_STEP_BIG): the table stays class indexed, cls[256] |
// tr[(nstates + 1) x ncls] (last row = "a walk begins": the start state's transitions; which bytes may begin one is
// in the marks already), two dependent LDS reads per step instead of one -- affordable now that no walk fails.
// LZ = 1 (PF_LAZY_END: '$' program on the LazyDFA search): the plan's table holds every state twice --
// rows nstates/2.. are the transitions as computed while the text's last byte is consumed -- and the lane keeps the
// two state masks of walk_lazy_end() (mrx_device.hpp): which variant a (state, last-byte-value) pair was first
// computed in decides the row for the rest of the text.  The idle row exists twice as well (its entry is the
// start state's transition).
template <int MODE, int ROUTE, int BITS = 0, int EMPTY = 0, int BM = 0, int CLSIDX = 0, int LZ = 0>
__global__ __launch_bounds__(64 * kWsWaves) void k_wstep(DevPlan p, const uint8_t* __restrict__ blob, Layout lay,
                                                         int64_t n, int32_t* __restrict__ counts,
                                                         const int64_t* __restrict__ prefix,
                                                         int32_t* __restrict__ spans, int64_t span_cap,
                                                         int32_t* __restrict__ out_s, int32_t* __restrict__ out_e) {
  constexpr int CH = 128, kRowPitch = CH + 16, LPR = CH / 16, RPI = 64 / LPR, NL = 64 / RPI;
  __shared__ __align__(16) uint8_t tiles[kWsWaves][64 * kRowPitch];
  extern __shared__ __align__(16) uint8_t lds[];
  // One byte-indexed table for the whole step: row q < nstates is DFA state q, row nstates is
  // "looking for a start".  Entry = next | ACC | START, or DEAD.  (Built from the plan's class
  // tables, so a step costs one dependent LDS read instead of three.)
  static_assert(!(BITS && ROUTE), "the bitset form has the plain route only");
  static_assert(!EMPTY || (!BITS && !ROUTE), "empty matches: plain route of a table plan");
  static_assert(!BM || (!BITS && !ROUTE && !EMPTY), "marks: plain route of a table plan");
  static_assert(!CLSIDX || BM, "the class-indexed table needs the marks (no first-byte filter in it)");
  static_assert(!LZ || (!ROUTE && !BITS && !EMPTY && !BM), "per-text lazy cache: plain route of a table plan");
  uint16_t* tab = (uint16_t*)lds;
  const uint8_t* clsT = lds;                       // CLSIDX: cls[256] | tr[(ns + 1) x ncls]
  uint16_t* trc = (uint16_t*)(lds + 256);
  const int ns = p.nstates, idle = ns;
  // BITS: mask[256] | follow8[nch][256] | first-byte filter[256]
  uint64_t* bmask = (uint64_t*)lds;
  const int nch = (p.bs_npos + 7) >> 3;
  uint64_t* bfol = bmask + 256;
  uint8_t* bfirst = (uint8_t*)(bfol + (size_t)nch * 256);
  const uint64_t bstart = p.bs_start[0], bmatch = p.bs_match[0];
  if (BITS) {
    const uint8_t* g_bcls = blob + p.off_bs_cls;
    const uint64_t* g_mask = (const uint64_t*)(blob + p.off_bs_mask);
    const uint64_t* g_fol = (const uint64_t*)(blob + p.off_bs_follow);
    const uint8_t* g_first = blob + p.off_first;
    const bool filt = (p.flags & PF_HAS_MATCHER) != 0;   // LazyDFA has_filter (pikevm.mojo:367-416)
    for (int e = threadIdx.x; e < 256; e += blockDim.x) {
      bmask[e] = g_mask[g_bcls[e]];
      bfirst[e] = filt ? g_first[e] : (uint8_t)1;
    }
    for (int e = threadIdx.x; e < nch * 256; e += blockDim.x) {
      const int j = e >> 8, v = e & 255;
      uint64_t u = 0;
      for (int k = 0; k < 8; ++k)
        if (((v >> k) & 1) && 8 * j + k < p.bs_npos) u |= g_fol[8 * j + k];
      bfol[e] = u;
    }
  } else if (CLSIDX) {
    const uint8_t* g_cls = blob + p.off_cls;
    const uint16_t* g_tr = (const uint16_t*)(blob + p.off_trans);
    for (int e = threadIdx.x; e < 256; e += blockDim.x) lds[e] = g_cls[e];
    for (int e = threadIdx.x; e < (ns + 1) * p.ncls; e += blockDim.x) {
      const int q = e / p.ncls, c = e - q * p.ncls;
      const uint32_t t = g_tr[(q < ns ? q : 0) * p.ncls + c];
      uint32_t v = t == 0xFFFFu ? (q < ns ? kWsDead : (uint32_t)idle) : ((t & 0x7FFFu) | ((t & 0x8000u) ? kWsAcc : 0u));
      if (q == ns && t != 0xFFFFu) v |= kWsStart;
      trc[e] = (uint16_t)v;
    }
  } else
  {
    const uint8_t* g_cls = blob + p.off_cls;
    const uint8_t* g_first = blob + p.off_first;
    const uint16_t* g_tr = (const uint16_t*)(blob + p.off_trans);
    const bool filt = (p.flags & PF_HAS_MATCHER) != 0;
    for (int e = threadIdx.x; e < (ns + 1) * 256; e += blockDim.x) {
      const int q = e >> 8, b = e & 255;
      uint32_t v;
      if (q < ns) {
        const uint32_t t = g_tr[q * p.ncls + g_cls[b]];
        v = t == 0xFFFFu ? kWsDead : ((t & 0x7FFFu) | ((t & 0x8000u) ? kWsAcc : 0u));
        if (ROUTE == 1 && v != kWsDead && g_first[b]) v |= kWsFirst;   // walk steps keep tracking the run start
      } else if (ROUTE == 1) {   // SCAN row: 2 = the required byte, 1 = first-class byte, 0 = other
        v = b == p.required_byte ? 2u : g_first[b] ? 1u : 0u;
      } else {   // find_first_class + the first step of the walk it starts
        const uint32_t t = g_tr[0 * p.ncls + g_cls[b]];
        v = ((filt && !g_first[b]) || t == 0xFFFFu) ? (uint32_t)idle
                                                    : ((t & 0x7FFFu) | ((t & 0x8000u) ? kWsAcc : 0u) | kWsStart);
      }
      tab[e] = (uint16_t)v;
    }
    if (LZ)   // row ns + 1: looking for a start, the start state's transition in its "at the end" variant
      for (int b = threadIdx.x; b < 256; b += blockDim.x) {
        const uint32_t t = g_tr[(ns >> 1) * p.ncls + g_cls[b]];
        tab[((ns + 1) << 8) + b] = (uint16_t)(((filt && !g_first[b]) || t == 0xFFFFu)
                                                  ? (uint32_t)idle : ((t & 0x7FFFu) | ((t & 0x8000u) ? kWsAcc : 0u) | kWsStart));
      }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* tile = tiles[wave];
  const int seg = lane % LPR, rsub = lane / LPR;
  const int64_t nw = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * kWsWaves + wave; w < nw; w += (int64_t)gridDim.x * kWsWaves) {
    const int64_t i = (w << 6) + lane;
    const bool live = i < n;
    const Text t = live ? lay.text(i) : Text(blob, 0);
    const uintptr_t addr = t.len > 0 ? (uintptr_t)t.ptr : (uintptr_t)blob;   // empty rows park on the blob
    const int mis = t.len > 0 ? (int)(addr & 15) : 0;
    const uintptr_t rb = addr & ~(uintptr_t)15;
    const int end = mis + t.len;   // frame coordinates: the text is [mis, end)
    __builtin_amdgcn_wave_barrier();
    *(uint4*)(tile + lane * kRowPitch + CH) = make_uint4((uint32_t)rb, (uint32_t)((uint64_t)rb >> 32), (uint32_t)end, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int max_end = end;
    for (int off = 32; off > 0; off >>= 1) max_end = max(max_end, __shfl_xor(max_end, off));

    int pos = mis, start = mis, last = -1, k = 0, rs = -1, re = -1;
    int rsb = mis, rsb_last = mis, hit = -1;   // ROUTE 1: run start of first-class bytes (now / at `last`), last hit
    int state = idle;
    uint64_t bset = 0;   // BITS: live positions of the current walk (0 = looking for a start)
    const bool skipped = lay.split > 0 && t.len >= lay.split;   // k_req_wave's text
    bool fin = !live || t.len == 0 || skipped;
    uint64_t lz_mid = 0, lz_end = 0;     // LZ: (state, last byte value) pairs first computed inside the text / on its last byte
    const int lz_byte = (LZ && live && t.len > 0) ? (int)t.ptr[t.len - 1] : -1;
    // LZ: the last FAILED walk, walked again one byte per step beside whatever the lane does now (see the step below):
    // its state at `pos` (-1: none), and the state the current walk had behind its first byte
    int lz_sh = -1, lz_s1 = 0;
    const uint32_t* mybm = nullptr;      // BM: my text's marks (frame coordinates)
    int bm_idx = -1;
    uint32_t bm_word = 0;
    uint4 bm_win = make_uint4(0, 0, 0, 0);   // the four words of the current window
    int bm_wb = -(1 << 30);
    bool bm_none = false;
    if (BM && live) {
      mybm = lay.bm + lay.bm_row(i);
      if (lay.bm_cnt[i] == 0) fin = true;   // no match begins anywhere in my text
    }
    int64_t wo = (MODE == STEP_EMIT && live) ? prefix[i] : 0;
    int slot_cap = kStepSlots;           // my slot row (wide rows: sized by the text, Layout::slot_row)
    int64_t slot0 = i * kStepSlots;
    if ((MODE == STEP_EMIT || MODE == STEP_SLOTS) && lay.wide_slots == 1 && live) slot0 = lay.slot_row(i, &slot_cap);
    if ((MODE == STEP_EMIT || MODE == STEP_SEARCH) && counts) {
      // EMIT after STEP_SLOTS: only texts that overflowed their slots.  wide_slots == 2 (the two-pass findall of these
      // plans; search behind a STEP_ANY pass): counts come from a first pass -- texts without a match are not scanned again
      if (live && (lay.wide_slots == 2 ? counts[i] == 0 : counts[i] <= slot_cap)) fin = true;
      if (__all(fin)) {
        if (MODE == STEP_SEARCH && live && !skipped) { out_s[i] = -1; out_e[i] = -1; }
        continue;
      }
    }
    if (MODE == STEP_SLOTS) wo = 0;
    // (no match begins in any of the 64 texts: nothing to load.  Not a `continue`: that costs this loop 40 registers)
    if (BM && __all(fin)) max_end = 0;
    const uint8_t* myrow = tile + lane * kRowPitch;
    const uint8_t* frame = (const uint8_t*)rb;   // frame position f is frame[f] in global memory
    uint4 back_w = make_uint4(0, 0, 0, 0);        // the 16-byte block of the frame a lane behind the window reads from
    int back_idx = -1;
    uint4 v[NL];
#define MRX_WS_LOAD(CB)                                                                   \
    do {                                                                                  \
      _Pragma("unroll") for (int j_ = 0; j_ < NL; ++j_) {                                  \
        const uint4 rs_ = *(const uint4*)(tile + (RPI * j_ + rsub) * kRowPitch + CH);      \
        uint32_t fo_ = (uint32_t)(CB) + seg * 16;                                          \
        if (fo_ >= rs_.z) fo_ = 0;                                                         \
        v[j_] = mrx_ldg((const uint4*)((const uint8_t*)(((uint64_t)rs_.y << 32) | rs_.x) + fo_)); \
      }                                                                                    \
    } while (0)
    if (max_end > 0) MRX_WS_LOAD(0);
    for (int wb = 0; wb < max_end; wb += CH) {
#pragma unroll
      for (int j = 0; j < NL; ++j) *(uint4*)(tile + (RPI * j + rsub) * kRowPitch + seg * 16) = v[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (wb + CH < max_end) MRX_WS_LOAD(wb + CH);   // next window, in flight while this one is stepped
      if (BM) {   // this window's marks: one aligned 16-byte load per lane
        if (!fin && wb < end) bm_win = *(const uint4*)(mybm + (wb >> 5));
        bm_wb = wb;
        bm_none = (bm_win.x | bm_win.y | bm_win.z | bm_win.w) == 0;
      }
      // One step of the search for this lane, given the byte at `pos` (ignored when act is false
      // or the text has ended).  Branch-free apart from the span store.
      auto step = [&](bool act, uint32_t byte) {
        const bool inside = pos < end;
        if (BITS) {
          const bool idl = bset == 0ull;
          const uint64_t src = idl ? (bfirst[byte] ? bstart : 0ull) : bset;   // a candidate byte starts a walk
          const uint64_t x = src & bmask[byte];
          uint64_t nx = 0;
          for (int j = 0; j < nch; ++j) nx |= bfol[(j << 8) + (int)((x >> (8 * j)) & 0xFFull)];
          const bool alive = act && inside && nx != 0ull;        // a walk goes on, or begins on this byte
          const bool skip = act && inside && idl && nx == 0ull;   // no walk starts here
          const bool stop = act && !alive && !skip;              // the text has ended, or the walk died
          const bool ends = stop && !idl;
          const bool matched = ends && last >= 0;
          if (MODE == STEP_EMIT) {
            if (matched) {
              if (wo < span_cap) *(int2*)(spans + 2 * wo) = make_int2(start - mis, last - mis);
              ++wo;
            }
          }
          if (MODE == STEP_SLOTS) {
            if (matched) {
              if (wo < slot_cap) *(int2*)(spans + 2 * (slot0 + wo)) = make_int2(start - mis, last - mis);
              ++wo;
            }
          }
          if (MODE == STEP_SEARCH) { rs = matched ? start - mis : rs; re = matched ? last - mis : re; }
          fin = fin || (stop && idl) || (MODE == STEP_SEARCH && matched);
          k += matched ? 1 : 0;
          const int after = matched ? last : start + 1;
          const bool begins = alive && idl;
          start = begins ? pos : start;
          last = begins ? -1 : last;
          const int nxt = pos + 1;
          last = (alive && (nx & bmatch)) ? nxt : last;   // LazyDFA is_match of the state entered (pikevm.mojo:834-835, 861-862)
          pos = (alive || skip) ? nxt : (ends ? after : pos);
          bset = alive ? nx : (stop ? 0ull : bset);   // (a lane that is not stepping keeps its walk)
          return;
        }
        if (BM) {
          if (act && state == idle && inside) {   // looking for a start: on to the next mark
            uint32_t wv;
            const int rel = pos - bm_wb;
            if (rel >= 0 && rel < CH) {   // the window's marks are in registers
              if (bm_none) { pos = bm_wb + CH; return; }   // none in the whole window: out of it in one step
              const int j = rel >> 5;
              wv = j == 0 ? bm_win.x : j == 1 ? bm_win.y : j == 2 ? bm_win.z : bm_win.w;
            } else {
              if ((pos >> 5) != bm_idx) { bm_idx = pos >> 5; bm_word = mybm[bm_idx]; }
              wv = bm_word;
            }
            const uint32_t rest = wv >> (pos & 31);
            if (!(rest & 1u)) {
              pos += rest ? __builtin_ctz(rest) : 32 - (pos & 31);
              return;
            }
          }
        }
        int row = state;
        if (LZ && act && inside && (int)byte == lz_byte) {
          // the transition of DFA state st0 on the text's last byte value: as the lazy cache has it by now
          const int st0 = state == idle ? 0 : state;
          const uint64_t bit = 1ull << st0;
          bool at_end = (lz_end & bit) != 0;
          if (!at_end && !(lz_mid & bit)) {   // first use: computed here, with '$' iff this is the last byte
            at_end = pos + 1 == end;
            if (at_end) lz_end |= bit; else lz_mid |= bit;
          }
          if (at_end) row = state == idle ? idle + 1 : (ns >> 1) + state;
        }
        const uint32_t e = CLSIDX ? trc[state * p.ncls + clsT[byte]] : tab[(row << 8) + byte];
        // LZ.  Upstream restarts a failed walk one byte on and walks again: quadratic where walks are long and fail (a
        // kilobyte of letters in front of a byte that is no digit: 0.6 GB/s on bench.py's mix in round 3).  A walk's
        // future depends on (position, state) only -- the cache is monotone, and every pair a walk met is decided for
        // good -- so the last failed walk R is stepped AGAIN beside the lane (its pairs are all decided: it reads the
        // masks, never writes them), and the moment the current walk W stands in R's state at R's position, W's
        // future is R's: no accepting state any more.  W then ends here -- failed, or matched up to its last accepting
        // position -- instead of at the byte that killed R.  In a run of letters that is the restart's first step.
        int sh_next = -1;
        if (LZ && act && inside && lz_sh >= 0) {
          int row2 = lz_sh;
          if ((int)byte == lz_byte && ((lz_end >> lz_sh) & 1ull)) row2 = (ns >> 1) + lz_sh;
          const uint32_t e2 = tab[(row2 << 8) + byte];
          sh_next = e2 == kWsDead ? -1 : (int)(e2 & 0x3FFFu);
        }
        if (ROUTE == 1) {
          const bool scanning = state == idle;
          // SCAN
          const bool s_end = act && scanning && !inside;
          const bool s_hit = act && scanning && inside && e == 2u;
          const bool s_adv = act && scanning && inside && e != 2u;
          // WALK
          const bool alive = act && !scanning && inside && e != kWsDead;
          const bool ends = act && !scanning && !alive;
          const bool matched = ends && last > hit;           // must end past the required byte
          if (MODE == STEP_EMIT) {
            if (matched) {
              if (wo < span_cap) *(int2*)(spans + 2 * wo) = make_int2(start - mis, last - mis);
              ++wo;
            }
          }
          if (MODE == STEP_SLOTS) {
            if (matched) {
              if (wo < slot_cap) *(int2*)(spans + 2 * (slot0 + wo)) = make_int2(start - mis, last - mis);
              ++wo;
            }
          }
          fin = fin || s_end;
          k += matched ? 1 : 0;
          const int nxt = pos + 1;
          // run-start tracking: a non-first-class byte moves it past itself
          const bool moves = (s_adv && e == 0u) || (alive && !(e & kWsFirst));
          const int rsb_n = moves ? nxt : rsb;
          if (alive && (e & kWsAcc)) { last = nxt; rsb_last = rsb_n; }
          if (s_hit) { hit = pos; start = rsb; last = -1; }
          pos = (s_adv || alive) ? nxt : s_hit ? rsb : ends ? (matched ? last : hit + 1) : pos;
          rsb = ends ? (matched ? rsb_last : hit + 1) : rsb_n;
          state = s_hit ? 0 : alive ? (int)(e & 0x1FFFu) : ends ? idle : state;
          return;
        }
        const bool alive = act && inside && e != kWsDead;
        const bool stop = act && !alive;                     // dead entry, or the text has ended
        const bool ends = stop && state != idle;             // a walk stops here
        // EMPTY: a walk always leaves a match (its start accepts); an idle step on a byte that starts no walk
        // and the idle step behind the last byte leave the empty match at `pos`
        const bool empty_here = EMPTY && act && state == idle && (inside ? !(e & kWsStart) : true);
        const bool matched = (ends && last >= 0) || empty_here;   // (without EMPTY matches are never empty)
        const int m_s = empty_here ? pos : start, m_e = empty_here ? pos : last;
        if (MODE == STEP_EMIT) {
          if (matched) {
            if (wo < span_cap) *(int2*)(spans + 2 * wo) = make_int2(m_s - mis, m_e - mis);
            ++wo;
          }
        }
        if (MODE == STEP_SLOTS) {   // spans = slot rows [n][kStepSlots][2]
          if (matched) {
            if (wo < slot_cap) *(int2*)(spans + 2 * (slot0 + wo)) = make_int2(m_s - mis, m_e - mis);
            ++wo;
          }
        }
        if (MODE == STEP_SEARCH) { rs = matched ? m_s - mis : rs; re = matched ? m_e - mis : re; }
        fin = fin || (stop && state == idle) || (MODE == STEP_SEARCH && matched);
        k += matched ? 1 : 0;
        // where the search resumes after a walk (EMPTY: one byte on when the walk only left the empty match)
        const int after = EMPTY ? (last > start ? last : start + 1) : (matched ? last : start + 1);
        const bool begins = alive && (e & kWsStart);
        start = begins ? pos : start;
        last = begins ? (EMPTY ? pos : -1) : last;
        const int nxt = pos + 1;
        last = (alive && (e & kWsAcc)) ? nxt : last;
        pos = alive ? nxt : (ends ? after : pos);
        state = alive ? (int)(e & 0x3FFFu) : (stop ? idle : state);
        if (LZ) {
          if (begins) lz_s1 = state;
          const bool merged = alive && state != idle && sh_next == state;   // W stands where R stood
          if (merged) {
            const bool m = last >= 0;   // W has accepted: [start, last) is its match, nothing behind this byte adds to it
            if (m) {
              if (MODE == STEP_EMIT) {
                if (wo < span_cap) *(int2*)(spans + 2 * wo) = make_int2(start - mis, last - mis);
                ++wo;
              }
              if (MODE == STEP_SLOTS) {
                if (wo < slot_cap) *(int2*)(spans + 2 * (slot0 + wo)) = make_int2(start - mis, last - mis);
                ++wo;
              }
              if (MODE == STEP_SEARCH) { rs = start - mis; re = last - mis; fin = true; }
              ++k;
            }
            pos = m ? last : start + 1;
            state = idle;
            lz_sh = m ? -1 : lz_s1;   // a failed W is the next R: at start + 1 it stood in lz_s1
          } else if (act) {
            // W failed by itself: it is the next R.  A match: nothing to walk beside.  Otherwise R moves on with the lane.
            lz_sh = (ends && !matched) ? lz_s1 : matched ? -1 : sh_next;
          }
        }
      };
      while (true) {
        // fast phase: 32 steps on the tile without any cross-lane vote; lanes that are ahead of the
        // window, behind it or finished run the same instructions as no-ops
#pragma unroll 1
        for (int blk = 0; blk < 4; ++blk) {
#pragma unroll 1
          for (int it = 0; it < 8; ++it) {
            const int rel = pos - wb;
            const bool act = !fin && rel >= 0 && (rel < CH || pos >= end);
            step(act, (uint32_t)myrow[rel & (CH - 1)]);
          }
          // (marks: a lane without a mark in the window leaves it in one step -- do not idle through the other steps)
          if (BM && !__any(!fin && pos >= wb && (pos < wb + CH || pos >= end))) break;
        }
        // a restart moved some lane behind the window: it reads those bytes from memory -- 16 at a time into
        // registers (round 4; a dependent byte load per step before: a lane that came back from a long failed walk
        // paid a memory round trip for every byte up to the window)
        while (__any(!fin && pos < wb)) {
          const bool act = !fin && pos < wb;
          uint32_t byte = 0u;
          if (act && pos >= mis) {
            if ((pos >> 4) != back_idx) { back_idx = pos >> 4; back_w = *(const uint4*)(frame + ((uint32_t)pos & ~15u)); }
            const int j = (pos >> 2) & 3;
            const uint32_t wv = j == 0 ? back_w.x : j == 1 ? back_w.y : j == 2 ? back_w.z : back_w.w;
            byte = (wv >> (8 * (pos & 3))) & 0xFFu;
          }
          step(act, byte);
        }
        if (!__any(!fin && (pos < wb + CH || pos >= end))) break;
      }
      __builtin_amdgcn_wave_barrier();
      if (__all(fin)) break;
    }
#undef MRX_WS_LOAD
    if (EMPTY && live && !skipped && t.len == 0) {   // the empty text: one try, at 0, and it matches
      k = 1; rs = re = 0;
      if (MODE == STEP_EMIT && !(counts && counts[i] <= slot_cap) && wo < span_cap) { spans[2 * wo] = 0; spans[2 * wo + 1] = 0; }
      if (MODE == STEP_SLOTS) *(int2*)(spans + 2 * slot0) = make_int2(0, 0);
    }
    if (live && !skipped) {
      if (MODE == STEP_COUNT || MODE == STEP_SLOTS) counts[i] = k;
      if (MODE == STEP_SEARCH) { out_s[i] = rs; out_e[i] = re; }
    }
  }
}

// ---- several walks in one pass (PF_MWALK, DevPlan::off_mw_*; host side: build_multiwalk() in mrx_plan.cpp) ------
// The stepper above re-scans: a walk that fails at byte q after starting at p sends the lane back to p + 1, and a
// match that ends before the byte that killed its walk sends it back to the match end.  Here the walks the
// reference would start one after the other run side by side -- up to four per text, oldest first, the whole list
// folded into ONE table state (a "configuration": has the oldest walk accepted + the DFA states of the live walks in
// age order) -- so every byte is looked at once, every lane steps in lockstep and nothing ever moves backwards.
// Per byte: class lookup, one dependent read of tab[configuration][class], and a few selects that move the walks'
// START registers as the entry says (a walk that leaves the list lets the younger ones move up, a walk that begins
// on the byte takes the first free slot); `last` = the position behind the oldest walk's latest accepting state.
// EMIT (entry bit 0) reports [start of slot 0, last) BEFORE the registers move.  At the end of the text the oldest
// walk reports if it has accepted (entry bit 10 of the last step).  Same modes and arguments as k_wstep, so the
// host swaps the kernel and nothing else (slot rows, second walk for overflowing texts, search, count).
__host__ __device__ inline size_t mwalk_table_bytes(const DevPlan& p) { return (size_t)p.mw_bytes; }

// KW: walk slots the plan needs (DevPlan::mw_k <= 2: two start registers and three selects per byte instead of ten)
// PK (texts below 64 KiB; modes that keep the registers): the starts as 16-bit halves of two registers and `last` in
// a third, moved by byte permutes -- v_perm_b32 picks every byte of its result from two source registers, so
// "slot 0 takes slot 2's start, slot 1 the current position" is ONE instruction with the right selector.  The
// selectors come from a table indexed by the entry's nine code bits (accepting | slot codes): one 8- or 16-byte LDS
// read and two (KW = 2) or four (KW = 4) permutes per byte instead of 7-24 compares and selects.
//   KW = 2:  R01 = perm(prpr, R01, selR)                                       prpr = position in both halves
//   KW = 4:  R01 = perm(prpr, perm(R23, R01, selA), selB),  R23 = perm(prpr, R23, selC)
//   both:    L   = perm(prpr + 0x10001, L, selL)                               (low half = last)
// perm(a, b, sel): result byte k = byte sel[k] of the eight bytes b (0-3) | a (4-7).
// EMP = 1 (PF_MW_EMPTY plans, build_emptywalk(): one walk, every state accepts): entry bit 11 = the empty match at this
// byte (reported behind the match the byte ends, if it ends one), and every text ends with the empty match at its length.
// EMP = 2 (DevPlan::mw_k == -2, build_emptywalk2()): walks that read up to fourteen bytes beyond their match.  32-byte
// entries (EwEntry): word 0 bit 0 report (start, last) registers, bit 1 last = pos + 1, bits 2-5 / 6-9 the try that takes
// over as the oldest walk (start = pos - a, last = start + len), bits 10-14 how many dead tries are reported, bits 16..
// next row; words 1-7: their (a, len), eight bits each.  end[config] behind the table: the same at the end of the text.
// EMP = 3 (mw_k == -3, PF_MW_TRIES): the same table form for a plan without empty matches (no try at len).
template <int MODE, int KW = 4, int PK = 0, int EMP = 0>
__global__ __launch_bounds__(64 * kWsWaves) void k_mwalk(DevPlan p, const uint8_t* __restrict__ blob, Layout lay,
                                                         int64_t n, int32_t* __restrict__ counts,
                                                         const int64_t* __restrict__ prefix,
                                                         int32_t* __restrict__ spans, int64_t span_cap,
                                                         int32_t* __restrict__ out_s, int32_t* __restrict__ out_e) {
  constexpr int CH = 128, kRowPitch = CH + 16, LPR = CH / 16, RPI = 64 / LPR, NL = 64 / RPI;
  static_assert(!PK || ((KW == 2 || KW == 4) && MODE != STEP_COUNT && MODE != STEP_ANY), "packed starts: two or four slots, modes with registers");
  static_assert(!EMP || (KW == 2 && !PK && (MODE == STEP_COUNT || MODE == STEP_EMIT || (MODE == STEP_SEARCH && EMP == 3))),
                "empty-match / pending-tries walk: count and emit passes; search for plans without empty matches");
  static_assert(EMP >= 0 && EMP <= 3, "0: multi-walk table, 1: one walk that never overshoots, 2: walks with pending tries behind them, 3: the same for plans without empty matches");
  constexpr bool TRIES = EMP == 2 || EMP == 3;     // 128-bit entries (EwEntry)
  constexpr bool LAST_TRY = EMP == 1 || EMP == 2;  // empty matches: the last try is at pos == len
  __shared__ __align__(16) uint8_t tiles[kWsWaves][64 * kRowPitch];
  __shared__ __align__(16) uint32_t plut[PK ? 512 * (KW == 2 ? 2 : 4) : 4];   // PK: the permute selectors per code
  if (PK) {
    for (int c = threadIdx.x; c < 512; c += blockDim.x) {
      const uint32_t a = c & 1, c0 = (c >> 1) & 7, c1 = (c >> 4) & 3, c2 = (c >> 6) & 3, c3 = (c >> 8) & 1;
      auto half = [](uint32_t b) { return b | ((b + 1) << 8); };   // the two byte indices of the half that begins at byte b
      const uint32_t selL = half(a ? 4 : 0) | (half(2) << 16);
      if (KW == 2) {
        const uint32_t n0 = c0 == 1 ? 2 : c0 == 4 ? 4 : 0, n1 = c1 == 3 ? 4 : 2;
        plut[2 * c] = half(n0) | (half(n1) << 16);
        plut[2 * c + 1] = selL;
      } else {
        const uint32_t a0 = c0 == 1 ? 2 : c0 == 2 ? 4 : c0 == 3 ? 6 : 0, a1 = c1 == 1 ? 4 : c1 == 2 ? 6 : 2;
        plut[4 * c] = half(a0) | (half(a1) << 16);                                       // selA: among the old starts
        plut[4 * c + 1] = half(c0 == 4 ? 4 : 0) | (half(c1 == 3 ? 4 : 2) << 16);          // selB: the position into slot 0 / 1
        plut[4 * c + 2] = half(c2 == 1 ? 2 : c2 == 2 ? 4 : 0) | (half(c3 == 1 ? 4 : 2) << 16);   // selC: slots 2, 3
        plut[4 * c + 3] = selL;
      }
    }
  }
  extern __shared__ __align__(16) uint8_t lds[];
  // PRE: the tables as LDS ADDRESSES -- cls4[byte] = 4 x class (at most 64 classes: it stays a byte), and an entry's
  // bits 16.. = LDS byte address of the next configuration's row instead of its index -- so that a byte's entry is at
  // (entry >> 16) + cls4[byte]: one add (the compiler folds the shift into its operand select) instead of shift,
  // shift, add3.  Needs the table to end below 64 KiB of LDS.
  typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
  typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds;   // (the low half of a flat LDS address is the LDS offset)
  const bool pre = !TRIES && p.mw_cshift <= 6 && lds_base + (uint32_t)p.mw_bytes <= 65536u;   // (TRIES: 128-bit entries, as stored)
  {
    const uint32_t* src = (const uint32_t*)(blob + p.off_mw_cls);   // cls[256] | tab[...], contiguous and 16-byte aligned
    uint32_t* dst = (uint32_t*)lds;
    for (int e = threadIdx.x; e < (p.mw_bytes >> 2); e += blockDim.x) {
      uint32_t w = src[e];
      if (pre) w = e < 64 ? (w << 2) & 0xFCFCFCFCu : (w & 0xFFFFu) | ((((w >> 16) << 2) + lds_base + 256u) << 16);
      dst[e] = w;
    }
  }
  __syncthreads();
  const uint8_t* clsT = lds;
  const uint32_t* tab = (const uint32_t*)(lds + 256);
  const uint32_t* tab64 = (const uint32_t*)(lds + 256);                                  // TRIES: 32-byte entries (EwEntry), eight words each
  const uint32_t* end64 = tab64 + (((size_t)p.mw_ncfg << p.mw_cshift) << 3);             // ... one per configuration behind them
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* tile = tiles[wave];
  const int seg = lane % LPR, rsub = lane / LPR;
  const int64_t nw = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * kWsWaves + wave; w < nw; w += (int64_t)gridDim.x * kWsWaves) {
    const int64_t i = (w << 6) + lane;
    const bool live = i < n;
    const Text t = live ? lay.text(i) : Text(blob, 0);
    const uintptr_t addr = t.len > 0 ? (uintptr_t)t.ptr : (uintptr_t)blob;   // empty rows park on the blob
    const int mis = t.len > 0 ? (int)(addr & 15) : 0;
    const uintptr_t rb = addr & ~(uintptr_t)15;
    const int end = mis + t.len;   // frame coordinates: the text is [mis, end)
    __builtin_amdgcn_wave_barrier();
    *(uint4*)(tile + lane * kRowPitch + CH) = make_uint4((uint32_t)rb, (uint32_t)((uint64_t)rb >> 32), (uint32_t)end, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int max_end = end;
    for (int off = 32; off > 0; off >>= 1) max_end = max(max_end, __shfl_xor(max_end, off));

    const bool skipped = lay.split > 0 && t.len >= lay.split;   // k_req_wave's text
    bool fin = !live || t.len == 0 || skipped;
    int64_t wo = (MODE == STEP_EMIT && live) ? prefix[i] : 0;
    int slot_cap = kStepSlots;
    int64_t slot0 = i * kStepSlots;
    if ((MODE == STEP_EMIT || MODE == STEP_SLOTS) && lay.wide_slots == 1 && live) slot0 = lay.slot_row(i, &slot_cap);
    if ((MODE == STEP_EMIT || MODE == STEP_SEARCH) && counts) {
      // EMIT after STEP_SLOTS: only texts that overflowed their slots.  wide_slots == 2 (the two-pass findall of these
      // plans; search behind a STEP_ANY pass): counts come from a first pass -- texts without a match are not scanned again
      if (live && (lay.wide_slots == 2 ? counts[i] == 0 : counts[i] <= slot_cap)) fin = true;
      if (!LAST_TRY && __all(fin)) {   // (empty-match plans: an empty text still has its empty match)
        if (MODE == STEP_SEARCH && live && !skipped) { out_s[i] = -1; out_e[i] = -1; }
        continue;
      }
    }
    if (MODE == STEP_SLOTS) wo = 0;
    uint32_t e = pre ? (lds_base + 256u) << 16 : 0u;   // the last entry taken: bits 16.. = row of the current configuration
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0, last = 0, k = 0, rs = -1, re = -1;
    uint32_t R01 = 0, R23 = 0, Lr = 0;   // PK: starts of slots 0 | 1, 2 | 3, last (16-bit halves)
    // EMIT / SLOTS: the lane's output row and how many spans it may still take, as a pointer and a 32-bit count (the
    // report branch is taken on nearly every byte step of a dense batch: 64-bit compares and adds in it cost 5 of 15)
    int2* const orow = (int2*)spans + (MODE == STEP_SLOTS ? slot0 : wo);
    const int64_t room64 = MODE == STEP_SLOTS ? (int64_t)slot_cap : span_cap - wo;
    const int oroom = room64 < 0 ? 0 : room64 > 0x7FFFFFFF ? 0x7FFFFFFF : (int)room64;
    const int vb = ((MODE == STEP_EMIT || MODE == STEP_SLOTS) && lay.vbase && live) ? lay.vbase[i] : 0;
    auto report = [&](int a, int b) {
      if (MODE == STEP_EMIT || MODE == STEP_SLOTS) {
        if (k < oroom) orow[k] = make_int2(a + vb, b + vb);
      }
      if (MODE == STEP_SEARCH) { rs = a; re = b; fin = true; }
      if (MODE == STEP_ANY) fin = true;
      ++k;
    };
    // EMP == 2: one entry -- W0's match, the dead tries the chase passes, the try that takes over, the next row
    auto ew2_apply = [&](const uint32_t* ent, const int base, const bool step) {
      const uint32_t x = ent[0];
      if (MODE == STEP_COUNT) {
        k += (int)(x & 1u) + (int)((x >> 10) & 31u);
      } else {
        if (x & 1u) report(s0, last);
        const int nrep = (MODE == STEP_SEARCH && fin) ? 0 : (int)((x >> 10) & 31u);   // (search: the first report is the answer)
        for (int r = 0; r < nrep; ++r) {   // (rare: the oldest walk died with tries behind it, or no walk begins on this byte)
          const uint32_t rf = ent[1 + (r >> 2)] >> (8 * (r & 3));
          const int st = base - (int)(rf & 15u);
          report(st, st + (int)((rf >> 4) & 15u));
          if (MODE == STEP_SEARCH) break;
        }
        const int ta = (int)((x >> 2) & 15u);
        if (ta) { s0 = base - (ta - 1); last = s0 + (int)((x >> 6) & 15u); }
        else if (x & 2u) last = base + 1;
      }
      if (step) e = x;
    };
    const uint8_t* myrow = tile + lane * kRowPitch;
    uint4 v[NL];
#define MRX_MW_LOAD(CB)                                                                   \
    do {                                                                                  \
      _Pragma("unroll") for (int j_ = 0; j_ < NL; ++j_) {                                  \
        const uint4 rs_ = *(const uint4*)(tile + (RPI * j_ + rsub) * kRowPitch + CH);      \
        uint32_t fo_ = (uint32_t)(CB) + seg * 16;                                          \
        if (fo_ >= rs_.z) fo_ = 0;                                                         \
        v[j_] = mrx_ldg((const uint4*)((const uint8_t*)(((uint64_t)rs_.y << 32) | rs_.x) + fo_)); \
      }                                                                                    \
    } while (0)
    if (max_end > 0) MRX_MW_LOAD(0);
    for (int wb = 0; wb < max_end; wb += CH) {
#pragma unroll
      for (int j = 0; j < NL; ++j) *(uint4*)(tile + (RPI * j + rsub) * kRowPitch + seg * 16) = v[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (wb + CH < max_end) MRX_MW_LOAD(wb + CH);   // next window, in flight while this one is stepped
      // One byte.  ACT: is this lane stepping (a wavefront whose 64 texts all cover the 16-byte group, none of them
      // finished, steps without the test -- FULL); PRE_: the tables hold LDS addresses (above)
#define MRX_MW_BYTE(BYTE_, F_, FULL_, PRE_)                                                                          \
      do {                                                                                                           \
        const int f = (F_);                           /* frame position, the same for every lane */                  \
        if constexpr (TRIES) {   /* FULL_: every lane's text covers the group (or the lane is finished) */        \
          if (!fin && ((FULL_) || (f >= mis && f < end))) ew2_apply(tab64 + (((e >> 16) + clsT[(BYTE_)]) << 3), f - mis, true); \
          break;                                                                                                     \
        }                                                                                                            \
        /* FULL_ = 2 (search): a lane that has its answer keeps stepping on whatever its row holds -- its registers   \
           are dead, only the report is guarded */                                                                   \
        const bool act = (FULL_) == 2 ? !fin : ((FULL_) || (!fin && f >= mis && f < end));                           \
        uint32_t en;                                                                                                 \
        if (PRE_) en = *(lds_cu32*)(uintptr_t)((e >> 16) + *(lds_cu8*)(uintptr_t)(lds_base + (BYTE_)));               \
        else en = tab[(e >> 16) + clsT[(BYTE_)]];                                                                    \
        const int pr = f - mis;                       /* text position of this byte */                               \
        if (PK) { if (act && (en & 1u)) report((int)(R01 & 0xFFFFu), (int)(Lr & 0xFFFFu)); }                         \
        else                                                                                                         \
        if (act && (en & 1u)) report(s0, last);      /* the oldest walk ended behind its last accepting position (rare branch) */ \
        if (EMP == 1) { if (act && (en & 0x800u)) report(pr, pr); }   /* ... and no walk begins on this byte: the empty match here */ \
        /* the start registers move as the entry says; plain selects, no branches (code 0 = stays; a lane that is    \
           not stepping takes code 0 everywhere) */                                                                  \
        const uint32_t ea = ((FULL_) == 2 || act) ? en : 0u;                                                         \
        if (PK) {                                                                                                    \
          const uint32_t prpr = (uint32_t)pr * 0x10001u;                                                             \
          const uint32_t* sel = plut + ((ea >> 1) & 0x1FFu) * (KW == 2 ? 2 : 4);                                     \
          if (KW == 2) {                                                                                             \
            const uint2 sv = *(const uint2*)sel;                                                                     \
            R01 = __builtin_amdgcn_perm(prpr, R01, sv.x);                                                            \
            Lr = __builtin_amdgcn_perm(prpr + 0x10001u, Lr, sv.y);                                                   \
          } else {                                                                                                   \
            const uint4 sv = *(const uint4*)sel;                                                                     \
            const uint32_t t01 = __builtin_amdgcn_perm(R23, R01, sv.x);                                              \
            R23 = __builtin_amdgcn_perm(prpr, R23, sv.z);                                                            \
            R01 = __builtin_amdgcn_perm(prpr, t01, sv.y);                                                            \
            Lr = __builtin_amdgcn_perm(prpr + 0x10001u, Lr, sv.w);                                                   \
          }                                                                                                          \
        } else                                                                                                       \
        if (MODE != STEP_COUNT && MODE != STEP_ANY) {                                                                \
        const uint32_t c0 = (ea >> 2) & 7u, c1 = (ea >> 5) & 3u;                                                     \
        int n0 = s0, n1 = s1;                                                                                        \
        n0 = c0 == 1u ? s1 : n0;                                                                                     \
        n0 = c0 == 4u ? pr : n0;                                                                                     \
        n1 = c1 == 3u ? pr : n1;                                                                                     \
        if (KW > 2) {                                                                                                \
          const uint32_t c2 = (ea >> 7) & 3u;                                                                        \
          int n2 = s2;                                                                                               \
          n0 = c0 == 2u ? s2 : n0;                                                                                   \
          n1 = c1 == 1u ? s2 : n1;                                                                                   \
          n2 = c2 == 2u ? pr : n2;                                                                                   \
          if (KW > 3) {                                                                                              \
            const uint32_t c3 = (ea >> 9) & 1u;                                                                      \
            n0 = c0 == 3u ? s3 : n0;                                                                                 \
            n1 = c1 == 2u ? s3 : n1;                                                                                 \
            n2 = c2 == 1u ? s3 : n2;                                                                                 \
            s3 = c3 == 1u ? pr : s3;                                                                                 \
          }                                                                                                          \
          s2 = n2;                                                                                                   \
        }                                                                                                            \
        s0 = n0; s1 = n1;                                                                                            \
        last = (ea & 2u) ? pr + 1 : last;                                                                            \
        }                                                                                                            \
        e = ((FULL_) == 2 || act) ? en : e;                                                                          \
      } while (0)
      for (int g = 0; g < CH / 16; ++g) {
        const int f0 = wb + g * 16;
        const uint4 wv = *(const uint4*)(myrow + g * 16);
        const uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
        // (lanes that are finished -- a text without a match in the emit pass, a search that has its answer, an empty
        // text -- do not keep the wavefront off the fast paths: they step on whatever their row holds, unreported)
        const bool covers = f0 >= mis && f0 + 16 <= end;
        // (the count pass keeps its loop lean: one fast form, no finished lanes in it)
        const bool all_full = pre && MODE != STEP_ANY && __all(MODE == STEP_COUNT ? (!fin && covers) : (fin || covers));
        if (TRIES && __all(fin || covers)) {   // (the pending-tries walk: no frame test per byte where no text begins or ends)
#pragma unroll
          for (int q = 0; q < 16; ++q) MRX_MW_BYTE((words[q >> 2] >> ((q & 3) * 8)) & 0xFFu, f0 + q, 1, false);
        } else
        if (all_full && MODE != STEP_COUNT && (MODE == STEP_SEARCH || __any(fin))) {
#pragma unroll
          for (int q = 0; q < 16; ++q) MRX_MW_BYTE((words[q >> 2] >> ((q & 3) * 8)) & 0xFFu, f0 + q, 2, true);
        } else if (all_full) {
#pragma unroll
          for (int q = 0; q < 16; ++q) MRX_MW_BYTE((words[q >> 2] >> ((q & 3) * 8)) & 0xFFu, f0 + q, 1, true);
        } else if (pre) {
#pragma unroll
          for (int q = 0; q < 16; ++q) MRX_MW_BYTE((words[q >> 2] >> ((q & 3) * 8)) & 0xFFu, f0 + q, 0, true);
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q) MRX_MW_BYTE((words[q >> 2] >> ((q & 3) * 8)) & 0xFFu, f0 + q, 0, false);
        }
        if (MODE != STEP_COUNT && !all_full && !fin && f0 + 16 >= end) {
          // the text ended in this group: its last report now (the oldest walk has accepted), then the lane rides along
          if constexpr (TRIES) {
            ew2_apply(end64 + (((e >> 16) >> p.mw_cshift) << 3), t.len, false);   // every walk dies behind the last byte
          } else
          if ((e >> 10) & 1u) {
            if (PK) report((int)(R01 & 0xFFFFu), (int)(Lr & 0xFFFFu)); else report(s0, last);
          }
          if (LAST_TRY) report(t.len, t.len);   // the last try, at pos == len: the empty match
          fin = true;
        }
      }
#undef MRX_MW_BYTE
      __builtin_amdgcn_wave_barrier();
      if (__all(fin || wb + CH >= end)) break;
    }
#undef MRX_MW_LOAD
    if constexpr (TRIES) {
      if (!fin) ew2_apply(end64 + (((e >> 16) >> p.mw_cshift) << 3), t.len, false);
    } else
    if (PK) { if (!fin && ((e >> 10) & 1u)) report((int)(R01 & 0xFFFFu), (int)(Lr & 0xFFFFu)); }
    else
    if (!fin && ((e >> 10) & 1u)) report(s0, last);   // end of the text: the oldest walk has accepted
    if (LAST_TRY) {   // the last try, at pos == len (the empty text: its only one)
      if (!fin || (live && !skipped && t.len == 0)) report(t.len, t.len);
    }
    if (live && !skipped) {
      if (MODE == STEP_COUNT || MODE == STEP_SLOTS || MODE == STEP_ANY) counts[i] = k;
      if (MODE == STEP_SEARCH) { out_s[i] = rs; out_e[i] = re; }
    }
  }
}

// ---- where matches begin: the right-to-left pass (PF_BACKSET, DevPlan::off_bk_*; host side: build_backset()) ------
// One lane per text, the text through the LDS tile window by window FROM ITS END, one class lookup and one table
// lookup per byte, no registers but the set number: bit (position & 31) of word (position >> 5) of the text's row
// of Layout::bm says whether the reference's walk from that position succeeds.  cnt[i] = marks of text i.
__global__ __launch_bounds__(64 * kWsWaves) void k_backscan(DevPlan p, const uint8_t* __restrict__ blob, Layout lay,
                                                            int64_t n, uint32_t* __restrict__ bm,
                                                            int32_t* __restrict__ cnt) {
  constexpr int CH = 128, kRowPitch = CH + 16, LPR = CH / 16, RPI = 64 / LPR, NL = 64 / RPI;
  __shared__ __align__(16) uint8_t tiles[kWsWaves][64 * kRowPitch];
  extern __shared__ __align__(16) uint8_t lds[];
  // PRE (as in k_mwalk): cls2[byte] = 2 x class and an entry = LDS byte address of the next set's row | mark -- rows
  // begin on multiples of 2 << cshift bytes and the table on a multiple of 128, so (entry & ~1) | cls2[byte] is the
  // address of the byte's entry: one v_and_or instead of shift, shift, add
  typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
  typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
  const uint32_t tab_base = lds_base + 256u;
  const bool pre = p.bk_cshift <= 6 && lds_base + (uint32_t)p.bk_bytes <= 65536u && (tab_base & 127u) == 0;
  {
    const uint32_t* src = (const uint32_t*)(blob + p.off_bk_cls);
    uint32_t* dst = (uint32_t*)lds;
    const int sh = p.bk_cshift + 1;
    for (int e = threadIdx.x; e < ((p.bk_bytes + 3) >> 2); e += blockDim.x) {
      uint32_t w = src[e];
      if (pre) {
        if (e < 64) w = (w << 1) & 0xFEFEFEFEu;
        else {
          const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
          w = ((((lo >> 1) << sh) + tab_base) | (lo & 1u)) | (((((hi >> 1) << sh) + tab_base) | (hi & 1u)) << 16);
        }
      }
      dst[e] = w;
    }
  }
  __syncthreads();
  const uint8_t* clsT = lds;
  const uint16_t* tab = (const uint16_t*)(lds + 256);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* tile = tiles[wave];
  const int seg = lane % LPR, rsub = lane / LPR;
  const int64_t nw = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * kWsWaves + wave; w < nw; w += (int64_t)gridDim.x * kWsWaves) {
    const int64_t i = (w << 6) + lane;
    const bool live = i < n;
    const Text t = live ? lay.text(i) : Text(blob, 0);
    const uintptr_t addr = t.len > 0 ? (uintptr_t)t.ptr : (uintptr_t)blob;
    const int mis = t.len > 0 ? (int)(addr & 15) : 0;
    const uintptr_t rb = addr & ~(uintptr_t)15;
    const int end = mis + t.len;
    __builtin_amdgcn_wave_barrier();
    *(uint4*)(tile + lane * kRowPitch + CH) = make_uint4((uint32_t)rb, (uint32_t)((uint64_t)rb >> 32), (uint32_t)end, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int max_end = end;
    for (int off = 32; off > 0; off >>= 1) max_end = max(max_end, __shfl_xor(max_end, off));
    const uint8_t* myrow = tile + lane * kRowPitch;
    uint32_t* myout = bm + (live ? lay.bm_row(i) : 0);
    // st: !pre: first entry of the current set's row; pre: the last entry taken (row address | mark)
    uint32_t st = pre ? tab_base + (((uint32_t)p.bk_start << p.bk_cshift) << 1) : (uint32_t)p.bk_start << p.bk_cshift, word = 0;
    int marks = 0;
    uint4 v[NL];
#define MRX_BK_LOAD(CB)                                                                   \
    do {                                                                                  \
      _Pragma("unroll") for (int j_ = 0; j_ < NL; ++j_) {                                  \
        const uint4 rs_ = *(const uint4*)(tile + (RPI * j_ + rsub) * kRowPitch + CH);      \
        uint32_t fo_ = (uint32_t)(CB) + seg * 16;                                          \
        if (fo_ >= rs_.z) fo_ = 0;                                                         \
        v[j_] = mrx_ldg((const uint4*)((const uint8_t*)(((uint64_t)rs_.y << 32) | rs_.x) + fo_)); \
      }                                                                                    \
    } while (0)
    const int wb_last = max_end > 0 ? ((max_end - 1) / CH) * CH : -CH;
    if (wb_last >= 0) MRX_BK_LOAD(wb_last);
    for (int wb = wb_last; wb >= 0; wb -= CH) {
#pragma unroll
      for (int j = 0; j < NL; ++j) *(uint4*)(tile + (RPI * j + rsub) * kRowPitch + seg * 16) = v[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (wb - CH >= 0) MRX_BK_LOAD(wb - CH);   // the window in front, in flight while this one is stepped
      uint4 mw = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int g = CH / 16 - 1; g >= 0; --g) {
        const int f0 = wb + g * 16;
        const uint4 wv = *(const uint4*)(myrow + g * 16);
        const uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
        const bool all_full = pre && __all(live && f0 >= mis && f0 + 16 <= end);
        if (all_full) {   // every lane's text covers the group: no frame test per byte
#pragma unroll
          for (int q = 15; q >= 0; --q) {
            const uint32_t b = (words[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
            st = *(lds_cu16*)(uintptr_t)((st & 0xFFFEu) | *(lds_cu8*)(uintptr_t)(lds_base + b));
            word |= (st & 1u) << ((g & 1) * 16 + q);
          }
        } else {
#pragma unroll
          for (int q = 15; q >= 0; --q) {
            const int f = f0 + q;
            const bool act = live && f >= mis && f < end;
            const uint32_t b = (words[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
            uint32_t en;
            if (pre) en = *(lds_cu16*)(uintptr_t)((st & 0xFFFEu) | *(lds_cu8*)(uintptr_t)(lds_base + b));
            else en = tab[st + clsT[b]];
            if (act) {
              st = pre ? en : (uint32_t)(en >> 1) << p.bk_cshift;
              word |= (uint32_t)(en & 1u) << (f & 31);
            }
          }
        }
        if ((g & 1) == 0) {   // a word of marks is complete (the same step for every lane)
          marks += __popc(word);
          if (g == 6) mw.w = word; else if (g == 4) mw.z = word; else if (g == 2) mw.y = word; else mw.x = word;
          word = 0;
        }
      }
      // the window's four words in one 16-byte store (rows begin on multiples of four words): single words from 64
      // lanes to 64 rows left the rows' lines partly written for the length of the pass
      if (live && wb < end && wb + CH > mis) *(uint4*)(myout + (wb >> 5)) = mw;
      __builtin_amdgcn_wave_barrier();
    }
#undef MRX_BK_LOAD
    if (live) cnt[i] = marks;
  }
}

// ---- bitset NFA, first pass: the union automaton -------------------------------------------------
// PF_BSTEP plans.  The restart-per-position search of the reference (LazyDFA.match_next / match_all,
// pikevm.mojo:754-817) walks the bytes of a failing region once per start position.  What it can find
// is bounded by a single linear pass: run ALL starts at once -- the set U of positions alive in ANY walk
// begun at a candidate byte so far,
//     U' = follow8[ ((U & mask[b]) | (start & mask[b] if b may start a walk)) ]
// -- and note where U' holds MATCH: exactly the positions where SOME walk (from some start) is in a
// matching state, i.e. a superset of the ends of the matches the reference reports (it resumes behind
// each match, so walks begun inside one do not count for it; a superset is all that is needed).  So
//   * a text without such a position has no match at all, and
//   * behind the last such position nothing can match; a match reported by the restart-per-position
//     search ends at one of them, so truncating the text there changes no result.
// k_bscan makes that pass (one lane per text, text through the LDS tile in coalesced 128-byte rows, one
// table read for mask | start-mask and ceil(positions / 8) follow reads per byte, no restarts, lanes in
// lockstep) and writes for every text the length the second pass (k_wstep<., 0, 1>) has to look at:
// mode 0 (search) the whole text if it has a match end, else 0; mode 1 (count / findall) the last match
// end, 0 if none.  W32: at most 32 positions -- 32-bit sets.
// 32-bit sets: the follow tables are indexed by chunks of up to 11 positions instead of 8 (2048 entries each) --
// 11 positions need one lookup per byte instead of two, 22 two instead of three
__host__ __device__ inline int bscan_nch32(int npos) { return (npos + 10) / 11; }
__host__ __device__ inline int bscan_cb32(int npos) { const int c = bscan_nch32(npos); return c ? (npos + c - 1) / c : 8; }
__host__ __device__ inline size_t bscan_table_bytes(int npos) {
  return npos <= 32 ? (size_t)2048 + (size_t)bscan_nch32(npos) * ((size_t)4 << bscan_cb32(npos))
                    : (size_t)4096 + (size_t)((npos + 7) / 8) * 2048;
}
// DFA = 1: the same pass for a table plan of the stepper's plain route (PF_STEPPABLE, at most 32 states):
// U is the set of DFA STATES some walk is in.  A state's successor depends on the byte, so the follow
// tables are per byte class -- follow8[class][j][v] = { delta(8 j + k, class) : k in v } -- and a byte's
// entry holds the offset of its class's tables instead of a mask; a walk beginning on the byte adds
// delta(start state, byte).  MATCH = an accepting state.
__host__ __device__ inline size_t bscan_dfa_table_bytes(int nstates, int ncls) {
  return (size_t)2048 + (size_t)ncls * ((nstates + 7) / 8) * 1024;
}
// NCH: follow-table lookups per byte where the launch site knows them (no branch per lookup); CHB: bytes of a text per
// tile row -- 64 halves the tile, so that five workgroups instead of three share a CU's LDS (the pass waits on LDS
// round trips: a byte's set depends on the lookup of the byte before)
template <int W32, int DFA = 0, int NCH = 0, int CHB = 128>
__global__ __launch_bounds__(64 * kWsWaves) void k_bscan(DevPlan p, const uint8_t* __restrict__ blob, Layout lay,
                                                         int64_t n, int mode, int32_t* __restrict__ limit,
                                                         int32_t* __restrict__ out2 = nullptr,
                                                         const int64_t* __restrict__ prefix = nullptr,
                                                         int64_t span_cap = 0) {
  // Modes 2-4 (programs whose matches all have one length L, DevPlan::bs_fixed_len): the restart-per-position loop
  // takes a match end e iff e - L does not lie inside the match taken before, so this pass IS the findall --
  // 2: limit[i] = number of matches; 3: their spans at prefix[i] (limit[i] = count of mode 2: texts without a
  // match are skipped); 4: search, limit[i] / out2[i] = start / end of the first match, -1 without one; 5: as 2, and
  // the spans into the text's slot row (L >= 4: a row of len / 4 + 32 slots holds them all).
  constexpr int CH = CHB, kRowPitch = CH + 16, LPR = CH / 16, RPI = 64 / LPR, NL = 64 / RPI;
  static_assert(!DFA || W32, "the state-set form is 32 bits wide");
  using Set = typename std::conditional<W32 != 0, uint32_t, uint64_t>::type;
  struct Ent { Set mask, sm; };   // positions that consume the byte (DFA: offset of the class's follow tables); what a walk starting on it adds
  __shared__ __align__(16) uint8_t tiles[kWsWaves][64 * kRowPitch];
  extern __shared__ __align__(16) uint8_t lds[];
  Ent* tbl = (Ent*)lds;
  Set* fol = (Set*)(tbl + 256);
  const int nch = NCH ? NCH : DFA ? (p.nstates + 7) >> 3 : W32 ? bscan_nch32(p.bs_npos) : (p.bs_npos + 7) >> 3;
  const int cb = (W32 && !DFA) ? bscan_cb32(p.bs_npos) : 8;   // positions per follow-table chunk
  const uint32_t cmask = (1u << cb) - 1u;
  Set bmatch = (Set)p.bs_match[0];
  if (DFA) {
    const uint8_t* g_cls = blob + p.off_cls;
    const uint8_t* g_first = blob + p.off_first;
    const uint16_t* g_tr = (const uint16_t*)(blob + p.off_trans);
    const bool filt = (p.flags & PF_HAS_MATCHER) != 0;
    const int ns = p.nstates, ncls = p.ncls;
    for (int e = threadIdx.x; e < 256; e += blockDim.x) {
      Ent t;
      t.mask = (Set)(g_cls[e] * nch * 256);
      const uint32_t t0 = g_tr[g_cls[e]];   // from the start state
      t.sm = (Set)(((filt && !g_first[e]) || t0 == 0xFFFFu) ? 0u : 1u << (t0 & 0x7FFFu));
      tbl[e] = t;
    }
    for (int e = threadIdx.x; e < ncls * nch * 256; e += blockDim.x) {
      const int c = e / (nch * 256), j = (e >> 8) % nch, v = e & 255;
      uint32_t u = 0;
      for (int k = 0; k < 8; ++k) {
        const int q = 8 * j + k;
        if (((v >> k) & 1) && q < ns) {
          const uint32_t t = g_tr[q * ncls + c];
          if (t != 0xFFFFu) u |= 1u << (t & 0x7FFFu);
        }
      }
      fol[e] = (Set)u;
    }
    uint32_t acc = 0;   // accepting states: bit 15 of any transition into them (every non-start state is entered by one)
    for (int e = 0; e < ns * ncls; ++e) {
      const uint32_t t = g_tr[e];
      if (t != 0xFFFFu && (t & 0x8000u)) acc |= 1u << (t & 0x7FFFu);
    }
    bmatch = (Set)acc;
  } else
  {
    const uint8_t* g_bcls = blob + p.off_bs_cls;
    const uint64_t* g_mask = (const uint64_t*)(blob + p.off_bs_mask);
    const uint64_t* g_fol = (const uint64_t*)(blob + p.off_bs_follow);
    const uint8_t* g_first = blob + p.off_first;
    const bool filt = (p.flags & PF_HAS_MATCHER) != 0;
    for (int e = threadIdx.x; e < 256; e += blockDim.x) {
      const uint64_t m = g_mask[g_bcls[e]];
      Ent t;
      t.mask = (Set)m;
      t.sm = (Set)((!filt || g_first[e]) ? (p.bs_start[0] & m) : 0ull);
      tbl[e] = t;
    }
    for (int e = threadIdx.x; e < (nch << cb); e += blockDim.x) {
      const int j = e >> cb, v = e & (int)cmask;
      uint64_t u = 0;
      for (int k = 0; k < cb; ++k)
        if (((v >> k) & 1) && cb * j + k < p.bs_npos) u |= g_fol[cb * j + k];
      fol[e] = (Set)u;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* tile = tiles[wave];
  const int seg = lane % LPR, rsub = lane / LPR;
  const int64_t nw = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * kWsWaves + wave; w < nw; w += (int64_t)gridDim.x * kWsWaves) {
    const int64_t i = (w << 6) + lane;
    const bool live = i < n;
    Text t = live ? lay.text(i) : Text(blob, 0);
    if (mode == 3 && live && limit[i] == 0) t = Text(blob, 0);   // emit pass: nothing to find here
    if (mode == 3 && __all(t.len == 0)) continue;
    const uintptr_t addr = t.len > 0 ? (uintptr_t)t.ptr : (uintptr_t)blob;   // empty rows park on the blob
    const int mis = t.len > 0 ? (int)(addr & 15) : 0;
    const uintptr_t rb = addr & ~(uintptr_t)15;
    const int end = mis + t.len;   // frame coordinates: the text is [mis, end)
    const int L = p.bs_fixed_len;
    int taken_end = mis, nmatch = 0, first_s = -1;   // modes 2-4: end of the match taken last (frame), matches so far
    int64_t span_at = (mode == 3 && live && t.len > 0) ? prefix[i] : 0;
    int64_t span_room = span_cap;
    if (mode == 5) {   // spans into the text's slot row (out2 = the rows, Layout::slot_row), gathered behind the prefix sums
      int cap = 0;
      span_at = live ? lay.slot_row(i, &cap) : 0;
      span_room = span_at + cap;
    }
    __builtin_amdgcn_wave_barrier();
    *(uint4*)(tile + lane * kRowPitch + CH) = make_uint4((uint32_t)rb, (uint32_t)((uint64_t)rb >> 32), (uint32_t)end, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int max_end = end;
    for (int off = 32; off > 0; off >>= 1) max_end = max(max_end, __shfl_xor(max_end, off));
    const uint8_t* myrow = tile + lane * kRowPitch;
    uint4 v[NL];
#define MRX_BS_LOAD(CB)                                                                   \
    do {                                                                                  \
      _Pragma("unroll") for (int j_ = 0; j_ < NL; ++j_) {                                  \
        const uint4 rs_ = *(const uint4*)(tile + (RPI * j_ + rsub) * kRowPitch + CH);      \
        uint32_t fo_ = (uint32_t)(CB) + seg * 16;                                          \
        if (fo_ >= rs_.z) fo_ = 0;                                                         \
        v[j_] = mrx_ldg((const uint4*)((const uint8_t*)(((uint64_t)rs_.y << 32) | rs_.x) + fo_)); \
      }                                                                                    \
    } while (0)
    Set U = 0;
    int last_end = 0;       // frame position behind the last byte after which U held MATCH (0: none; > mis otherwise)
    bool found = false;
    if (max_end > 0) MRX_BS_LOAD(0);
    for (int wb = 0; wb < max_end; wb += CH) {
#pragma unroll
      for (int j = 0; j < NL; ++j) *(uint4*)(tile + (RPI * j + rsub) * kRowPitch + seg * 16) = v[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (wb + CH < max_end) MRX_BS_LOAD(wb + CH);   // next window, in flight while this one is stepped
#pragma unroll 2
      for (int g = 0; g < CH / 16; ++g) {
        const uint4 wv = *(const uint4*)(myrow + g * 16);
        const uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t hits = 0;
        // (a wavefront whose 64 texts all cover the whole group -- every group of a fixed-pitch batch of full rows --
        // steps without the per-byte frame test)
        const bool all_full = __all(wb + g * 16 >= mis && wb + g * 16 + 16 <= end);
#define MRX_BS_STEP16(FULL)                                                                                       \
        _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                                          \
          const uint32_t b = (words[k >> 2] >> ((k & 3) * 8)) & 0xFFu;                                            \
          const Ent e = tbl[b];                                                                                   \
          const Set x = DFA ? U : ((U & e.mask) | e.sm);                                                          \
          Set nx = DFA ? e.sm : (Set)0;                                                                           \
          if (DFA) {                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
              if (j < nch) nx |= fol[(int)e.mask + (j << 8) + (int)((x >> (8 * j)) & 0xFFu)];   /* (wave uniform) */ \
          } else if (W32) {                                                                                       \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                         \
              if (j < nch) nx |= fol[(j << cb) + (int)((x >> (cb * j)) & cmask)];   /* (wave uniform) */          \
          } else {                                                                                                \
            for (int j = 0; j < nch; ++j) nx |= fol[(j << 8) + (int)((x >> (8 * j)) & 0xFFu)];                    \
          }                                                                                                       \
          if (FULL) {                                                                                             \
            U = nx;                                                                                               \
            hits |= (nx & bmatch) ? (1u << k) : 0u;                                                               \
          } else {                                                                                                \
            const int f = wb + g * 16 + k;                                                                        \
            const bool inside = f >= mis && f < end;                                                              \
            U = inside ? nx : (Set)0;                                                                             \
            hits |= (inside && (nx & bmatch)) ? (1u << k) : 0u;                                                   \
          }                                                                                                       \
        }
        if (all_full) { MRX_BS_STEP16(true) } else { MRX_BS_STEP16(false) }
#undef MRX_BS_STEP16
        if (mode >= 2) {
          while (hits) {   // match ends of this group, left to right
            const int e = wb + g * 16 + __builtin_ctz(hits) + 1;
            hits &= hits - 1;
            if (e - L < taken_end) continue;   // begins inside the match taken before it
            if ((mode == 3 || mode == 5) && span_at + nmatch < span_room) *(int2*)(out2 + 2 * (span_at + nmatch)) = make_int2(e - L - mis, e - mis);
            if (mode == 4 && !found) first_s = e - L - mis;
            taken_end = e;
            ++nmatch;
            found = true;
          }
        } else
        if (hits) { last_end = wb + g * 16 + (32 - __builtin_clz(hits)); found = true; }
      }
      __builtin_amdgcn_wave_barrier();
      if ((mode == 0 || mode == 4) && __all(found || wb + CH >= end)) break;   // search: every text has its answer
    }
#undef MRX_BS_LOAD
    if (mode == 4) {
      if (live) { limit[i] = first_s; out2[i] = first_s >= 0 ? first_s + L : -1; }
    } else if (mode == 2 || mode == 5) {
      if (live) limit[i] = nmatch;
    } else if (mode < 2)
    if (live) limit[i] = mode == 0 ? (found ? t.len : 0) : (found ? last_end - mis : 0);
  }
}

// First occurrence of a literal of at most 32 bytes in every text (the MemchrPrefilter of
// HybridMatcher.match_next, matcher.mojo:784-796: find(literal, start), then the engine searches from
// there).  Shift-and: bit k of R = "the last k + 1 bytes are the literal's first k + 1"; one mask read
// and three register operations per byte, the text through the same LDS tile as k_bscan, a wavefront
// stops when each of its texts has its answer.  starts[i] = position of the occurrence, -1 = none.
// FULL: no early exit; pre[i] = {first occurrence or -1, last occurrence << 1 | "a newline in the text"} --
// what NFAEngine's '.*' fast paths look at (nfa.mojo:577-585, 403-430), once per text instead of once per lane.
// PIECES: a lane takes C bytes of a text (plus the lit_len - 1 before them) instead of a whole text -- few long
// texts would leave most lanes without work -- and the pieces' answers meet in acc[text] = {first, last, newline}
// (minimum / maximum / or; k_litscan_join writes pre[] from them).  vfirst[t] = first piece of text t, vfirst[n]
// = their number, read here so that the host does not have to wait for it.
__device__ __forceinline__ int64_t litscan_text_of(const int64_t* __restrict__ vfirst, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (vfirst[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}
template <bool FULL, bool PIECES = false>
__global__ __launch_bounds__(64 * kWsWaves) void k_litscan(const uint8_t* __restrict__ lit, int lit_len, const uint8_t* __restrict__ blob,
                                                           Layout lay, int64_t n, int32_t* __restrict__ starts,
                                                           int2* __restrict__ pre, const int64_t* __restrict__ vfirst = nullptr,
                                                           int C = 0, int4* __restrict__ acc = nullptr) {
  constexpr int CH = 128, kRowPitch = CH + 16, LPR = CH / 16, RPI = 64 / LPR, NL = 64 / RPI;
  __shared__ __align__(16) uint8_t tiles[kWsWaves][64 * kRowPitch];
  __shared__ uint32_t maskt[256];
  for (int b = threadIdx.x; b < 256; b += blockDim.x) {
    uint32_t m = 0;
    for (int k = 0; k < lit_len; ++k) if (lit[k] == b) m |= 1u << k;
    maskt[b] = m;
  }
  __syncthreads();
  const uint32_t full = 1u << (lit_len - 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* tile = tiles[wave];
  const int seg = lane % LPR, rsub = lane / LPR;
  const int64_t nitems = PIECES ? vfirst[n] : n;
  const int64_t nw = (nitems + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * kWsWaves + wave; w < nw; w += (int64_t)gridDim.x * kWsWaves) {
    const int64_t i = (w << 6) + lane;
    const bool live = i < nitems;
    Text t = live && !PIECES ? lay.text(i) : Text(blob, 0);
    int64_t owner = 0;
    int piece_at = 0;
    if (PIECES && live) {
      owner = litscan_text_of(vfirst, n, i);
      const int k = (int)(i - vfirst[owner]);
      const Text whole = lay.text(owner);
      piece_at = k ? k * C - (lit_len - 1) : 0;   // C >= lit_len
      const int64_t stop = (int64_t)(k + 1) * C;
      t = Text(whole.ptr + piece_at, (int)(stop < whole.len ? stop : whole.len) - piece_at);
    }
    const uintptr_t addr = t.len > 0 ? (uintptr_t)t.ptr : (uintptr_t)blob;
    const int mis = t.len > 0 ? (int)(addr & 15) : 0;
    const uintptr_t rb = addr & ~(uintptr_t)15;
    const int end = mis + t.len;
    __builtin_amdgcn_wave_barrier();
    *(uint4*)(tile + lane * kRowPitch + CH) = make_uint4((uint32_t)rb, (uint32_t)((uint64_t)rb >> 32), (uint32_t)end, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int max_end = end;
    for (int off = 32; off > 0; off >>= 1) max_end = max(max_end, __shfl_xor(max_end, off));
    const uint8_t* myrow = tile + lane * kRowPitch;
    uint4 v[NL];
#define MRX_LS_LOAD(CB)                                                                   \
    do {                                                                                  \
      _Pragma("unroll") for (int j_ = 0; j_ < NL; ++j_) {                                  \
        const uint4 rs_ = *(const uint4*)(tile + (RPI * j_ + rsub) * kRowPitch + CH);      \
        uint32_t fo_ = (uint32_t)(CB) + seg * 16;                                          \
        if (fo_ >= rs_.z) fo_ = 0;                                                         \
        v[j_] = mrx_ldg((const uint4*)((const uint8_t*)(((uint64_t)rs_.y << 32) | rs_.x) + fo_)); \
      }                                                                                    \
    } while (0)
    uint32_t R = 0;
    int found_at = -1;   // frame position of the first occurrence's last byte
    int last_at = -1;    // FULL: of the last occurrence's
    uint32_t nl = 0;     // FULL: a newline inside the text
    if (max_end > 0) MRX_LS_LOAD(0);
    for (int wb = 0; wb < max_end; wb += CH) {
#pragma unroll
      for (int j = 0; j < NL; ++j) *(uint4*)(tile + (RPI * j + rsub) * kRowPitch + seg * 16) = v[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (wb + CH < max_end) MRX_LS_LOAD(wb + CH);
#pragma unroll 2
      for (int g = 0; g < CH / 16; ++g) {
        const uint4 wv = *(const uint4*)(myrow + g * 16);
        const uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t hits = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const uint32_t b = (words[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
          R = ((R << 1) | 1u) & maskt[b];
          const int f = wb + g * 16 + k;
          const bool inside = f >= mis && f < end;
          R = inside ? R : 0u;
          hits |= (R & full) ? (1u << k) : 0u;
          if (FULL) nl |= (inside && b == 10u) ? 1u : 0u;
        }
        if (hits && found_at < 0) found_at = wb + g * 16 + __builtin_ctz(hits);
        if (FULL && hits) last_at = wb + g * 16 + (31 - __builtin_clz(hits));
      }
      __builtin_amdgcn_wave_barrier();
      if (!FULL && __all(found_at >= 0 || wb + CH >= end)) break;
    }
#undef MRX_LS_LOAD
    const int first_pos = found_at >= 0 ? found_at - mis - (lit_len - 1) : -1;
    if (PIECES) {
      if (live && first_pos >= 0) {
        atomicMin((unsigned int*)&acc[owner].x, (unsigned int)(first_pos + piece_at));
        atomicMax(&acc[owner].y, last_at - mis - (lit_len - 1) + piece_at);
      }
      if (live && nl) atomicOr(&acc[owner].z, 1);
      continue;
    }
    if (live && starts) starts[i] = first_pos;
    if (live && pre) {
      const int last_pos = last_at >= 0 ? last_at - mis - (lit_len - 1) : -1;
      pre[i] = make_int2(first_pos, FULL ? (int)(((uint32_t)last_pos << 1) | nl) : 0);
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_litscan_pieces(Layout lay, int64_t n, int C, int32_t* __restrict__ cnt,
                                                           int4* __restrict__ acc) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int len = lay.text(t).len;
    cnt[t] = len <= C ? 1 : (len + C - 1) / C;
    acc[t] = make_int4(-1, -1, 0, 0);
  }
}
__global__ __launch_bounds__(kBlock) void k_litscan_join(int64_t n, const int4* __restrict__ acc, int2* __restrict__ pre) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int4 a = acc[t];
    pre[t] = make_int2(a.x, (int)(((uint32_t)a.y << 1) | (uint32_t)a.z));
  }
}

// slot rows -> CSR spans: one lane per text, rows of at most kStepSlots spans
__global__ __launch_bounds__(kBlock) void k_slots_gather(int64_t n, const int32_t* __restrict__ counts,
                                                         const int64_t* __restrict__ prefix,
                                                         const int32_t* __restrict__ slots,
                                                         int32_t* __restrict__ spans, int64_t span_cap) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = counts[i];
    if (c > kStepSlots) continue;   // re-walked by k_wstep<STEP_EMIT>
    const int64_t w = prefix[i];
    for (int j = 0; j < c; ++j)
      if (w + j < span_cap) *(int2*)(spans + 2 * (w + j)) = *(const int2*)(slots + 2 * (i * kStepSlots + j));
  }
}


// The stepper's two routes for long texts: one WAVEFRONT per text instead of one lane per text.
// ROUTE 1 (required byte):
// In HybridMatcher._match_all_required_byte (matcher.mojo:864-898) the attempt made at a hit --
// back up over first-class bytes, DFAEngine.match_first from there, keep it if it ends past the
// hit -- depends on the text only; what the loop carries from one hit to the next is `pos` (the
// hits before it are skipped).  So the wavefront sweeps its text 1 KiB at a time, 16 bytes per
// lane: every lane finds the required bytes in its 16 bytes and evaluates its first one at or past
// `pos` (a failed attempt moves the lane to its next hit), then the earliest kept attempt of the
// wavefront is the next match, `pos` moves to its end, lanes whose attempt now lies before `pos`
// drop it, and so on until the block has no hit left.  Text bytes come from a three-block LDS
// window (previous, current, next); a back-up or walk that leaves it reads global memory.
// ROUTE 0 (DFAEngine.match_all / match_next on PF_STEPPABLE plans, as k_wstep<., 0>) has the same
// shape: the candidates are the bytes a walk may start on (first-class filter and a live first
// transition), the attempt is the anchored walk from the candidate itself, it is kept when it
// reaches an accepting state, and a failed attempt moves on to the lane's next candidate byte.
constexpr int kRqBlock = 1024, kRqWaves = 4;
static_assert(kRqBlock == 1 << 10, "byte_at() shifts by 10");
__host__ __device__ inline size_t reqwave_table_bytes(int nstates) { return (size_t)nstates * 512 + 256; }
// BIG = 1 (PF_STEP_BIG): automata too large for a byte-indexed table.  The table is then indexed by
// byte class -- tab[state * ncls + cls[byte]], states kept premultiplied by ncls -- which costs a
// second LDS read per step, independent of the state.
__host__ __device__ inline size_t reqwave_big_bytes(int nstates, int ncls) { return (size_t)nstates * ncls * 2 + 512; }

template <int MODE, int ROUTE, int BIG = 0>
__global__ __launch_bounds__(64 * kRqWaves) void k_req_wave(DevPlan p, const uint8_t* __restrict__ blob, Layout lay,
                                                            int64_t n, int32_t* __restrict__ counts,
                                                            const int64_t* __restrict__ prefix,
                                                            int32_t* __restrict__ spans, int64_t span_cap,
                                                            int32_t* __restrict__ out_s, int32_t* __restrict__ out_e) {
  __shared__ __align__(16) uint8_t wins[kRqWaves][3 * kRqBlock];
  __shared__ int32_t ress[ROUTE == 0 ? kRqWaves : 1][ROUTE == 0 ? kRqBlock : 1];   // ROUTE 0: end of the walk from each byte of the block
  extern __shared__ __align__(16) uint8_t lds[];
  uint16_t* tab = (uint16_t*)lds;                 // tab[q * 256 + byte] = next | ACC, or DEAD
  const int ns = p.nstates;
  // ROUTE 1: fc[byte] = byte is in the first element's class; ROUTE 0: a walk may start on byte
  uint8_t* fc = lds + (BIG ? (size_t)ns * p.ncls * 2 : (size_t)ns * 512);
  uint8_t* clsb = fc + 256;                       // BIG: byte -> class
  {
    const uint8_t* g_cls = blob + p.off_cls;
    const uint8_t* g_first = blob + p.off_first;
    const uint16_t* g_tr = (const uint16_t*)(blob + p.off_trans);
    if (BIG) {
      for (int e = threadIdx.x; e < ns * p.ncls; e += blockDim.x) {
        const uint32_t t = g_tr[e];
        tab[e] = (uint16_t)(t == 0xFFFFu ? kWsDead : (((t & 0x7FFFu) * p.ncls) | ((t & 0x8000u) ? kWsAcc : 0u)));
      }
      for (int b = threadIdx.x; b < 256; b += blockDim.x) clsb[b] = g_cls[b];
    } else
    for (int e = threadIdx.x; e < ns * 256; e += blockDim.x) {
      const uint32_t t = g_tr[(e >> 8) * p.ncls + g_cls[e & 255]];
      tab[e] = (uint16_t)(t == 0xFFFFu ? kWsDead : ((t & 0x7FFFu) | ((t & 0x8000u) ? kWsAcc : 0u)));
    }
    const bool filt = (p.flags & PF_HAS_MATCHER) != 0;
    for (int b = threadIdx.x; b < 256; b += blockDim.x)
      fc[b] = ROUTE == 1 ? g_first[b] : (uint8_t)!((filt && !g_first[b]) || g_tr[g_cls[b]] == 0xFFFFu);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* win = wins[wave];
  int32_t* res = ress[ROUTE == 0 ? wave : 0];
  const uint32_t req = (uint32_t)p.required_byte * 0x01010101u;
  for (int64_t i = (int64_t)blockIdx.x * kRqWaves + wave; i < n; i += (int64_t)gridDim.x * kRqWaves) {
    const Text t = lay.text(i);
    if (lay.split > 0 && t.len < lay.split) continue;                       // k_wstep's text
    int slot_cap = kStepSlots;
    int64_t slot0 = i * kStepSlots;
    if ((MODE == STEP_EMIT || MODE == STEP_SLOTS) && lay.wide_slots) slot0 = lay.slot_row(i, &slot_cap);
    if (MODE == STEP_EMIT && counts && counts[i] <= slot_cap) continue;     // its spans are in the slot row
    int k = 0, rs = -1, re = -1;
    if (t.len > 0) {
      const uintptr_t addr = (uintptr_t)t.ptr;
      const int mis = (int)(addr & 15);
      const uint8_t* frame = (const uint8_t*)(addr & ~(uintptr_t)15);   // frame position f is frame[f]
      const int end = mis + t.len;                                      // the text is frame [mis, end)
      const int nblk = (end + kRqBlock - 1) / kRqBlock;
      const int64_t wo = MODE == STEP_EMIT ? prefix[i] : 0;
      auto load_block = [&](int b) {   // my 16 bytes of block b (zeros past the last 16-byte block of the text)
        const int o = b * kRqBlock + 16 * lane;
        return o < end ? mrx_ldg((const uint4*)(frame + o)) : make_uint4(0, 0, 0, 0);
      };
      __builtin_amdgcn_wave_barrier();
      *(uint4*)(win + 16 * lane) = load_block(0);
      uint4 nxt = nblk > 1 ? load_block(1) : make_uint4(0, 0, 0, 0);
      int pos = mis;
      for (int b = 0; b < nblk; ++b) {
        if (b + 1 < nblk) *(uint4*)(win + ((b + 1) % 3) * kRqBlock + 16 * lane) = nxt;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (b + 2 < nblk) nxt = load_block(b + 2);   // in flight while this block is resolved
        // byte at frame position f: from the window when f lies in blocks b-1 .. b+1
        const int wlo = (b > 0 ? b - 1 : 0) * kRqBlock, whi = (b + 2 < nblk ? b + 2 : nblk) * kRqBlock;
        const int slot_lo = (wlo / kRqBlock) % 3;   // window slot of block wlo / kRqBlock; the next ones follow cyclically
        auto byte_at = [&](int f) -> uint32_t {
          const uint32_t d = (uint32_t)(f - wlo);
          if (d < (uint32_t)(whi - wlo)) {
            uint32_t sl = slot_lo + (d >> 10);
            sl = sl >= 3u ? sl - 3u : sl;
            return win[sl * kRqBlock + (d & (kRqBlock - 1))];
          }
          return frame[f];
        };
        // required bytes among my 16 bytes
        const int gpos = b * kRqBlock + 16 * lane;
        uint32_t hm = 0;
        {
          const uint4 v = *(const uint4*)(win + (b % 3) * kRqBlock + 16 * lane);
          const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t x = w4[j] ^ req;   // zero bytes = hits
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (ROUTE == 1) { if (((x >> (8 * c)) & 0xFFu) == 0u) hm |= 1u << (4 * j + c); }
              else if (fc[(w4[j] >> (8 * c)) & 0xFFu]) hm |= 1u << (4 * j + c);
            }
          }
          // keep frame positions in [mis, end)
          const int lo = mis - gpos, hi = end - gpos;
          if (lo > 0) hm &= lo >= 16 ? 0u : ~((1u << lo) - 1u);
          if (hi < 16) hm &= hi <= 0 ? 0u : ((1u << hi) - 1u);
        }
        // the anchored walk from frame position st: end of the longest match, or -1
        auto walk_from = [&](int st) {
          int q = st, state = 0, last = -1;
          while (q < end) {
            const uint32_t e = BIG ? tab[state + clsb[byte_at(q)]] : tab[(state << 8) + byte_at(q)];
            if (e == kWsDead) break;
            state = (int)(e & (BIG ? 0x7FFFu : 0x3FFFu));
            ++q;
            if (e & kWsAcc) last = q;
          }
          return last;
        };
        auto emit = [&](bool mine, int cs, int ce) {
          if (mine) {
            if (MODE == STEP_EMIT) {
              if (wo + k < span_cap) *(int2*)(spans + 2 * (wo + k)) = make_int2(cs - mis, ce - mis);
            }
            if (MODE == STEP_SLOTS) {
              if (k < slot_cap) *(int2*)(spans + 2 * (slot0 + k)) = make_int2(cs - mis, ce - mis);
            }
          }
        };
        const int rel0 = pos - gpos;   // candidates before pos are never visited
        if (rel0 > 0) hm &= rel0 >= 16 ? 0u : ~((1u << rel0) - 1u);
        if (ROUTE == 0) {
          // Candidates are dense (every byte a walk may start on): evaluate each of them once, all
          // lanes busy, and keep the ends; picking the matches is then a chain of mask lookups.  (With
          // the lazy scheme below a lane whose attempt is overtaken by a match re-walks its remaining
          // candidates one by one while the other 63 wait.)
          uint32_t vm = 0, todo = hm;   // vm: my candidates whose walk reached an accepting state
          while (__any(todo != 0u)) {
            if (todo != 0u) {
              const int j = __builtin_ctz(todo);
              todo &= todo - 1u;
              const int last = walk_from(gpos + j);
              if (last >= 0) { vm |= 1u << j; res[16 * lane + j] = last; }
            }
          }
          while (true) {
            const int rel = pos - gpos;
            uint32_t m = vm;
            if (rel > 0) m &= rel >= 16 ? 0u : ~((1u << rel) - 1u);
            const uint64_t wm = __ballot(m != 0u);
            if (wm == 0ull) break;
            const int winner = __builtin_ctzll(wm);
            const int j = m ? __builtin_ctz(m) : 0;
            const int ce = m ? res[16 * lane + j] : 0;
            emit(lane == winner, gpos + j, ce);
            pos = __builtin_amdgcn_readlane(ce, winner);
            ++k;
            if (MODE == STEP_SEARCH) { rs = __builtin_amdgcn_readlane(gpos + j, winner) - mis; re = pos - mis; break; }
          }
        } else {
        int ch = 0, cs = 0, ce = 0;   // my current attempt: hit, start, end (valid while `have`)
        bool have = false;
        while (true) {
          // hits before pos are skipped; an attempt whose hit fell before pos is void
          const int rel = pos - gpos;
          if (rel > 0) hm &= rel >= 16 ? 0u : ~((1u << rel) - 1u);
          if (have && ch < pos) have = false;
          // every lane with hits left gets a kept attempt (or runs out of hits)
          while (__any(!have && hm != 0u)) {
            if (!have && hm != 0u) {
              const int h = gpos + __builtin_ctz(hm);
              int st = h;
              while (st > mis && fc[byte_at(st - 1)]) --st;
              const int last = walk_from(st);
              if (last > h) { have = true; ch = h; cs = st; ce = last; }
              else hm &= hm - 1u;   // failed: on to my next hit
            }
          }
          // lanes hold consecutive 16-byte groups: the earliest kept attempt is the lowest lane's
          const uint64_t wm = __ballot(have);
          if (wm == 0ull) break;   // no hit left in this block
          const int winner = __builtin_ctzll(wm);
          emit(lane == winner, cs, ce);
          pos = __builtin_amdgcn_readlane(ce, winner);
          ++k;
          if (MODE == STEP_SEARCH) { rs = __builtin_amdgcn_readlane(cs, winner) - mis; re = pos - mis; break; }
        }
        }
        __builtin_amdgcn_wave_barrier();
        if (MODE == STEP_SEARCH && k) break;
      }
    }
    if (lane == 0 && (MODE == STEP_COUNT || MODE == STEP_SLOTS)) counts[i] = k;
    if (lane == 0 && MODE == STEP_SEARCH) { out_s[i] = rs; out_e[i] = re; }
  }
}

// the same for wide rows (Layout::wide_slots): one wavefront per text
__global__ __launch_bounds__(kBlock) void k_slots_gather_wide(Layout lay, int64_t n, const int32_t* __restrict__ counts,
                                                              const int64_t* __restrict__ prefix,
                                                              const int32_t* __restrict__ slots,
                                                              int32_t* __restrict__ spans, int64_t span_cap) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += nwaves) {
    int cap;
    const int64_t row = lay.slot_row(i, &cap);
    const int c = counts[i];
    if (c > cap) continue;   // re-walked by k_req_wave<STEP_EMIT>
    const int64_t w = prefix[i];
    for (int j = lane; j < c; j += 64)
      if (w + j < span_cap) *(int2*)(spans + 2 * (w + j)) = *(const int2*)(slots + 2 * (row + j));
  }
}


template <int MODE, int BT = 0>
__global__ __launch_bounds__(kBlock) void k_findall(DevPlan p, const uint8_t* __restrict__ blob,
                                                    Layout lay, int64_t n,
                                                    int32_t* __restrict__ counts,
                                                    const int64_t* __restrict__ prefix,
                                                    int32_t* __restrict__ spans, int64_t span_cap) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Ctx c = stage_tables(p, blob, lds);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const Text t = lay.text(i);
    if (MODE == FA_COUNT) {
      int k = 0;
      for_each_match<BT>(c, t, [&](int, int) { ++k; });
      counts[i] = k;
    } else {
      int64_t w = prefix[i];
      for_each_match<BT>(c, t, [&](int s, int e) {
        if (w < span_cap) *(int2*)(spans + 2 * w) = make_int2(s, e);
        ++w;
      });
    }
  }
}

// ---- streaming findall ----------------------------------------------------------
// Each wavefront owns 64 consecutive texts (fixed pitch).  Per CHUNK (128 bytes of
// every text, i.e. one cache line per text):
//   1. CHUNK/16 coalesced wave loads (16 B per lane; 8 lanes cover one text's line,
//      8 texts per instruction) bring 64 texts x CHUNK bytes into registers (issued one chunk ahead, so their
//      latency hides behind the walk of the previous chunk) and then into an LDS
//      tile with a (CHUNK+16)-byte row pitch: the per-lane 16-byte read-back is
//      bank conflict free;
//   2. every lane reads its own CHUNK bytes back (ds_read_b128) and steps the
//      <=4-state search automaton once per byte.  The transition for byte b is
//      (stcol[b] >> 4*state) & 15 with stcol[] a 256-entry u16 table in LDS: the
//      lookup depends only on the byte, never on the state, so the serial chain is
//      two register ops per byte and the lookups of a group issue back to back.
// Entry bits: next<<2 | EMIT<<1 | NEWSTART (built by build_stream_cols()).
// Events are not turned into spans here.  Per 16-byte group the two event bits of
// every byte are packed into one word F (bit 2k = NEWSTART, bit 2k+1 = EMIT of byte
// k); a group that holds an EMIT produces ONE 16-byte record {F, start carried into
// the group, byte position of the group, #matches of the text before it | lane}.
// The records of a wavefront go to ONE dense stream per wavefront: the lanes that
// have a record in a group are ranked with ballot/mbcnt and write consecutive
// 16-byte slots, so the stores are coalesced (per-lane scattered 16-byte stores
// were measured to cost 3x the whole rest of the kernel, tools/stream_ablate.hip).
// At most one record per text and group, so a stream of 64 * (len/16 + 2) slots
// cannot overflow.  k_decode turns records into CSR spans; records carry their
// text (lane) and rank, so their order in the stream does not matter.  This keeps
// the hot loop free of per-match work whose trip count would otherwise be the
// maximum over the 64 lanes.
constexpr int kStreamWaves = 4;
// Bytes of every text staged per step.  128 = one full cache line per text and step:
// with 64 the two halves of a line are requested a chunk-time apart and part of the
// lines are fetched from HBM twice (rocprof FETCH_SIZE).
#ifndef MRX_STREAM_CHUNK
#define MRX_STREAM_CHUNK 128
#endif
// Measurement only (tools/ablate.sh): switch pieces of the streaming loop off to see what
// bounds it.  Results are WRONG for any value but 0; never set in a product build.
//   1 no record stores   2 no global loads after a wavefront's first chunk
//   4 no column lookup   16 no event handling
// chunk of the class-table form (AUTO == 2): its dependent lookups are latency bound
#ifndef MRX_STREAM_CHUNK_TABLE
#define MRX_STREAM_CHUNK_TABLE MRX_STREAM_CHUNK
#endif
#ifndef MRX_ABLATE
#define MRX_ABLATE 0
#endif
struct EvRec {   // 16 bytes
  uint32_t F;
  int32_t start;     // start of the walk that is alive when the group begins
  int32_t pos_base;  // text position of the group's first byte
  uint32_t meta;     // (lane << 26) | matches of this text before this group
};
constexpr uint32_t kRecBeforeMask = (1u << 26) - 1u;
#ifndef MRX_FUSED_BATCH
#define MRX_FUSED_BATCH 8
#endif
constexpr int kFusedBatch = MRX_FUSED_BATCH;   // ST_FUSED: independent record loads in flight per lane
// REC32: fixed-pitch batches of texts up to kRec32MaxLen bytes.  Positions fit 16 bits, so one 16-byte
// record carries the event words of TWO adjacent groups (32 text bytes): {F of the even group, F of the
// odd group, start | (pos_base + 16) << 16, meta}, start / pos_base / matches-before taken at the even
// group.  Fewer records for the same events: 13 instead of 19 per KiB on the bench workload.
#ifndef MRX_REC32_MAX_LEN
#define MRX_REC32_MAX_LEN 65300
#endif
constexpr int kRec32MaxLen = MRX_REC32_MAX_LEN;
// position field of a REC32 record: text position of the even group + kRecPosBias.  The even group of a pair may lie in
// front of its text -- by up to 15 bytes in a 16-byte frame, up to 127 in the 128-byte frames of ragged batches
// (round 4) -- so the bias keeps the field positive: 16 (the pair's odd group holds text) + 128.
constexpr int kRecPosBias = 144;
static_assert(MRX_REC32_MAX_LEN + 16 + kRecPosBias < 65536, "position field is 16 bits");

__host__ __device__ inline int64_t rec_row_len(int64_t max_len) { return max_len / 16 + 2; }
// CSR batches: wavefront w's record region.  A text yields at most len/16 + 3 records (its frame
// has at most len/16 + 2 groups, plus the match that ends at len), so 64 texts starting at byte
// offset `first_off` need at most span/16 + 192 slots; regions floor(off/16) + 256 w apart cannot
// overlap.  Whole buffer: total_bytes/16 + 256 * waves + 256 slots.
__host__ __device__ inline int64_t rec_region_start(int64_t first_off, int64_t w) {
  return (first_off >> 4) + 256 * w;
}

// ST_SEARCH: first match only (regex.search).  ST_FIRST: regex.match_first -- the plan's anchored
// automaton (DevPlan::off_fa_*) run from byte 0, EMIT bit = "the state entered accepts"; a lane is
// finished when it enters the dead state, finished rows are no longer fetched.
// ST_FUSED: findall in ONE launch.  As ST_RECORDS, but a wavefront keeps the event records of its 64
// texts in a region of its own (reused task after task, so the lines stay in L2 / Infinity Cache
// instead of streaming out to HBM and back), and when it reaches the end of its texts it (1) publishes
// the number of matches of its 64 texts, (2) obtains the number of matches of all texts before them
// from the wavefronts that own those texts (decoupled look-back over one 8-byte {status, count}
// descriptor per 64 texts), and (3) expands its own records straight into their final CSR position --
// no record stream through HBM, no second and third launch.  64-text tasks are handed out in text
// order through a ticket counter: a wavefront only ever waits for tasks with lower numbers, and those
// are held by wavefronts that are already running.
// ST_ROWS: findall of batches full of matches (round 4).  The event words of every 32 bytes of a text go to a row of
// fixed pitch -- 8 bytes per 32 text bytes, position implied, nothing else -- instead of a 16-byte record per 32 bytes
// that hold a match end; k_decode_rows derives the rest.  Texts at a fixed 16-byte aligned pitch whose common length is
// a multiple of the chunk only (`recs` = the rows, `rec_row` = pairs per row).
enum { ST_RECORDS = 0, ST_COUNT = 1, ST_SEARCH = 2, ST_FIRST = 3, ST_FUSED = 4, ST_ROWS = 5 };

struct FusedArgs {
  // [0] ticket counter, [1] error word, [2 ..] one descriptor per 64-text task, then two words per
  // group of 64 tasks (span count + reports so far; running total at the group's start); zeroed per call
  unsigned long long* ctrl;
  int64_t* prefix;            // [n + 1] CSR offsets of the texts' spans (output)
  int32_t* spans;             // [span_cap][2] (output)
  int64_t span_cap;
  int64_t* total_out;         // number of spans of the batch
  int64_t rec_cap;            // records a wavefront's region holds (>= the most one task can produce)
  int32_t debug;              // measurement only (MRX_FUSED_DEBUG): 1 no record expansion, 2 no look-back, 4 no span stores
};
// A wavefront's own records -> spans at their final CSR position.  lane = record here, so the per-match
// work is spread evenly whatever the texts look like.  fused_fill_tile places the spans [tb, tb + tile)
// of the wavefront's range in an LDS tile (the text tile, free once the scan is over) -- that needs my_rel
// (spans of the wavefront's texts before each lane's) only, so it runs while the look-back is still in
// flight; fused_store_tile writes the tile out in coalesced 8-byte stores once `base` (spans of all texts
// before the wavefront's) is known.  DIRECT: a wavefront with more spans than three tiles hold writes
// them straight to memory instead of re-reading its records once per tile.
template <bool REC32, bool DIRECT>
__device__ __forceinline__ void fused_fill_tile(const EvRec* __restrict__ wave_recs, int total_recs, int my_rel, int tb,
                                                int64_t base, uint8_t* tile_bytes, int tile_nbytes,
                                                int32_t* __restrict__ spans, int64_t span_cap, int fixed_len, int lane) {
  using Slot = typename std::conditional<REC32, uint32_t, int2>::type;   // REC32: positions fit 16 bits
  Slot* tile = (Slot*)tile_bytes;
  const int tile_cap = tile_nbytes / (int)sizeof(Slot);
  for (int j = 0; j < total_recs; j += 64 * kFusedBatch) {
    uint4 rr[kFusedBatch];
#pragma unroll
    for (int u = 0; u < kFusedBatch; ++u) {
      const int o = j + u * 64 + lane;
      rr[u] = make_uint4(0, 0, 0, 0);
      if (o < total_recs) rr[u] = mrx_ldg((const uint4*)(wave_recs + o));   // L1 bypassed: my own stores, read from L2
    }
#pragma unroll
    for (int u = 0; u < kFusedBatch; ++u) {
      const uint4 r = rr[u];   // {F, start, pos_base, meta} / REC32: {F even, F odd, start | (pos + 16) << 16, meta}
      const int rel_t = __shfl(my_rel, (int)(r.w >> 26));
      int dst = rel_t + (int)(r.w & kRecBeforeMask) - tb;
      uint32_t Fw = r.x;
      int pb = REC32 ? (int)(r.z >> 16) - kRecPosBias : (int)r.z;
      int rstart = REC32 ? (int)(r.z & 0xFFFFu) : (int)r.y;
#pragma unroll
      for (int half = 0; half < (REC32 ? 2 : 1); ++half) {
        uint32_t em = Fw & 0xAAAAAAAAu;
        const uint32_t ns = Fw & 0x55555555u;
        while (em) {
          const int kk = __builtin_ctz(em) >> 1;              // byte of this EMIT
          const uint32_t nsb = ns & ((1u << (2 * kk)) - 1u);   // NEWSTARTs strictly before it
          int st = nsb ? pb + ((31 - __builtin_clz(nsb)) >> 1) : rstart;
          if (fixed_len > 0) st = pb + kk - fixed_len;
          if (DIRECT) {
            if (base + dst < span_cap) mrx_stg_span(spans + 2 * (base + dst), st, pb + kk);
          } else if (dst >= 0 && dst < tile_cap) {
            if constexpr (REC32) tile[dst] = ((uint32_t)st << 16) | (uint32_t)(pb + kk);
            else tile[dst] = make_int2(st, pb + kk);
          }
          ++dst;
          em &= em - 1;
        }
        if (REC32) {   // on to the odd group
          if (ns) rstart = pb + ((31 - __builtin_clz(ns)) >> 1);
          pb += 16;
          Fw = r.y;
        }
      }
    }
  }
}
template <bool REC32>
__device__ __forceinline__ void fused_store_tile(const uint8_t* tile_bytes, int cnt, int64_t dst0,
                                                 int32_t* __restrict__ spans, int64_t span_cap, int lane) {
  using Slot = typename std::conditional<REC32, uint32_t, int2>::type;
  const Slot* tile = (const Slot*)tile_bytes;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int k = lane; k < cnt; k += 64) {
    const int64_t dst = dst0 + k;
    if (dst < span_cap) {
      if constexpr (REC32) { const uint32_t v = tile[k]; mrx_stg_span(spans + 2 * dst, (int)(v >> 16), (int)(v & 0xFFFFu)); }
      else { const int2 v = tile[k]; mrx_stg_span(spans + 2 * dst, v.x, v.y); }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// Finish a scanned task: base from the look-back, CSR offsets of its 64 texts, its records -> spans.
template <bool REC32>
__device__ __forceinline__ void fused_finish(const FusedArgs& fz, const EvRec* __restrict__ wave_recs, int64_t w,
                                             int64_t nw, int64_t n, int my_cnt, int wrec, uint8_t* tile, int tile_bytes,
                                             int fixed_len, int lane) {
  // matches of my texts before this lane's, of the wavefront, and of the batch before the wavefront
  int incl = my_cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v_ = __shfl_up(incl, d);
    if (lane >= d) incl += v_;
  }
  const int my_rel = incl - my_cnt;
  const int total_spans = __shfl(incl, 63);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my record stores have reached L2
  __builtin_amdgcn_wave_barrier();
  const int tile_cap = tile_bytes / (REC32 ? 4 : 8);
  int64_t base;
  if (total_spans <= tile_cap) {
    // the usual case: all spans of the wavefront fit one tile -- expand the records while the
    // look-back is in flight, write the tile out when the base has arrived
    if (!(fz.debug & 1))
    fused_fill_tile<REC32, false>(wave_recs, wrec, my_rel, 0, 0, tile, tile_bytes, fz.spans, fz.span_cap, fixed_len, lane);
    base = (fz.debug & 2) ? 0 : fused_lookback(fz.ctrl, w, nw, lane);
    if (!(fz.debug & 4) && base >= 0)
    fused_store_tile<REC32>(tile, total_spans, base, fz.spans, fz.span_cap, lane);
  } else {
    base = fused_lookback(fz.ctrl, w, nw, lane);
    if (base < 0) {
    } else
    if (total_spans > 3 * tile_cap) {
      fused_fill_tile<REC32, true>(wave_recs, wrec, my_rel, 0, base, tile, tile_bytes, fz.spans, fz.span_cap, fixed_len, lane);
    } else {
      for (int tb = 0; tb < total_spans; tb += tile_cap) {
        fused_fill_tile<REC32, false>(wave_recs, wrec, my_rel, tb, 0, tile, tile_bytes, fz.spans, fz.span_cap, fixed_len, lane);
        fused_store_tile<REC32>(tile, total_spans - tb < tile_cap ? total_spans - tb : tile_cap, base + tb, fz.spans,
                                fz.span_cap, lane);
      }
    }
  }
  const int64_t my_text = (w << 6) + lane;
  if (base < 0) {   // a consumer of the CSR must not take it for complete (the host reports the error word as well)
    if (lane == 0) { fz.prefix[n] = -1; *fz.total_out = -1; }
    return;
  }
  if (my_text < n) fz.prefix[my_text] = base + my_rel;
  if (my_text == n - 1) { fz.prefix[n] = base + incl; *fz.total_out = base + incl; }
}

// AUTO = 1: byte-column automaton (<= 4 states, described above).
// AUTO = 4: the class-table automaton two bytes at a time (DevPlan::off_stg_pair): half as many
//           dependent lookups; only with a reset byte (no per-byte predicates) and never for pieces.
// AUTO = 2: class-table automaton for any streamable plan: cls[byte] (u8, LDS) is looked
//           up ahead for the whole group, then trans[state_row + cls] (u16, LDS) is a
//           dependent lookup per byte -- latency bound, hidden by the other wavefronts.
// CSR = 1: ragged batch (texts back to back, int64 offsets[n+1]) instead of a fixed pitch.  A text
//           is then walked in the frame of the 16-byte blocks that hold it: it starts `a` bytes
//           into its first block (a = address & 15); frame bytes before the text and after its
//           end are no-ops exactly like the bytes past the end of a short text.  Row base, a and
//           frame length of every text live in the 16 pad bytes behind its tile row.
// VIRT = 1 (with CSR = 1): the batch is a list of PIECES of long texts, cut at synchronising bytes of
// the search automaton (DevPlan::off_st_sync).  Piece v is the bytes [offsets[v], offsets[v] + vlen[v])
// of `data`, walked from the idle state; its first byte is a synchronising byte (or the first byte of
// its text), so from there on the walk is the one the whole text's walk takes.  The events of its
// first vskip[v] & 0x7FFFFFFF bytes belong to the piece before it and are dropped; bit 31 of vskip
// marks the last piece of a text (the only one that may end a match at the end of the text).
template <int MODE, int CH, int AUTO, int CSR, int VIRT = 0, int REC32 = 0>
#ifndef MRX_FUSED_WAVES
#define MRX_FUSED_WAVES 4
#endif
__global__ __launch_bounds__(64 * kStreamWaves, (MODE == ST_FUSED ? MRX_FUSED_WAVES : 1)) void k_stream_findall(
    DevPlan p, const uint8_t* __restrict__ blob, const uint8_t* __restrict__ data, int64_t stride,
    const int32_t* __restrict__ lens, int32_t common_len, const int64_t* __restrict__ offsets,
    int64_t n, int32_t* __restrict__ counts,
    int32_t* __restrict__ wave_nrecs, EvRec* __restrict__ recs, int64_t rec_row,
    int32_t* __restrict__ out_s, int32_t* __restrict__ out_e,
    const int32_t* __restrict__ vlen = nullptr, const uint32_t* __restrict__ vskip = nullptr,
    const FusedArgs* __restrict__ fzp = nullptr) {
  // (ST_FUSED's arguments sit in device memory -- written by k_fused_init, which also zeroes the ticket
  // word and the descriptors -- and are read where they are used, once per task: as kernel arguments
  // they would be live in scalar registers across the scan loop, which has none to spare)
#define fz (*fzp)
  constexpr bool RECS = MODE == ST_RECORDS || MODE == ST_FUSED;   // event records are produced
  constexpr bool ROWS = MODE == ST_ROWS;
  static_assert(!(MODE == ST_FUSED && VIRT), "pieces of long texts keep the three-launch form");
  static_assert(!ROWS || (!CSR && !VIRT && !REC32), "event rows: texts at a fixed aligned pitch");
  constexpr int kChunk = CH;
  constexpr int kRowPitch = CH + 16;      // +16: the per-lane 16-byte read-back is bank-conflict free
  constexpr int LPR = CH / 16;            // lanes that cover one text row in a load instruction
  constexpr int RPI = 64 / LPR;           // text rows per load instruction
  constexpr int NL = 64 / RPI;            // load instructions per chunk (= CH / 16)
  __shared__ __align__(16) uint8_t tiles[kStreamWaves][64 * kRowPitch];
  __shared__ __align__(16) uint16_t col_lds[256];
  __shared__ __align__(16) uint16_t col32_lds[AUTO == 5 ? 256 : 2];   // code columns (DevPlan::off_stcol32)
  static_assert(AUTO != 5 || MODE != ST_FIRST, "the anchored automaton has no code-column form");
  // pmask[x]: the first x bytes of a 16-byte group set.  Texts that end (or, in a frame, begin) inside
  // a group: when the automaton has a reset byte (DevPlan::st_reset_byte -- every state goes idle, no
  // walk starts, accepting states emit), the bytes outside the text are replaced by it and the group
  // takes the same branch-free steps as a full one; the match that runs to the end of the text is then
  // emitted by the first byte behind it, at the same position the end-of-text rule gives.
  __shared__ __align__(16) uint4 pmask[17];
  if (threadIdx.x < 17) {
    const int x = threadIdx.x;
    uint32_t w[4];
    for (int j = 0; j < 4; ++j) {
      const int nb = x - 4 * j;
      w[j] = nb <= 0 ? 0u : nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
    }
    pmask[x] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // (a piece that is not the last of its text must not end a match at its end: its walk simply runs on
  // into the bytes behind it -- they are the text's own -- and the events from there on are dropped)
  // (not in the count-only variants of the one-byte class table and the wide columns -- plans without a
  // pair table, rare: there the compiler hoists the mask lookups of all eight groups and ends up at 200+
  // VGPRs; they keep the per-byte predicates)
  // ST_FUSED: tasks are handed out in text order by a ticket counter, four at a time: ONE atomic per
  // workgroup and round (asked for wavefront by wavefront, the tickets of a grid queue up at the counter
  // word for tens of microseconds, and a wavefront's loads cannot complete past its own pending atomic).
  // Wavefront 0 asks one round ahead and hands the answer over through an LDS ring {round + 1, first task}.
  __shared__ unsigned long long blk_ticket[4];
  if (MODE == ST_FUSED && threadIdx.x == 0) {
    blk_ticket[1] = blk_ticket[2] = blk_ticket[3] = 0ull;
    blk_ticket[0] = (1ull << 32) | (uint32_t)__hip_atomic_fetch_add(fz.ctrl, (unsigned long long)kStreamWaves, __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT);
  }
  const bool use_fill = AUTO == 5 || (MODE != ST_FIRST && p.st_reset_byte >= 0 && !(MODE == ST_COUNT && (AUTO == 2 || AUTO == 3)));
  const uint32_t fillw = (uint32_t)(p.st_reset_byte & 0xFF) * 0x01010101u;
  extern __shared__ __align__(16) uint8_t stg_lds[];  // AUTO == 2: cls | trans | accept
  if (AUTO == 5) {
    const uint16_t* src = (const uint16_t*)(blob + p.off_stcol32);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) col32_lds[i] = src[i];
  } else if (AUTO == 1) {
    const uint16_t* src = (const uint16_t*)(blob + (MODE == ST_FIRST ? p.off_fa_col : p.off_stcol));
    for (int i = threadIdx.x; i < 256; i += blockDim.x) col_lds[i] = src[i];
  } else if (AUTO == 3) {
    const uint32_t* src = (const uint32_t*)(blob + (MODE == ST_FIRST ? p.off_fa_col : p.off_stcol));
    uint32_t* dst = (uint32_t*)stg_lds;
    for (int i = threadIdx.x; i < 512; i += blockDim.x) dst[i] = src[i];
  } else {
    const uint32_t* src = (const uint32_t*)(blob + (MODE == ST_FIRST ? p.off_fa_cls : p.off_stg_cls));
    uint32_t* dst = (uint32_t*)stg_lds;
    const int words = (MODE == ST_FIRST ? p.fa_bytes : p.stg_bytes) >> 2;
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const uint64_t* col64_lds = (const uint64_t*)stg_lds;   // AUTO == 3
  const uint8_t* cls_lds = stg_lds;
  const uint16_t* tr_lds = (const uint16_t*)(stg_lds + (MODE == ST_FIRST ? p.off_fa_trans - p.off_fa_cls
                                                                        : p.off_stg_trans - p.off_stg_cls));
  const uint8_t* acc_lds = stg_lds + (p.off_stg_acc - p.off_stg_cls);
  const uint32_t* pair_lds = (const uint32_t*)(stg_lds + (p.off_stg_pair - p.off_stg_cls));   // AUTO == 4
  // q4 value of the anchored automaton's dead state: its row offset (class table) or field shift (columns)
  const uint32_t fa_dead = AUTO == 2 ? (uint32_t)p.fa_nstates << p.fa_cshift
                                     : (uint32_t)p.fa_nstates * (AUTO == 1 ? 4u : 8u);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint8_t* tile = tiles[wave];
  const int64_t nwaves_total = (n + 63) >> 6;
  const uint32_t accmask = p.st_accept_mask;
  const int seg = lane % LPR;
  const int rsub = lane / LPR;

  // ST_FUSED: tasks come from the ticket counter, one 64-text task per ticket, always asked for one
  // task ahead.  (Several consecutive tasks per ticket would serialise the launch: a ticket's first
  // task could only learn its base after the previous ticket's owner had scanned ALL its tasks.)
  // The look-back and the expansion of a task's records run one task LATE, after the scan of the
  // wavefront's next task: by then the wavefronts in front have long published their counts, so
  // the differences in speed between wavefronts (every task has to wait for ALL tasks before it) are
  // absorbed instead of stalling the fast ones.  Two record regions per wavefront, used alternately.
  unsigned long long tk_next = 0ull;   // wavefront 0: the atomic whose answer is the next round's first task
  bool tk_asked = false;
  uint32_t round = 0;
  bool pend = false;       // a scanned task whose spans are not written yet
  int64_t pend_w = 0;
  int pend_cnt = 0, pend_wrec = 0, pend_half = 0, half = 0;
  for (int64_t w = (int64_t)blockIdx.x * kStreamWaves + wave;; w += (int64_t)gridDim.x * kStreamWaves) {
    if (MODE == ST_FUSED) {
      unsigned long long tv;
      while (true) {
        tv = *(volatile unsigned long long*)&blk_ticket[round & 3u];
        if ((uint32_t)(tv >> 32) == round + 1u) break;
        __builtin_amdgcn_s_sleep(1);
      }
      const int64_t first = (int64_t)__builtin_amdgcn_readfirstlane((uint32_t)tv);
      w = first + wave;
      tk_asked = wave == 0 && first < nwaves_total;
      if (tk_asked) {   // the next round's tasks: the answer is back long before the scan below is over
        if (lane == 0)
          tk_next = __hip_atomic_fetch_add(fz.ctrl, (unsigned long long)kStreamWaves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::: "memory");
      } else if (wave == 0 && lane == 0) {   // nothing left: tell the others
        blk_ticket[(round + 1u) & 3u] = ((unsigned long long)(round + 2u) << 32) | (uint32_t)nwaves_total;
      }
      ++round;
    }
    if (w >= nwaves_total) break;
    const int64_t base_text = w << 6;
    const int64_t my_text = base_text + lane;
    const bool live = my_text < n;
    int my_len, mis = 0;   // mis: bytes of my first 16-byte block that precede the text (CSR)
    if (CSR) {
      // offsets == nullptr: fixed pitch that is not 16-byte aligned (or too wide for 32-bit row
      // offsets) -- same frame treatment, text i at data + i * stride
      int64_t o0;
      if (offsets) {
        o0 = live ? offsets[my_text] : 0;
        my_len = live ? (VIRT ? vlen[my_text] : (int)(offsets[my_text + 1] - o0)) : 0;
      } else {
        o0 = my_text * stride;
        my_len = live ? (lens ? lens[my_text] : common_len) : 0;
      }
      // an empty text owns no block: park its row on the plan blob (valid memory, never read as text)
      const uintptr_t addr = my_len > 0 ? (uintptr_t)(data + o0) : (uintptr_t)blob;
      // The frame begins at the 128-byte LINE that holds the text's first byte (round 4; the 16-byte block before):
      // a chunk of a row is then one whole line, where a row at an arbitrary offset had every line of its text loaded
      // by two consecutive chunks -- the non-temporal loads keep nothing, so the kernel moved twice the batch (ragged
      // count: FETCH_SIZE 601 MB raw for 570 MB of text against 537 MB raw for 1074 MB at a fixed pitch; profiles/
      // r04_ragged.md).  Up to 112 more masked bytes per text; never leaves the page of the text's first byte.
      // (match_first keeps the 16-byte frame: its probe reads the text's first block only.)
      constexpr uintptr_t kFrameMask = MODE == ST_FIRST ? 15 : 127;
      mis = my_len > 0 ? (int)(addr & kFrameMask) : 0;
      const uintptr_t rb = addr & ~kFrameMask;
      *(uint4*)(tile + lane * kRowPitch + CH) =
          make_uint4((uint32_t)rb, (uint32_t)((uint64_t)rb >> 32), (uint32_t)(mis + my_len), 0u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      my_len = live ? (lens ? lens[my_text] : common_len) : 0;
    }
    const uint32_t vsk = (VIRT && live) ? vskip[my_text] : 0u;
    const int skip = (int)(vsk & 0x7FFFFFFFu);   // VIRT: events before this position are the previous piece's
    const int flen = mis + my_len;  // length of my text's frame
    int max_len = flen;  // longest frame in this wavefront decides the trip count
    for (int off = 32; off > 0; off >>= 1) max_len = max(max_len, __shfl_xor(max_len, off));

    // rows this lane stages: texts RPI*j + lane/LPR of the wavefront.  Addresses are a
    // wave-uniform base plus a 32-bit lane offset (rows past the batch end are clamped).
    const uint8_t* wbase = data + base_text * stride;
    const int64_t rows_here = n - base_text < 64 ? n - base_text : 64;
    uint32_t roff[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int r = RPI * j + rsub;
      if (r >= rows_here) r = (int)rows_here - 1;
      roff[j] = (uint32_t)(r * stride) + seg * 16;
    }
    // The pitch is a multiple of 16, so a 16-byte load that starts inside a row stays
    // inside it; past the row end the first bytes are read instead (and ignored).
#define MRX_LOAD_CHUNK(CB)                                                       \
    do {                                                                         \
      uint32_t cb_ = (uint32_t)(CB);                                             \
      if (!CSR && (int64_t)cb_ + seg * 16 >= stride) cb_ = (uint32_t)0 - (uint32_t)(seg * 16); \
      _Pragma("unroll") for (int j_ = 0; j_ < NL; ++j_)                          \
        if (MODE != ST_FIRST || !((skip_rows >> (RPI * j_ + rsub)) & 1ull)) {     \
          if (CSR) {                                                             \
            const uint4 rs_ = *(const uint4*)(tile + (RPI * j_ + rsub) * kRowPitch + CH); \
            uint32_t fo_ = cb_ + seg * 16;                                       \
            if (fo_ >= rs_.z) fo_ = 0;  /* past the frame: re-read its first block */ \
            v[j_] = MRX_LDG((const uint4*)((const uint8_t*)(((uint64_t)rs_.y << 32) | rs_.x) + fo_)); \
          } else {                                                               \
            v[j_] = MRX_LDG((const uint4*)(wbase + (roff[j_] + cb_)));                   \
          }                                                                      \
        }                                                                        \
    } while (0)
    uint64_t skip_rows = 0;  // ST_FIRST: rows (= lanes) whose walk has ended

    uint32_t q4 = 0;  // 4 * state (AUTO == 5: bit offset of the state's field, low two bits = its code)
    uint32_t q_codes = 0;   // AUTO == 5: the code word of the previous group (its top field = the state before this group)
    int start = 0;
    int cnt = 0;
    int wrec = 0;  // records written by this wavefront so far (wave uniform)
    bool done = !live;          // ST_SEARCH / ST_FIRST: this lane has its answer
    int res_s = -1, res_e = (MODE == ST_FIRST && live && p.fa_start_acc) ? 0 : -1;
    EvRec* wave_recs = (MODE == ST_RECORDS)
        ? recs + ((CSR && offsets) ? rec_region_start(offsets[base_text], w) : base_text * rec_row)
        : (MODE == ST_FUSED) ? recs + (((int64_t)blockIdx.x * kStreamWaves + wave) * 2 + half) * fz.rec_cap : nullptr;

    if (MODE == ST_FIRST) {
      // Probe: most anchored walks end within a few bytes.  Every lane reads the first 16 bytes
      // of its own text (one load instruction per wavefront instead of a full 128-byte chunk per
      // text); when that settles all 64 texts the wavefront is done.
      uint4 pv = make_uint4(0, 0, 0, 0);
      if (live && my_len > 0) {
        if (CSR) pv = *(const uint4*)(((uintptr_t)(data + (offsets ? offsets[my_text] : my_text * stride))) & ~(uintptr_t)15);
        else pv = *(const uint4*)(data + my_text * stride);
      }
      const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
      uint32_t pq = 0, pF = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uint32_t b = (pw[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
        const bool outside = k < mis || k >= flen;
        uint32_t e;
        if (AUTO == 2) {
          e = tr_lds[pq + cls_lds[b]];
          if (outside) e = pq << 2;
          pq = e >> 2;
        } else if (AUTO == 1) {
          e = (uint32_t)col_lds[b] >> pq;
          if (outside) e = pq;
          pq = e & 0xCu;
        } else {
          e = (uint32_t)(col64_lds[b] >> pq);
          if (outside) e = pq;
          pq = e & 0x38u;
        }
        pF = __builtin_amdgcn_alignbit(e, pF, 2);
      }
      const uint32_t pem = pF & 0xAAAAAAAAu;
      if (__all(!live || pq == fa_dead || flen <= 16)) {
        if (live) {
          int e_ = pem ? ((31 - __builtin_clz(pem)) >> 1) + 1 - mis : (p.fa_start_acc ? 0 : -1);
          if (AUTO == 2 && p.off_fa_end >= 0 && stg_lds[(p.off_fa_end - p.off_fa_cls) + (pq >> p.fa_cshift)]) e_ = my_len;
          out_s[my_text] = e_ >= 0 ? 0 : -1;
          out_e[my_text] = e_;
        }
        continue;
      }
    }
    uint4 v[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) v[j] = make_uint4(0, 0, 0, 0);
    if (max_len > 0) MRX_LOAD_CHUNK(0);
    uint8_t* wr = tile + rsub * kRowPitch + seg * 16;
    uint4 rowacc[ROWS ? 8 : 1];   // ST_ROWS: the event words of the last four chunks
    for (int cbase = 0; cbase < max_len; cbase += kChunk) {
#pragma unroll
      for (int j = 0; j < NL; ++j) *(uint4*)(wr + j * RPI * kRowPitch) = v[j];
      // the tile is private to this wavefront: wave-level ordering is enough
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (MODE == ST_SEARCH) { if (__all(done)) break; }
      if (MODE == ST_FIRST) {
        skip_rows = __ballot(done || cbase + kChunk >= flen);
        if (__all(done)) break;
      }
      if (MODE == ST_FUSED && tk_asked) {   // (wave uniform) the first chunk is here, so is the older atomic's answer
        const uint32_t nf = __builtin_amdgcn_readfirstlane((uint32_t)tk_next);
        if (lane == 0) blk_ticket[round & 3u] = ((unsigned long long)(round + 1u) << 32) | nf;
        tk_asked = false;
      }
      if (!(MRX_ABLATE & 2))
      if (cbase + kChunk < max_len) MRX_LOAD_CHUNK(cbase + kChunk);  // prefetch next chunk

      const int lim = flen - cbase;    // frame bytes [lo, lim) of this chunk are text (lim may be <= 0 or > CH)
      const int lo = mis - cbase;      // > 0 only in a misaligned text's first chunk (CSR)
      const bool all_inside = __all(lim >= kChunk && lo <= 0);
      const bool full = all_inside || use_fill;   // branch-free steps: no byte needs a predicate
      uint32_t F_even = 0, meta_even = 0, sp_even = 0;   // REC32: the even group of the current pair
      uint2 rowbuf[ROWS ? kChunk / 32 : 1];               // ST_ROWS: the chunk's event words
#pragma unroll
      for (int g = 0; g < kChunk / 16; ++g) {
        const uint4 wv = *(const uint4*)(tile + lane * kRowPitch + g * 16);
        uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
        if (use_fill && !all_inside) {   // wave uniform
          const int limf = (VIRT && !(vsk >> 31)) ? 0x3FFFFFFF : lim;   // no fill behind a piece that is not the last
          const int a = min(max(lo - g * 16, 0), 16), b = min(max(limf - g * 16, 0), 16);   // inside: [a, b)
          const uint4 pa = pmask[a], pb = pmask[b > a ? b : a];
          const uint32_t m[4] = {pb.x & ~pa.x, pb.y & ~pa.y, pb.z & ~pa.z, pb.w & ~pa.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) words[j] = (words[j] & m[j]) | (fillw & ~m[j]);
        }
        uint32_t F = 0;
        if (AUTO == 4) {
          uint32_t cc[16], pi[8];
#pragma unroll
          for (int k = 0; k < 16; ++k)
            cc[k] = cls_lds[(words[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
          for (int k = 0; k < 8; ++k) pi[k] = (cc[2 * k] << p.st_cshift) | cc[2 * k + 1];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t e = pair_lds[q4 + pi[k]];            // q4 = row offset of the state in the pair table
            q4 = e >> 4;
            F = __builtin_amdgcn_alignbit(e, F, 4);             // the flags of both bytes
          }
        } else if (AUTO == 2) {
          uint32_t cc[16];
#pragma unroll
          for (int k = 0; k < 16; ++k)
            cc[k] = cls_lds[(words[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            uint32_t e = tr_lds[q4 + cc[k]];                    // q4 = row offset of the state
            if (!full) { if (g * 16 + k >= lim || g * 16 + k < lo) e = q4 << 2; }   // outside the text: no-op
            q4 = e >> 2;
            F = __builtin_amdgcn_alignbit(e, F, 2);
          }
        } else if (AUTO == 3) {
          // wide byte columns: q4 is the bit offset of the state's 8-bit field
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint64_t cw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
              cw[k] = col64_lds[(words[(8 * h + k) >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              uint32_t e = (uint32_t)(cw[k] >> q4);
              if (!full) { if (g * 16 + 8 * h + k >= lim || g * 16 + 8 * h + k < lo) e = q4; }
              q4 = e & 0x38u;
              F = __builtin_amdgcn_alignbit(e, F, 2);
            }
          }
        } else if (AUTO == 5) {
          // code columns: one shift per byte (the count is taken modulo 32, so the fields above the state's
          // own need no masking), the state's 2-bit code recorded per byte, events from two code words
          uint32_t cv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k)
            cv[k] = col32_lds[(words[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            q4 = cv[k] >> (q4 & 31u);
            F = __builtin_amdgcn_alignbit(q4, F, 2);
          }
          const uint32_t qn = F, qp = __builtin_amdgcn_alignbit(qn, q_codes, 30);   // codes after / before each byte
          q_codes = qn;
          F = (qp & ~qn & 0xAAAAAAAAu) | (qn & ~qp & 0x55555555u);   // EMIT: accepting -> not; NEWSTART: not first -> first
        } else if (full) {
          uint32_t cv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k)
            cv[k] = (MRX_ABLATE & 4) ? ((words[k >> 2] >> ((k & 3) * 8)) & 0xFFu) * 0x0101u
                                     : col_lds[(words[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const uint32_t e = cv[k] >> q4;
            q4 = e & 0xCu;
            F = __builtin_amdgcn_alignbit(e, F, 2);  // F = (F >> 2) | (e << 30)
          }
        } else {
          // a text of this wavefront ends inside the chunk: bytes past the end are no-ops
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const uint32_t b = (words[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            uint32_t e = col_lds[b] >> q4;
            if (g * 16 + k >= lim || g * 16 + k < lo) e = q4;  // keep the state, no event bits
            q4 = e & 0xCu;
            F = __builtin_amdgcn_alignbit(e, F, 2);
          }
        }
        uint32_t em = F & 0xAAAAAAAAu;
        const uint32_t ns = F & 0x55555555u;
        const int gbase = cbase + g * 16 - mis;  // text position of the group's first byte
        if (VIRT) {
          const int dsk = skip - gbase;
          if (dsk > 0) { em &= dsk >= 16 ? 0u : ~((1u << (2 * dsk)) - 1u); F = ns | em; }
          if (use_fill && !(vsk >> 31)) {   // the events at and behind my piece's end are the next piece's
            const int de = my_len - gbase;
            if (de < 16) { em &= de <= 0 ? 0u : ((1u << (2 * de)) - 1u); F = ns | em; }
          }
        }
        if (MRX_ABLATE & 16) { cnt += (F == 0x12345u); continue; }
        if (ROWS) {
          if ((g & 1) == 0) F_even = F;
          else rowbuf[g >> 1] = make_uint2(F_even, F);
        } else
        if (RECS && REC32) {
          if ((g & 1) == 0) {
            F_even = F;
            sp_even = (uint32_t)start | ((uint32_t)(gbase + kRecPosBias) << 16);
            meta_even = ((uint32_t)lane << 26) | ((uint32_t)cnt & kRecBeforeMask);
          } else {
            const bool any = ((F_even | F) & 0xAAAAAAAAu) != 0u;
            const uint64_t has = __ballot(any);
            if (has) {  // wave uniform
              if (any) {
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
                *(uint4*)(wave_recs + wrec + rank) = make_uint4(F_even, F, sp_even, meta_even);
              }
              wrec += __builtin_popcountll(has);
            }
          }
        } else if (RECS) {
          const uint64_t has = __ballot(em != 0);
          if (has) {  // wave uniform
            if (em) {
              const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
              EvRec r;
              r.F = F; r.start = start; r.pos_base = gbase;
              r.meta = ((uint32_t)lane << 26) | ((uint32_t)cnt & kRecBeforeMask);
              if (!(MRX_ABLATE & 1)) wave_recs[wrec + rank] = r;
              else if (r.F == 0x12345u && r.start == -77) wave_recs[0] = r;  // keep r alive
            }
            wrec += __builtin_popcountll(has);
          }
        }
        if (MODE == ST_FIRST) {
          if (em) res_e = gbase + ((31 - __builtin_clz(em)) >> 1) + 1;  // last accepting position so far
          if (q4 == fa_dead) done = true;
        }
        if (MODE == ST_SEARCH) {
          if (!done && em) {  // leftmost match = first EMIT of the text
            const int kk = __builtin_ctz(em) >> 1;
            const uint32_t nsb = ns & ((1u << (2 * kk)) - 1u);
            res_s = nsb ? gbase + ((31 - __builtin_clz(nsb)) >> 1) : start;
            res_e = gbase + kk;
            if (p.st_fixed_len > 0) res_s = res_e - p.st_fixed_len;   // exact-literal automaton
            done = true;
          }
        }
        cnt += __builtin_popcount(em);
        if (ns) start = gbase + ((31 - __builtin_clz(ns)) >> 1);
      }
      if constexpr (ROWS) {
        // 32 bytes of event words per lane and chunk.  Written chunk by chunk, a row's 128-byte line is touched four times
        // some microseconds apart and the lines in flight (64 per wavefront) are as many as L2 holds: WRITE_SIZE read
        // 7.2 GB for 4.3 GB of rows.  So four chunks are kept in registers and a lane writes a whole line at a time.
        static_assert(kChunk == 128, "event rows: four pairs per chunk");
#pragma unroll
        for (int q = 0; q < 6; ++q) rowacc[q] = rowacc[q + 2];
        rowacc[6] = make_uint4(rowbuf[0].x, rowbuf[0].y, rowbuf[1].x, rowbuf[1].y);
        rowacc[7] = make_uint4(rowbuf[2].x, rowbuf[2].y, rowbuf[3].x, rowbuf[3].y);
        const int ci = cbase >> 7;
        if ((ci & 3) == 3 || cbase + kChunk >= max_len) {   // (wave uniform: one common length)
          const int nc = (ci & 3) + 1;   // the accumulator's last nc chunks are new
          if (live) {
            uint4* const dst = (uint4*)((uint2*)recs + my_text * rec_row) + 2 * (ci - (nc - 1));
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (q >= 8 - 2 * nc) dst[q - (8 - 2 * nc)] = rowacc[q];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (MODE == ST_FUSED && tk_asked) {   // a task without a single chunk (64 empty texts)
      const uint32_t nf = __builtin_amdgcn_readfirstlane((uint32_t)tk_next);
      if (lane == 0) blk_ticket[round & 3u] = ((unsigned long long)(round + 1u) << 32) | nf;
      tk_asked = false;
    }
    // end of text: a walk that is in an accepting state ends at len
    {
      const bool tail = MODE != ST_FIRST && live && (!VIRT || (vsk >> 31)) &&
                        (AUTO == 5 ? ((p.st_acc32 >> (q4 & 31u)) & 1u) != 0
                         : AUTO == 4 ? acc_lds[q4 >> (2 * p.st_cshift)] != 0
                         : AUTO == 2 ? acc_lds[q4 >> p.st_cshift] != 0
                                   : ((accmask >> (q4 >> (AUTO == 3 ? 3 : 2))) & 1u) != 0);
      if (RECS) {
        const uint64_t has = __ballot(tail);
        if (tail) {
          const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
          EvRec r;
          r.F = 2u; r.start = start; r.pos_base = my_len;  // EMIT at byte 0 of a group placed at len
          r.meta = ((uint32_t)lane << 26) | ((uint32_t)cnt & kRecBeforeMask);
          if (REC32) *(uint4*)(wave_recs + wrec + rank) = make_uint4(2u, 0u, (uint32_t)start | ((uint32_t)(my_len + kRecPosBias) << 16), r.meta);
          else
          wave_recs[wrec + rank] = r;
        }
        wrec += __builtin_popcountll(has);
        if (MODE == ST_RECORDS && lane == 0) wave_nrecs[w] = wrec;
      }
      if (tail) ++cnt;
      if (MODE == ST_RECORDS) {
        // matches of the whole wavefront: the CSR offsets are built from these 1/64th as many sums
        int wt = live ? cnt : 0;
        for (int off = 32; off > 0; off >>= 1) wt += __shfl_xor(wt, off);
        if (lane == 0) wave_nrecs[nwaves_total + w] = wt;
      }
      if (MODE == ST_FIRST) {
        if (live) {
          // OnePass '$' fixup: the walk reached the end of the text alive in an end-accepting state
          if (AUTO == 2 && p.off_fa_end >= 0 && stg_lds[(p.off_fa_end - p.off_fa_cls) + (q4 >> p.fa_cshift)]) res_e = my_len;
          out_s[my_text] = res_e >= 0 ? 0 : -1;
          out_e[my_text] = res_e;
        }
      } else if (MODE == ST_SEARCH) {
        if (live) {
          if (!done && tail) { res_s = p.st_fixed_len > 0 ? my_len - p.st_fixed_len : start; res_e = my_len; }
          out_s[my_text] = res_s;
          out_e[my_text] = res_e;
        }
      } else if (MODE == ST_FUSED) {
        // my task's count goes out at once; the task scanned before it is finished now
        int wt = live ? cnt : 0;
        for (int off = 32; off > 0; off >>= 1) wt += __shfl_xor(wt, off);
        fused_publish(fz.ctrl, w, nwaves_total, (uint32_t)wt, lane);
        if (pend)
          fused_finish<REC32 != 0>(fz, recs + (((int64_t)blockIdx.x * kStreamWaves + wave) * 2 + pend_half) * fz.rec_cap,
                                   pend_w, nwaves_total, n, pend_cnt, pend_wrec, tile, 64 * kRowPitch, p.st_fixed_len, lane);
        pend = true; pend_w = w; pend_cnt = live ? cnt : 0; pend_wrec = wrec; pend_half = half;
        half ^= 1;
      } else {
        if (live) counts[my_text] = cnt;
      }
    }
  }
  if (MODE == ST_FUSED && pend)
    fused_finish<REC32 != 0>(fz, recs + (((int64_t)blockIdx.x * kStreamWaves + wave) * 2 + pend_half) * fz.rec_cap,
                             pend_w, nwaves_total, n, pend_cnt, pend_wrec, tile, 64 * kRowPitch, p.st_fixed_len, lane);
#undef MRX_LOAD_CHUNK
#undef fz
}

// ---- ragged batches: texts handed to lanes as the lanes fall free -----------------------------------
// k_stream_findall gives a wavefront 64 consecutive texts, one per lane, and runs as long as the longest
// of them: with lengths U[64, 1024] 46 % of the lane-steps are idle.  Here a wavefront owns kDynTexts
// consecutive texts and a lane that reaches the end of its text takes the next unassigned one -- at the
// next 128-byte chunk boundary, because the text tile is filled chunk by chunk for all 64 rows at once.
// A row's descriptor in the tile's pad bytes is {base, frame end, fallback offset}; for a text taken over
// at chunk c the base is moved back by c x 128 bytes, so the common chunk counter keeps addressing every
// row, and the lane's frame coordinates carry the same shift (mis = c x 128 + address & 15).  The lanes
// that will finish inside the current chunk are known before it is processed (frame end <= chunk end), so
// their next texts are chosen -- ballot, rank, one wave-uniform cursor -- and their descriptors rewritten
// BEFORE the next chunk is prefetched.  Needs the reset byte (DevPlan::st_reset_byte): bytes of a chunk
// outside the lane's text are replaced by it and every group takes the branch-free steps, which also
// emits the match that runs to the end of a text unless the text ends exactly on the chunk boundary
// (checked when the lane lets go of the text).  Records name the text by its index in the task (8 bits)
// next to 24 bits of "matches of this text so far"; k_decode<., ., ., DYN> reads them that way.
#ifndef MRX_DYN_TEXTS
#define MRX_DYN_TEXTS 256
#endif
constexpr int kDynTexts = MRX_DYN_TEXTS;   // 256, 512 or 1024
constexpr int kDynShift = kDynTexts <= 256 ? 24 : kDynTexts <= 512 ? 23 : 22;   // record meta: text index above, matches so far below
constexpr uint32_t kDynBeforeMask = (1u << kDynShift) - 1u;
static_assert(kDynTexts % 64 == 0 && kDynTexts <= 1024, "task size");
// record region of task w: a text yields at most len / 16 + 3 records, so kDynTexts of them starting at byte
// first_off need at most bytes / 16 + 3 kDynTexts slots; regions floor(off / 16) + 4 kDynTexts w apart cannot overlap
__host__ __device__ inline int64_t rec_region_dyn(int64_t first_off, int64_t w) { return (first_off >> 4) + 4 * kDynTexts * w; }

template <int MODE, int AUTO, int REC32>
__global__ __launch_bounds__(64 * kStreamWaves) void k_stream_dyn(
    DevPlan p, const uint8_t* __restrict__ blob, const uint8_t* __restrict__ data, const int64_t* __restrict__ offsets,
    const int32_t* __restrict__ vlen, int64_t n, int32_t* __restrict__ counts, int32_t* __restrict__ wave_nrecs,
    EvRec* __restrict__ recs, int32_t* __restrict__ out_s, int32_t* __restrict__ out_e) {
  static_assert(MODE == ST_RECORDS || MODE == ST_COUNT || MODE == ST_SEARCH, "search modes only");
#ifndef MRX_DYN_CH
#define MRX_DYN_CH 128
#endif
  constexpr int CH = MRX_DYN_CH, kRowPitch = CH + 16, LPR = CH / 16, RPI = 64 / LPR, NL = 64 / RPI;
  __shared__ __align__(16) uint8_t tiles[kStreamWaves][64 * kRowPitch];
  __shared__ __align__(16) uint16_t col_lds[256];
  __shared__ __align__(16) uint4 pmask[17];   // pmask[x]: the first x bytes of a 16-byte group set
  // the task's texts in the order they are handed out: LONGEST FIRST (by length class), so that what is left for the
  // end of the task are short texts and the lanes finish close together
  __shared__ uint16_t perm_all[kStreamWaves][kDynTexts];
  __shared__ int hist_all[kStreamWaves][32];
  if (threadIdx.x < 17) {
    const int x = threadIdx.x;
    uint32_t w[4];
    for (int j = 0; j < 4; ++j) {
      const int nb = x - 4 * j;
      w[j] = nb <= 0 ? 0u : nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
    }
    pmask[x] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  const uint32_t fillw = (uint32_t)(p.st_reset_byte & 0xFF) * 0x01010101u;
  extern __shared__ __align__(16) uint8_t stg_lds[];
  if (AUTO == 1) {
    const uint16_t* src = (const uint16_t*)(blob + p.off_stcol);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) col_lds[i] = src[i];
  } else if (AUTO == 3) {
    const uint32_t* src = (const uint32_t*)(blob + p.off_stcol);
    uint32_t* dst = (uint32_t*)stg_lds;
    for (int i = threadIdx.x; i < 512; i += blockDim.x) dst[i] = src[i];
  } else {
    const uint32_t* src = (const uint32_t*)(blob + p.off_stg_cls);
    uint32_t* dst = (uint32_t*)stg_lds;
    for (int i = threadIdx.x; i < (p.stg_bytes >> 2); i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const uint64_t* col64_lds = (const uint64_t*)stg_lds;   // AUTO == 3
  const uint8_t* cls_lds = stg_lds;
  const uint16_t* tr_lds = (const uint16_t*)(stg_lds + (p.off_stg_trans - p.off_stg_cls));
  const uint8_t* acc_lds = stg_lds + (p.off_stg_acc - p.off_stg_cls);
  const uint32_t* pair_lds = (const uint32_t*)(stg_lds + (p.off_stg_pair - p.off_stg_cls));   // AUTO == 4
  const uint32_t accmask = p.st_accept_mask;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* tile = tiles[wave];
  const int seg = lane % LPR, rsub = lane / LPR;
  const int64_t ntasks = (n + kDynTexts - 1) / kDynTexts;
  auto accepting = [&](uint32_t q4) -> bool {
    return AUTO == 4 ? acc_lds[q4 >> (2 * p.st_cshift)] != 0
         : AUTO == 2 ? acc_lds[q4 >> p.st_cshift] != 0
                     : ((accmask >> (q4 >> (AUTO == 3 ? 3 : 2))) & 1u) != 0;
  };
  for (int64_t task = (int64_t)blockIdx.x * kStreamWaves + wave; task < ntasks; task += (int64_t)gridDim.x * kStreamWaves) {
    const int64_t T0 = task * kDynTexts;
    const int64_t T1 = T0 + kDynTexts < n ? T0 + kDynTexts : n;
    const int ntask = (int)(T1 - T0);
    int next = 64;                                // position in the hand-out order of the first text nobody has taken (wave uniform)
    // what a lane knows about the text it is walking
    int64_t my_text = T0 + lane;
    bool active = lane < ntask;
    int my_len = 0, mis = 0, flen = 0;           // flen: frame position behind the text's last byte
    // row descriptor of a text (byte offset o0, length len; t < 0: none) taken over at chunk `shift`
    auto describe = [&](int64_t t, int64_t o0, int len, int shift, int& len_o, int& mis_o, int& flen_o) {
      uintptr_t addr = (uintptr_t)blob;
      if (t < 0) len = 0;
      if (len > 0) addr = (uintptr_t)(data + o0);
      // (the frame begins at the 128-byte line of the text's first byte: see k_stream_findall)
      const int m0 = len > 0 ? (int)(addr & (CH - 1)) : 0;
      const uintptr_t rb = (addr & ~(uintptr_t)(CH - 1)) - (uintptr_t)shift;
      len_o = len; mis_o = shift + m0; flen_o = shift + m0 + len;
      // (no text, or an empty one: frame end 0 -- every load of the row falls back to the row's first block)
      *(uint4*)(tile + lane * kRowPitch + CH) =
          make_uint4((uint32_t)rb, (uint32_t)((uint64_t)rb >> 32), len > 0 ? (uint32_t)flen_o : 0u, (uint32_t)shift);
    };
    // where the task's texts lie: lane l keeps offsets[T0 + 64 j + l] (and the view lengths) for every j, read once;
    // a lane taking text ti gets them by cross-lane reads instead of a dependent global load in front of the next
    // chunk's prefetch.  (off_of / len_of are cross-lane: every lane of the wavefront calls them together.)
    constexpr int KJ = kDynTexts / 64;
    int64_t offr[KJ + 1];
    int lenr[KJ];
#pragma unroll
    for (int j = 0; j <= KJ; ++j) {
      const int64_t t = T0 + 64 * j + lane < T1 ? T0 + 64 * j + lane : T1;   // (clamped: offsets[T1] exists, views excepted)
      offr[j] = vlen ? offsets[t < T1 ? t : T1 - 1] : offsets[t];
      if (j < KJ) lenr[j] = vlen ? vlen[t < T1 ? t : T1 - 1] : 0;
    }
    auto off_of = [&](int ti) -> int64_t {
      int64_t v = __shfl(offr[0], ti & 63);
#pragma unroll
      for (int j = 1; j <= KJ; ++j) { const int64_t u = __shfl(offr[j], ti & 63); v = (ti >> 6) == j ? u : v; }
      return v;
    };
    if (!vlen) {   // lengths from the offsets: the next text's offset is the neighbour lane's (lane 63: the next row's lane 0)
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        const int64_t up = __shfl_down(offr[j], 1), wrap = __shfl(offr[j + 1], 0);
        lenr[j] = (int)((lane == 63 ? wrap : up) - offr[j]);
      }
    }
    auto len_of = [&](int ti) -> int {
      int v = __shfl(lenr[0], ti & 63);
#pragma unroll
      for (int j = 1; j < KJ; ++j) { const int u = __shfl(lenr[j], ti & 63); v = (ti >> 6) == j ? u : v; }
      return v;
    };
    // hand-out order: counting sort of the task's texts by length class (32 classes of the task's longest text),
    // longest class first
    uint16_t* perm = perm_all[wave];
    int* hist = hist_all[wave];
    {
      int mylen[KJ], tmax = 0;
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        const int ti = 64 * j + lane;
        const int l = len_of(ti < ntask ? ti : 0);
        mylen[j] = ti < ntask ? l : 0;
        tmax = max(tmax, mylen[j]);
      }
      for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));
      const int sh = tmax >= 32 ? 32 - __builtin_clz((unsigned)tmax) - 5 : 0;   // tmax >> sh < 32
      __builtin_amdgcn_wave_barrier();
      if (lane < 32) hist[lane] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      int slot[KJ], cls[KJ];
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        cls[j] = 31 - (mylen[j] >> sh);
        slot[j] = 64 * j + lane < ntask ? atomicAdd(&hist[cls[j]], 1) : 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      int hv = lane < 32 ? hist[lane] : 0, incl = hv;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
      __builtin_amdgcn_wave_barrier();
      if (lane < 32) hist[lane] = incl - hv;   // first position of the class
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int j = 0; j < KJ; ++j)
        if (64 * j + lane < ntask) perm[hist[cls[j]] + slot[j]] = (uint16_t)(64 * j + lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    {
      const int ti = active ? (int)perm[lane] : 0;
      const int64_t o0 = off_of(ti);
      const int l0 = len_of(ti);
      my_text = T0 + ti;
      describe(active ? my_text : -1, o0, l0, 0, my_len, mis, flen);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t q4 = 0;
    int start = 0, cnt = 0, wtotal = 0;
    int wrec = 0;
    bool done = false;
    int res_s = -1, res_e = -1;
    EvRec* wave_recs = MODE == ST_RECORDS ? recs + rec_region_dyn(offsets[T0], task) : nullptr;
    uint4 v[NL];
#define MRX_DYN_LOAD(CB)                                                                  \
    do {                                                                                  \
      _Pragma("unroll") for (int j_ = 0; j_ < NL; ++j_) {                                  \
        const uint4 rs_ = *(const uint4*)(tile + (RPI * j_ + rsub) * kRowPitch + CH);      \
        uint32_t fo_ = (uint32_t)(CB) + seg * 16;                                          \
        if (fo_ >= rs_.z) fo_ = rs_.w;   /* outside the row's frame: its first block */    \
        v[j_] = MRX_LDG((const uint4*)((const uint8_t*)(((uint64_t)rs_.y << 32) | rs_.x) + fo_)); \
      }                                                                                    \
    } while (0)
    MRX_DYN_LOAD(0);
    uint8_t* wr = tile + rsub * kRowPitch + seg * 16;
    for (int cbase = 0;; cbase += CH) {
#pragma unroll
      for (int j = 0; j < NL; ++j) *(uint4*)(wr + j * RPI * kRowPitch) = v[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // lanes that are through with their text once this chunk is done take their next one now
      const bool fin = active && (flen <= cbase + CH || (MODE == ST_SEARCH && done));
      const uint64_t fm = __ballot(fin);
      int64_t n_text = -1;
      int n_len = 0, n_mis = 0, n_flen = 0;
      if (fm) {   // wave uniform
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0));
        const int cpos = next + rank;                                   // my place in the hand-out order, if I take a text
        const int cti = cpos < ntask ? (int)perm[cpos] : 0;
        const int64_t c0 = off_of(cti);                                 // (all lanes take part in the cross-lane reads)
        const int cl = len_of(cti);
        if (fin) {
          n_text = cpos < ntask ? T0 + cti : -1;
          describe(n_text, c0, cl, cbase + CH, n_len, n_mis, n_flen);
        }
        next += __builtin_popcountll(fm);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      const bool any_next = __any((active && !fin) || n_text >= 0);
      if (any_next) MRX_DYN_LOAD(cbase + CH);   // next chunk, in flight while this one is walked

      const int lim = active ? flen - cbase : 0;   // frame bytes [lo, lim) of this chunk are my text
      const int lo = active ? mis - cbase : 0;
      const bool all_inside = __all(lim >= CH && lo <= 0);
      uint32_t F_even = 0, meta_even = 0, sp_even = 0;
      // (count: the groups are independent but for two registers, and fully unrolled the compiler interleaves
      // all eight -- 180-256 VGPRs; two at a time keeps it at the other modes' size)
      constexpr int kUnroll = MODE == ST_COUNT ? 2 : CH / 16;
#pragma unroll kUnroll
      for (int g = 0; g < CH / 16; ++g) {
        const uint4 wv = *(const uint4*)(tile + lane * kRowPitch + g * 16);
        uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
        if (!all_inside) {   // wave uniform
          const int a = min(max(lo - g * 16, 0), 16), b = min(max(lim - g * 16, 0), 16);   // inside: [a, b)
          const uint4 pa = pmask[a], pb = pmask[b > a ? b : a];
          const uint32_t m[4] = {pb.x & ~pa.x, pb.y & ~pa.y, pb.z & ~pa.z, pb.w & ~pa.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) words[j] = (words[j] & m[j]) | (fillw & ~m[j]);
        }
        uint32_t F = 0;
        if (AUTO == 4) {
          uint32_t cc[16], pi[8];
#pragma unroll
          for (int k = 0; k < 16; ++k) cc[k] = cls_lds[(words[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
          for (int k = 0; k < 8; ++k) pi[k] = (cc[2 * k] << p.st_cshift) | cc[2 * k + 1];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t e = pair_lds[q4 + pi[k]];
            q4 = e >> 4;
            F = __builtin_amdgcn_alignbit(e, F, 4);
          }
        } else if (AUTO == 2) {
          uint32_t cc[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) cc[k] = cls_lds[(words[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const uint32_t e = tr_lds[q4 + cc[k]];
            q4 = e >> 2;
            F = __builtin_amdgcn_alignbit(e, F, 2);
          }
        } else if (AUTO == 3) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint64_t cw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) cw[k] = col64_lds[(words[(8 * h + k) >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const uint32_t e = (uint32_t)(cw[k] >> q4);
              q4 = e & 0x38u;
              F = __builtin_amdgcn_alignbit(e, F, 2);
            }
          }
        } else {
          uint32_t cv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) cv[k] = col_lds[(words[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const uint32_t e = cv[k] >> q4;
            q4 = e & 0xCu;
            F = __builtin_amdgcn_alignbit(e, F, 2);
          }
        }
        const uint32_t em = F & 0xAAAAAAAAu;
        const uint32_t ns = F & 0x55555555u;
        const int gbase = cbase + g * 16 - mis;   // text position of the group's first byte
        const uint32_t tix = (uint32_t)(my_text - T0) << kDynShift;
        if (MODE == ST_RECORDS && REC32) {
          if ((g & 1) == 0) {
            F_even = F;
            sp_even = (uint32_t)start | ((uint32_t)(gbase + kRecPosBias) << 16);
            meta_even = tix | ((uint32_t)cnt & kDynBeforeMask);
          } else {
            const bool any = ((F_even | F) & 0xAAAAAAAAu) != 0u;
            const uint64_t has = __ballot(any);
            if (has) {
              if (any) {
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
                *(uint4*)(wave_recs + wrec + rank) = make_uint4(F_even, F, sp_even, meta_even);
              }
              wrec += __builtin_popcountll(has);
            }
          }
        } else if (MODE == ST_RECORDS) {
          const uint64_t has = __ballot(em != 0);
          if (has) {
            if (em) {
              const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
              *(uint4*)(wave_recs + wrec + rank) = make_uint4(F, (uint32_t)start, (uint32_t)gbase, tix | ((uint32_t)cnt & kDynBeforeMask));
            }
            wrec += __builtin_popcountll(has);
          }
        }
        if (MODE == ST_SEARCH) {
          if (!done && em) {
            const int kk = __builtin_ctz(em) >> 1;
            const uint32_t nsb = ns & ((1u << (2 * kk)) - 1u);
            res_s = nsb ? gbase + ((31 - __builtin_clz(nsb)) >> 1) : start;
            res_e = gbase + kk;
            if (p.st_fixed_len > 0) res_s = res_e - p.st_fixed_len;
            done = true;
          }
        }
        cnt += __builtin_popcount(em);
        if (ns) start = gbase + ((31 - __builtin_clz(ns)) >> 1);
      }
      __builtin_amdgcn_wave_barrier();
      // let go of finished texts: a walk still in an accepting state ends at the end of its text (only when
      // the text ended exactly on the chunk boundary -- otherwise the reset byte behind it has emitted it)
      {
        const bool tail = fin && accepting(q4) && !(MODE == ST_SEARCH && done);
        if (MODE == ST_RECORDS) {
          const uint64_t has = __ballot(tail);
          if (has) {
            if (tail) {
              const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
              const uint32_t meta = ((uint32_t)(my_text - T0) << kDynShift) | ((uint32_t)cnt & kDynBeforeMask);
              if (REC32) *(uint4*)(wave_recs + wrec + rank) = make_uint4(2u, 0u, (uint32_t)start | ((uint32_t)(my_len + kRecPosBias) << 16), meta);
              else *(uint4*)(wave_recs + wrec + rank) = make_uint4(2u, (uint32_t)start, (uint32_t)my_len, meta);
            }
            wrec += __builtin_popcountll(has);
          }
        }
        if (fin) {
          if (tail) ++cnt;
          if (MODE == ST_SEARCH) {
            if (!done && tail) { res_s = p.st_fixed_len > 0 ? my_len - p.st_fixed_len : start; res_e = my_len; }
            out_s[my_text] = res_s;
            out_e[my_text] = res_e;
          } else {
            counts[my_text] = cnt;
            wtotal += cnt;
          }
          my_text = n_text; active = n_text >= 0;
          my_len = n_len; mis = n_mis; flen = active ? n_flen : 0;
          q4 = 0; start = 0; cnt = 0; done = false; res_s = -1; res_e = -1;
        }
      }
      if (!any_next) break;
    }
#undef MRX_DYN_LOAD
    if (MODE == ST_RECORDS) {
      for (int off = 32; off > 0; off >>= 1) wtotal += __shfl_xor(wtotal, off);
      if (lane == 0) { wave_nrecs[task] = wrec; wave_nrecs[ntasks + task] = wtotal; }
    }
  }
}

// ST_FUSED set-up, one small launch in front of the scan: zero the ticket word, the error word and the
// descriptors, and place the scan's arguments in device memory.
__global__ __launch_bounds__(kBlock) void k_fused_init(unsigned long long* __restrict__ ctrl, int64_t words,
                                                       FusedArgs* __restrict__ dst, FusedArgs args) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x)
    ctrl[i] = 0ull;
  if (blockIdx.x == 0 && threadIdx.x == 0) *dst = args;
}

// records -> CSR spans.  One wavefront per 64 consecutive texts (the wavefront that
// produced the stream).  The wavefront's output range [prefix[first], prefix[last+1])
// is contiguous but the stream is ordered by (group, lane), so spans are first
// scattered into an LDS tile at their final relative position and the tile is then
// written out with fully coalesced 8-byte stores.  Per tile the stream is read once
// (coalesced 16-byte loads; re-reads for further tiles come from L2).
#ifndef MRX_DECODE_TILE
#define MRX_DECODE_TILE 2048
#endif
constexpr int kDecodeTile = MRX_DECODE_TILE;  // spans per LDS tile and wavefront (16 KiB): one pass for typical wavefronts
// (above 3 x TILE spans per wavefront k_decode makes one pass with direct stores: kDecodeDirect inside the kernel)
#ifndef MRX_DECODE_BATCH
#define MRX_DECODE_BATCH 8
#endif
constexpr int kDecodeBatch = MRX_DECODE_BATCH;    // independent 16-byte record loads in flight per lane

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;  // per thread
constexpr int kScanTile = kScanBlock * kScanItems;

// PACK16: every position of the batch fits 16 bits (texts of at most 65535 bytes), so a span takes
// 4 bytes in the LDS tile instead of 8 -- half the LDS per wavefront, twice the resident
// wavefronts for this latency-bound kernel.
// VBASE: pieces of long texts (see k_stream_findall VIRT): positions are piece-relative in the records
// and become text-relative by adding vbase[piece]; `prefix` is then per piece (k_virt_prefix picks
// the texts' entries).
// DYN: the records of k_stream_dyn -- a stream per task of kDynTexts texts, records name their text by its
// index in the task (meta >> 24) and count its matches so far in 24 bits.
#ifndef MRX_DYN_DECODE_TILE
#define MRX_DYN_DECODE_TILE 2048   // k_stream_dyn's 256-text tasks (A/B: tools/variants.sh)
#endif
// TILE: spans per LDS tile and wavefront.  2048 keeps five workgroups per CU; batches of texts of 768 bytes and more
// (16-bit positions) take 3072 -- config 4's wavefronts hold 2 600 spans and needed two passes over their records
// (findall 0.505 -> 0.452 ms), config 2 is unchanged, 256-byte texts (config 3) lose 3 % to the lower occupancy and
// keep 2048.
template <bool PACK16, bool VBASE = false, bool REC32 = false, bool DYN = false, int TILE = kDecodeTile>
__global__ __launch_bounds__(kBlock) void k_decode(int64_t n, const int32_t* __restrict__ wave_nrecs,
                                                   const EvRec* __restrict__ recs, int64_t rec_row,
                                                   const int64_t* __restrict__ offsets,
                                                   const int32_t* __restrict__ counts,
                                                   const int64_t* __restrict__ wave_base,
                                                   const int64_t* __restrict__ scan_block_sums,
                                                   int64_t* __restrict__ prefix,
                                                   int32_t* __restrict__ spans, int64_t span_cap,
                                                   int fixed_len, int64_t* __restrict__ total_out,
                                                   const int32_t* __restrict__ vbase = nullptr,
                                                   const int64_t* __restrict__ base = nullptr, int reverse = 0) {
  // reverse: wavefronts take the 64-text groups last first -- the records the scan wrote last are the ones still
  // in L2 / Infinity Cache when this kernel starts
  // base: spans of the texts in front of this launch's (the second half of a split findall, see findall_split)
  static_assert(!(PACK16 && VBASE), "text-relative positions of a long text do not fit 16 bits");
  static_assert(!(DYN && VBASE), "pieces are not handed out dynamically");
  using Slot = typename std::conditional<PACK16, uint32_t, int2>::type;
  __shared__ Slot tile_all[kBlock / 64][TILE];
  __shared__ int dense_upto[kBlock / 64][64];   // dense path: spans of each of the wavefront's texts expanded so far
  constexpr int kDecodeTile = TILE, kDecodeDirect = 3 * TILE;   // (shadow the file-scope defaults)
  __shared__ int rel_all[DYN ? kBlock / 64 : 1][DYN ? kDynTexts : 1];   // DYN: spans of the task's texts before each text
  constexpr int kTexts = DYN ? kDynTexts : 64;
  constexpr uint32_t kBefore = DYN ? kDynBeforeMask : kRecBeforeMask;
  constexpr int kTextShift = DYN ? kDynShift : 26;
  const int lane = threadIdx.x & 63;
  Slot* tile = tile_all[threadIdx.x >> 6];
  int* rel_lds = rel_all[DYN ? (threadIdx.x >> 6) : 0];
  const int64_t nw = (n + kTexts - 1) / kTexts;
  const int waves_per_block = blockDim.x >> 6;
  for (int64_t w0 = (int64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6); w0 < nw;
       w0 += (int64_t)gridDim.x * waves_per_block) {
    const int64_t w = reverse ? nw - 1 - w0 : w0;
    const int64_t first = w * kTexts;
    const int64_t i = first + lane;
    // CSR offsets of my 64 texts: exclusive scan of their counts on top of the wavefront's base
    // wave_base is exclusive within its k_scan_local tile of kScanTile wavefronts; the tiles before
    // it (at most a few dozen sums) are added here instead of by two more scan launches
    int64_t pre0 = wave_base[w] + (base ? *base : 0);
    {
      const int64_t tiles_before = w / kScanTile;
      int64_t part = 0;
      for (int64_t b = lane; b < tiles_before; b += 64) part += scan_block_sums[b];
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
      pre0 += part;
    }
    int my_rel = 0, total_spans = 0;
    if (DYN) {
      __builtin_amdgcn_wave_barrier();
      int carry = 0;
      for (int sb = 0; sb < kDynTexts; sb += 64) {
        const int64_t t = first + sb + lane;
        const int c = t < n ? counts[t] : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int v = __shfl_up(incl, d);
          if (lane >= d) incl += v;
        }
        rel_lds[sb + lane] = carry + incl - c;
        if (t < n) prefix[t] = pre0 + carry + incl - c;
        if (t == n - 1) { prefix[n] = pre0 + carry + incl; *total_out = pre0 + carry + incl; }
        carry += __shfl(incl, 63);
      }
      total_spans = carry;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
    const int my_cnt = i < n ? counts[i] : 0;
    int incl = my_cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if (lane >= d) incl += v;
    }
    my_rel = incl - my_cnt;   // start of my text's spans in the wavefront's range
    total_spans = __shfl(incl, 63);
    if (i < n) prefix[i] = pre0 + my_rel;
    if (i == n - 1) { prefix[n] = pre0 + incl; *total_out = pre0 + incl; }
    }
    const int total_recs = wave_nrecs[w];
    const EvRec* wave_recs = recs + (DYN ? rec_region_dyn(offsets[first], w)
                                         : offsets ? rec_region_start(offsets[first], w) : first * rec_row);
    const int my_vb = (VBASE && i < n) ? vbase[i] : 0;
    if (total_spans > kDecodeDirect) {
      // dense matches: the tile passes would re-read the stream total_spans / kDecodeTile times.  One pass instead.
      // Round 3: the spans of a batch of records no longer leave as per-lane 8-byte stores scattered over the 64 texts'
      // regions (config 5: 262 K partly written lines in flight across the device, more than L2 holds, so most of
      // them reached HBM as read-modify-writes: 1.65 TB/s).  The LDS tile becomes 64 ROW WINDOWS, one per text of the
      // wavefront, kRowCap spans each: a record's spans go to their text's row at (index within the text - spans of
      // that text already written) -- the index is in the record (matches of the text before it) -- and after every
      // batch each row's new spans, which are contiguous in the output, leave with one coalesced store instruction
      // per row.  A span that falls behind its row's window (one text far denser than a batch's average) is stored
      // directly, as before.
      constexpr int kRowCap = (DYN || !PACK16) ? 0 : kDecodeTile / 64;   // (16-bit positions only: 48 spans of 4 bytes)
      int done_t = 0;                       // lane t: spans of text t written so far
      for (int j = 0; j < total_recs; j += 64 * kDecodeBatch) {
        EvRec rr[kDecodeBatch];
#pragma unroll
        for (int u = 0; u < kDecodeBatch; ++u) {
          const int o = j + u * 64 + lane;
          rr[u].F = 0; rr[u].start = 0; rr[u].pos_base = 0; rr[u].meta = 0;
          if (o < total_recs) {   // last use of the record: non-temporal
            const uint4 q_ = mrx_ldg((const uint4*)(wave_recs + o));
            rr[u].F = q_.x; rr[u].start = (int32_t)q_.y; rr[u].pos_base = (int32_t)q_.z; rr[u].meta = q_.w;
          }
        }
        if (kRowCap > 0) {
          dense_upto[threadIdx.x >> 6][lane] = done_t;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int u = 0; u < kDecodeBatch; ++u) {
          EvRec r = rr[u];
          const int txt = DYN ? 0 : (int)(r.meta >> 26);
          const int rel_t = DYN ? rel_lds[r.meta >> kTextShift] : __shfl(my_rel, txt);
          if (VBASE) { const int vb = __shfl(my_vb, (int)(r.meta >> 26)); r.start += vb; r.pos_base += vb; }
          const int done_r = kRowCap > 0 ? __shfl(done_t, txt) : 0;   // spans of my record's text already written
          int within = (int)(r.meta & kBefore);                       // index of my record's first span within its text
          int64_t dst = pre0 + rel_t + within;
          uint32_t Fw = r.F;
          int pb = REC32 ? (int)((uint32_t)r.pos_base >> 16) - kRecPosBias : r.pos_base;
          int rstart = REC32 ? (int)((uint32_t)r.pos_base & 0xFFFFu) : r.start;
#pragma unroll
          for (int half = 0; half < (REC32 ? 2 : 1); ++half) {
            uint32_t em = Fw & 0xAAAAAAAAu;
            const uint32_t ns = Fw & 0x55555555u;
            while (em) {
              const int kk = __builtin_ctz(em) >> 1;
              const uint32_t nsb = ns & ((1u << (2 * kk)) - 1u);
              int st = nsb ? pb + ((31 - __builtin_clz(nsb)) >> 1) : rstart;
              if (fixed_len > 0) st = pb + kk - fixed_len;
              const int slot = within - done_r;
              if (kRowCap > 0 && slot < kRowCap) {
                if constexpr (PACK16) tile[txt * (kRowCap > 0 ? kRowCap : 1) + slot] = ((uint32_t)st << 16) | (uint32_t)(pb + kk);
              } else if (dst < span_cap) {
                *(int2*)(spans + 2 * dst) = make_int2(st, pb + kk);
              }
              ++dst; ++within;
              em &= em - 1;
            }
            if (REC32) {
              if (ns) rstart = pb + ((31 - __builtin_clz(ns)) >> 1);
              pb += 16;
              Fw = (uint32_t)r.start;
            }
          }
          if (kRowCap > 0 && (r.F | (REC32 ? (uint32_t)r.start : 0u)) != 0u) atomicMax(&dense_upto[threadIdx.x >> 6][txt], within);
        }
        if constexpr (PACK16 && !DYN) {
          // every record up to here is expanded: row t holds the spans [done_t, upto_t) of text t (those beyond the
          // window went out directly); one coalesced store per row
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int upto_t = dense_upto[threadIdx.x >> 6][lane];
          const int64_t row_dst = pre0 + my_rel + done_t;
          int row_n = upto_t - done_t;
          if (row_n > kRowCap) row_n = kRowCap;
          for (int r_ = 0; r_ < 64; ++r_) {
            const int n_r = __shfl(row_n, r_);
            if (n_r <= 0) continue;          // (wave uniform)
            const int64_t d_r = __shfl(row_dst, r_);
            if (lane < n_r && d_r + lane < span_cap) {
              const uint32_t v = (uint32_t)tile[r_ * (kRowCap > 0 ? kRowCap : 1) + lane];
              mrx_stg_span(spans + 2 * (d_r + lane), (int)(v >> 16), (int)(v & 0xFFFFu));
            }
          }
          done_t = upto_t;
          __builtin_amdgcn_wave_barrier();
        }
      }
      continue;
    }
    for (int tb = 0; tb < total_spans; tb += kDecodeTile) {
      for (int j = 0; j < total_recs; j += 64 * kDecodeBatch) {
        // several independent 16-byte loads in flight per lane before any is consumed
        EvRec rr[kDecodeBatch];
#pragma unroll
        for (int u = 0; u < kDecodeBatch; ++u) {
          const int o = j + u * 64 + lane;
          rr[u].F = 0; rr[u].start = 0; rr[u].pos_base = 0; rr[u].meta = 0;
          if (o < total_recs) {   // last use of the record: non-temporal
            const uint4 q_ = mrx_ldg((const uint4*)(wave_recs + o));
            rr[u].F = q_.x; rr[u].start = (int32_t)q_.y; rr[u].pos_base = (int32_t)q_.z; rr[u].meta = q_.w;
          }
        }
#pragma unroll
        for (int u = 0; u < kDecodeBatch; ++u) {
          EvRec r = rr[u];
          const int rel_t = DYN ? rel_lds[r.meta >> kTextShift] : __shfl(my_rel, (int)(r.meta >> 26));
          if (VBASE) { const int vb = __shfl(my_vb, (int)(r.meta >> 26)); r.start += vb; r.pos_base += vb; }
          int dst = rel_t + (int)(r.meta & kBefore) - tb;
          // REC32: {F even, F odd, start | (pos + 16) << 16, meta} -- two event words per record
          uint32_t Fw = r.F;
          int pb = REC32 ? (int)((uint32_t)r.pos_base >> 16) - kRecPosBias : r.pos_base;
          int rstart = REC32 ? (int)((uint32_t)r.pos_base & 0xFFFFu) : r.start;
#pragma unroll
          for (int half = 0; half < (REC32 ? 2 : 1); ++half) {
            uint32_t em = Fw & 0xAAAAAAAAu;
            const uint32_t ns = Fw & 0x55555555u;
            while (em) {
              const int kk = __builtin_ctz(em) >> 1;              // byte of this EMIT
              const uint32_t nsb = ns & ((1u << (2 * kk)) - 1u);   // NEWSTARTs strictly before it
              int st = nsb ? pb + ((31 - __builtin_clz(nsb)) >> 1) : rstart;
              if (fixed_len > 0) st = pb + kk - fixed_len;
              if (dst >= 0 && dst < kDecodeTile) {
                if constexpr (PACK16) tile[dst] = ((uint32_t)st << 16) | (uint32_t)(pb + kk);
                else tile[dst] = make_int2(st, pb + kk);
              }
              ++dst;
              em &= em - 1;
            }
            if (REC32) {   // on to the odd group: the walk alive at its start began at the even group's last NEWSTART
              if (ns) rstart = pb + ((31 - __builtin_clz(ns)) >> 1);
              pb += 16;
              Fw = (uint32_t)r.start;
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int cnt = total_spans - tb < kDecodeTile ? total_spans - tb : kDecodeTile;
      for (int k = lane; k < cnt; k += 64) {
        const int64_t dst = pre0 + tb + k;
        if (dst < span_cap) {
          if constexpr (PACK16) { const uint32_t v = tile[k]; mrx_stg_span(spans + 2 * dst, (int)(v >> 16), (int)(v & 0xFFFFu)); }
          else { const int2 v = tile[k]; mrx_stg_span(spans + 2 * dst, v.x, v.y); }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// event rows -> CSR spans (behind k_stream_findall<ST_ROWS>).  G lanes per text (a wavefront, half or quarter of one: 64 / G
// texts per wavefront); a round is 2 G pairs of event words = 64 G bytes of text, two pairs (one 16-byte load) per lane.
// What a record carried is derived here: the index of a lane's first span within its text is a prefix sum over the
// lanes' match-end counts, the walk alive at a lane's first byte began at the last NEWSTART of the lanes in front (a
// running maximum) -- both carried from round to round.  The spans of a round are contiguous in the output: they are
// laid out in the text's part of an LDS tile (16-bit positions, 4 bytes a span) and leave as coalesced 8-byte stores.
// The match that ends with the text has no event (the scan counts it): the text's count says whether there is one.
// prefix[] is complete before this launch (device_scan over the scan's counts).
constexpr int kRowsTile = 1024;   // spans per wavefront and pass
template <int G>
__global__ __launch_bounds__(kBlock) void k_decode_rows(int64_t n, const uint2* __restrict__ rows, int64_t row_pairs, int npairs,
                                                        int text_len, const int32_t* __restrict__ counts,
                                                        const int64_t* __restrict__ prefix, int32_t* __restrict__ spans,
                                                        int64_t span_cap, int fixed_len) {
  static_assert(G == 16 || G == 32 || G == 64, "lanes per text");
  constexpr int TPW = 64 / G, kTile = kRowsTile / TPW;   // texts per wavefront; spans per text and pass
  __shared__ uint32_t tile_all[kBlock / 64][kRowsTile];
  const int lane = threadIdx.x & 63, gl = lane & (G - 1), grp = lane / G;
  uint32_t* tile = tile_all[threadIdx.x >> 6] + grp * kTile;
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x >> 6) * TPW;
  auto load_pairs = [&](int64_t i, int p0) {   // pairs p0 + 2 gl, p0 + 2 gl + 1 of text i (zero beyond the text)
    const int a = p0 + 2 * gl;
    uint4 q = make_uint4(0, 0, 0, 0);
    if (i < n && a < npairs) {
      q = mrx_ldg((const uint4*)(rows + i * row_pairs + a));   // (row_pairs is even: 16-byte aligned; last use of the words)
      if (a + 1 >= npairs) { q.z = 0; q.w = 0; }
    }
    return q;
  };
  int64_t i = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * TPW + grp;
  uint4 q_next = load_pairs(i, 0);
  int64_t base_next = i < n ? prefix[i] : 0;
  int total_next = i < n ? counts[i] : 0;
  // (the trip count is the wavefront's: a group beyond the batch rides along with nothing to do)
  for (; __any(i < n); i += ngroups) {
    uint4 q = q_next;
    const int64_t base = base_next;
    const int total_i = total_next;
    if (npairs <= 2 * G) {   // the next text's first round travels while this one is expanded
      q_next = load_pairs(i + ngroups, 0);
      base_next = i + ngroups < n ? prefix[i + ngroups] : 0;
      total_next = i + ngroups < n ? counts[i + ngroups] : 0;
    }
    int carry_cnt = 0, carry_start = 0;
    for (int p0 = 0; p0 < npairs; p0 += 2 * G) {
      if (p0 > 0) q = load_pairs(i, p0);
      const uint32_t W[4] = {q.x, q.y, q.z, q.w};
      const int pb0 = 32 * (p0 + 2 * gl);
      int c = 0, ln = -1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c += __builtin_popcount(W[k] & 0xAAAAAAAAu);
        const uint32_t ns = W[k] & 0x55555555u;
        if (ns) ln = pb0 + 16 * k + ((31 - __builtin_clz(ns)) >> 1);
      }
      int incl = c, mx = ln;
#pragma unroll
      for (int d = 1; d < G; d <<= 1) {
        const int v = __shfl_up(incl, d, G), m = __shfl_up(mx, d, G);
        if (gl >= d) { incl += v; mx = max(mx, m); }
      }
      const int round_total = __shfl(incl, G - 1, G);
      int ex_start = __shfl_up(mx, 1, G);
      if (gl == 0) ex_start = -1;
      const int rstart0 = ex_start >= 0 ? ex_start : carry_start;
      const int before = incl - c;   // spans of this round in front of my first
      for (int tb = 0; __any(tb < round_total); tb += kTile) {
        int dst = before - tb;
        int rs = rstart0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t em = W[k] & 0xAAAAAAAAu;
          const uint32_t ns = W[k] & 0x55555555u;
          const int pb = pb0 + 16 * k;
          while (em) {
            const int kk = __builtin_ctz(em) >> 1;
            const uint32_t nsb = ns & ((1u << (2 * kk)) - 1u);   // NEWSTARTs strictly before this EMIT
            int st = nsb ? pb + ((31 - __builtin_clz(nsb)) >> 1) : rs;
            if (fixed_len > 0) st = pb + kk - fixed_len;
            if (dst >= 0 && dst < kTile) tile[dst] = ((uint32_t)st << 16) | (uint32_t)(pb + kk);
            ++dst;
            em &= em - 1;
          }
          if (ns) rs = pb + ((31 - __builtin_clz(ns)) >> 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int left = round_total - tb;
        const int cntp = left < 0 ? 0 : left < kTile ? left : kTile;
        const int64_t d0 = base + carry_cnt + tb;
        for (int k = gl; k < cntp; k += G) {
          const uint32_t v = tile[k];
          if (d0 + k < span_cap) mrx_stg_span(spans + 2 * (d0 + k), (int)(v >> 16), (int)(v & 0xFFFFu));
        }
        __builtin_amdgcn_wave_barrier();
      }
      carry_cnt += round_total;
      const int last = __shfl(mx, G - 1, G);
      if (last >= 0) carry_start = last;
    }
    if (i < n && carry_cnt < total_i && gl == 0) {   // the match that runs to the end of the text
      const int64_t d = base + carry_cnt;
      if (d < span_cap) mrx_stg_span(spans + 2 * d, fixed_len > 0 ? text_len - fixed_len : carry_start, text_len);
    }
    if (npairs > 2 * G) {
      q_next = load_pairs(i + ngroups, 0);
      base_next = i + ngroups < n ? prefix[i + ngroups] : 0;
      total_next = i + ngroups < n ? counts[i + ngroups] : 0;
    }
  }
}

// match_first of a "one or more bytes of a class" plan on long texts (DevPlan::off_fa_run): the match is
// the run of class bytes at 0, so a wavefront sweeps its text 1 KiB at a time and stops at the first byte
// outside the class instead of one lane walking it all (the reference's range_* / predefined_word
// benchmarks: one 10 KB match per text).
__global__ __launch_bounds__(kBlock) void k_first_run(DevPlan p, const uint8_t* __restrict__ blob, Layout lay, int64_t n,
                                                      int32_t* __restrict__ out_s, int32_t* __restrict__ out_e) {
  __shared__ uint8_t in_cls[256];
  for (int b = threadIdx.x; b < 256; b += blockDim.x) in_cls[b] = blob[p.off_fa_run + b];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += nwaves) {
    const Text t = lay.text(i);
    int run = t.len;   // no byte outside the class: the whole text
    if (t.len > 0) {
      const uintptr_t addr = (uintptr_t)t.ptr;
      const int mis = (int)(addr & 15);
      const uint8_t* frame = (const uint8_t*)(addr & ~(uintptr_t)15);
      const int end = mis + t.len;
      for (int blk = 0; blk < end; blk += 1024) {
        const int o = blk + 16 * lane;
        uint32_t bad = 0;   // bit k: frame byte o + k is text and outside the class
        if (o < end) {
          const uint4 v = mrx_ldg((const uint4*)(frame + o));
          const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (!in_cls[(w4[k >> 2] >> ((k & 3) * 8)) & 0xFFu]) bad |= 1u << k;
          const int lo = mis - o, hi = end - o;
          if (lo > 0) bad &= lo >= 16 ? 0u : ~((1u << lo) - 1u);
          if (hi < 16) bad &= hi <= 0 ? 0u : ((1u << hi) - 1u);
        }
        const uint64_t any = __ballot(bad != 0u);
        if (any) {
          const int w = __builtin_ctzll(any);
          const int pos = __builtin_amdgcn_readlane(o + (bad ? __builtin_ctz(bad) : 0), w);
          run = pos - mis;
          break;
        }
      }
    }
    if (lane == 0) { out_s[i] = run > 0 ? 0 : -1; out_e[i] = run > 0 ? run : -1; }
  }
}

// ---- long texts in pieces (k_stream_findall VIRT) -------------------------------------------
// Text t is cut every C bytes; piece k owns the text bytes [k C, min(len, (k + 1) C)) and starts its
// walk at the last synchronising byte before k C, at most kVirtBack bytes back.  A cut without such
// a byte is not made (the piece before it runs on, its own piece is empty).  Text t has
// max(1, ceil(len / C)) pieces, numbered vfirst[t] .. vfirst[t + 1] - 1 (vfirst = prefix sums).
constexpr int kVirtBack = 256;   // < 1008, see rec_region_start: pieces overlap by at most this much

__device__ __forceinline__ int virt_back(const uint8_t* __restrict__ sync, const uint8_t* __restrict__ txt, int c) {
  for (int b = 1; b <= kVirtBack && b <= c; ++b)
    if (sync[txt[c - b]]) return b;
  return -1;
}
// the text that owns piece v: last t with vfirst[t] <= v
__device__ __forceinline__ int64_t virt_text_of(const int64_t* __restrict__ vfirst, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;   // vfirst[lo] <= v < vfirst[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (vfirst[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}

// texts of one length: the same number of pieces each, no prefix sums needed
__global__ __launch_bounds__(kBlock) void k_virt_uniform(int64_t n, int64_t cpt, int64_t* __restrict__ vfirst) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n; t += (int64_t)gridDim.x * blockDim.x)
    vfirst[t] = t * cpt;
}
__global__ __launch_bounds__(kBlock) void k_virt_count(Layout lay, int64_t n, int C, int32_t* __restrict__ cnt) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int len = lay.text(t).len;
    cnt[t] = len <= C ? 1 : (len + C - 1) / C;
  }
}

// back[v] for piece v = (t, k): how far before the cut k C its synchronising byte lies; 0 for k = 0,
// -1 when there is none within kVirtBack bytes (the piece before it then runs on through this one)
__global__ __launch_bounds__(kBlock) void k_virt_check(Layout lay, int64_t n, const int64_t* __restrict__ vfirst,
                                                       int64_t nv, int C, const uint8_t* __restrict__ sync,
                                                       int32_t* __restrict__ back) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = virt_text_of(vfirst, n, v);
    const int k = (int)(v - vfirst[t]);
    back[v] = k == 0 ? 0 : virt_back(sync, lay.text(t).ptr, k * C);
  }
}

__global__ __launch_bounds__(kBlock) void k_virt_fill(Layout lay, int64_t n, const int64_t* __restrict__ vfirst,
                                                      int64_t nv, int C, const int32_t* __restrict__ back,
                                                      int64_t* __restrict__ vstart, int32_t* __restrict__ vlen,
                                                      uint32_t* __restrict__ vskip, int32_t* __restrict__ vbase,
                                                      int disjoint) {
  // disjoint (stepper plans): a piece ends where the next one begins -- at that piece's synchronising byte --
  // instead of running on to its cut with the overlap's events masked out (streaming plans)
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = virt_text_of(vfirst, n, v);
    const int64_t v0 = vfirst[t];
    const int k = (int)(v - v0), cpt = (int)(vfirst[t + 1] - v0);
    const Text tx = lay.text(t);
    const int64_t abs0 = tx.ptr - lay.data;
    const int b = back[v];
    const int64_t c = (int64_t)k * C;   // < len, or 0 for the one piece of an empty text
    // an empty piece sits at its cut: piece starts never decrease, which the record regions rely on
    // (rec_region_start)
    int64_t st = abs0 + c;
    int ln = 0, base = (int)c;
    uint32_t sk = 0;
    if (b >= 0) {
      // my piece runs to the next cut that has a synchronising byte, or to the end of the text
      int j = k + 1;
      while (j < cpt && back[v0 + j] == -1) ++j;
      const int e = j == cpt ? tx.len : (disjoint ? j * C - back[v0 + j] : j * C);
      st = abs0 + c - b; ln = e - (int)c + b; base = (int)c - b;
      sk = disjoint ? 0u : ((uint32_t)b | (e == tx.len ? 0x80000000u : 0u));
    }
    vstart[v] = st; vlen[v] = ln; vskip[v] = sk; vbase[v] = base;
  }
}

// How many of the first `nbytes` bytes of a batch can begin a walk of a stepper plan (start state has a transition on
// the byte, and the first-byte filter lets it through): the plain route's candidates.  Dense candidates are where the
// wavefront-per-text kernel loses to one lane per piece; sparse ones where it wins.
__global__ __launch_bounds__(kBlock) void k_candidate_density(DevPlan p, const uint8_t* __restrict__ blob,
                                                              const uint8_t* __restrict__ data, int64_t nbytes,
                                                              unsigned int* __restrict__ hits) {
  __shared__ uint8_t starts[256];
  const uint8_t* cls = blob + p.off_cls;
  const uint8_t* first = blob + p.off_first;
  const uint16_t* tr = (const uint16_t*)(blob + p.off_trans);
  const bool filt = (p.flags & PF_HAS_MATCHER) != 0;
  for (int b = threadIdx.x; b < 256; b += blockDim.x) starts[b] = (tr[cls[b]] != 0xFFFFu && (!filt || first[b])) ? 1 : 0;
  __syncthreads();
  unsigned int k = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * blockDim.x) k += starts[data[i]];
  for (int off = 32; off > 0; off >>= 1) k += __shfl_xor(k, off);
  if ((threadIdx.x & 63) == 0 && k) atomicAdd(hits, k);
}
// spans of piece v are piece-relative: make them text-relative (stepper plans, whose kernels know nothing of pieces).
// One lane per SPAN (round 4; a wavefront per piece before: 131 K pieces of a few spans each took 167 us for 7 M spans
// on the reference's phone texts -- a quarter of the call): a workgroup takes 1024 consecutive spans, finds the pieces
// that hold the first and the last of them by bisection of the piece offsets (once, wavefront 0), and every lane bisects
// within that short range (the lines are the workgroup's own: L1 / L2 hits).
__global__ __launch_bounds__(kBlock) void k_virt_add_base(int64_t nv, const int64_t* __restrict__ vprefix,
                                                          const int32_t* __restrict__ vbase, int32_t* __restrict__ spans,
                                                          int64_t span_cap) {
  __shared__ int64_t range[2];
  const int64_t total = vprefix[nv] < span_cap ? vprefix[nv] : span_cap;
  constexpr int kPer = 4;
  for (int64_t k0 = (int64_t)blockIdx.x * kBlock * kPer; k0 < total; k0 += (int64_t)gridDim.x * kBlock * kPer) {
    const int64_t k1 = k0 + kBlock * kPer - 1 < total - 1 ? k0 + kBlock * kPer - 1 : total - 1;
    __syncthreads();
    if (threadIdx.x < 128) {   // the last piece v with vprefix[v] <= k: wavefront 0 for k0, wavefront 1 for k1,
      // 64 probes per round trip (a plain bisection is 17 dependent loads: the latency of this kernel)
      const int lane = threadIdx.x & 63;
      const int64_t k = threadIdx.x < 64 ? k0 : k1;
      int64_t lo = 0, hi = nv;   // invariant: vprefix[lo] <= k < vprefix[hi] (vprefix[nv] = all spans > k)
      while (hi - lo > 1) {
        const int64_t step = (hi - lo + 63) >> 6;
        const int64_t probe = lo + (int64_t)(lane + 1) * step;   // lane's probe; beyond hi: "greater"
        const bool le = probe < hi && vprefix[probe] <= k;
        const uint64_t m = __ballot(le);
        const int cnt = __builtin_popcountll(m);   // probes are increasing, so the lanes that answer "<=" are a prefix
        const int64_t nlo = lo + (int64_t)cnt * step;
        const int64_t nhi = lo + (int64_t)(cnt + 1) * step;
        lo = nlo < hi ? nlo : lo;
        hi = nhi < hi ? nhi : hi;
      }
      if (lane == 0) range[threadIdx.x >> 6] = lo;
    }
    __syncthreads();
    const int64_t v_lo = range[0], v_hi = range[1];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int64_t k = k0 + (int64_t)j * kBlock + threadIdx.x;
      if (k > k1) break;
      int64_t lo = v_lo, hi = v_hi + 1;
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (vprefix[mid] <= k) lo = mid; else hi = mid;
      }
      const int b = vbase[lo];
      if (b != 0) {
        int2 sp = *(int2*)(spans + 2 * k);
        sp.x += b; sp.y += b;
        *(int2*)(spans + 2 * k) = sp;
      }
    }
  }
}
// per-text entries of the per-piece prefix sums / counts
__global__ __launch_bounds__(kBlock) void k_virt_prefix(int64_t n, const int64_t* __restrict__ vfirst,
                                                        const int64_t* __restrict__ vprefix,
                                                        int64_t* __restrict__ prefix) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n; t += (int64_t)gridDim.x * blockDim.x)
    prefix[t] = vprefix[vfirst[t]];
}
__global__ __launch_bounds__(kBlock) void k_virt_sum(int64_t n, const int64_t* __restrict__ vfirst,
                                                     const int32_t* __restrict__ vcounts,
                                                     int32_t* __restrict__ counts) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    int c = 0;
    for (int64_t v = vfirst[t]; v < vfirst[t + 1]; ++v) c += vcounts[v];
    counts[t] = c;
  }
}
// search: the first piece of a text that holds a match has the text's first match
__global__ __launch_bounds__(kBlock) void k_virt_first(int64_t n, const int64_t* __restrict__ vfirst,
                                                       const int32_t* __restrict__ vs,
                                                       const int32_t* __restrict__ ve, const int32_t* __restrict__ vbase,
                                                       int32_t* __restrict__ out_s, int32_t* __restrict__ out_e) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    int rs = -1, re = -1;
    for (int64_t v = vfirst[t]; v < vfirst[t + 1]; ++v)
      if (vs[v] >= 0) { rs = vs[v] + vbase[v]; re = ve[v] + vbase[v]; break; }
    out_s[t] = rs; out_e[t] = re;
  }
}
// ---- the `start` argument (Engine.match_first(text, start), engine.mojo:4-37) ------------------
// Every route of the reference treats match_first / match_next / is_match at `start` as the same
// operation at 0 on the bytes [start, len) with `start` added to the result (DFAEngine
// ._try_match_at_position and the search loops only ever look at text[pos:], dfa.mojo:1875-2026;
// LazyDFA builds its start state once, pikevm.mojo:714; exact-literal and prefilter paths call
// find(literal, start), matcher.mojo:768-796).  What differs is decided per text by k_view_fix: a
// '^' plan on the DFA or OnePass route answers None for start > 0 (dfa.mojo:1866-1867, 1887-1891,
// onepass.mojo:445), and start > len has its own answers (View::beyond_*).  So the kernels run
// unchanged on a view of the batch: text i = [offsets[i] + start_i, len_i - start_i).
__global__ __launch_bounds__(kBlock) void k_view_build(Layout lay, int64_t n, int32_t start, const int32_t* __restrict__ starts,
                                                       int64_t* __restrict__ vstart, int32_t* __restrict__ vlen,
                                                       uint32_t* __restrict__ vskip, int beyond_neg) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) {
    if (i == n) {   // an end offset for the kernels that size things by the batch's byte count
      vstart[n] = lay.offsets ? lay.offsets[n] : n * lay.stride;
      break;
    }
    const Text tx = lay.text(i);   // (also a view: the prefilter's view of a view)
    const int64_t a = tx.ptr - lay.data;
    const int len = tx.len;
    const int s0 = starts ? starts[i] : start;
    const int sc = s0 < 0 ? 0 : s0 > len ? len : s0;
    vstart[i] = a + sc;
    // beyond_neg (the backtracking matcher's match_first): a start behind the end is a view of NEGATIVE length --
    // position 0 of it lies s0 - len bytes behind the text's end, where no byte matches and '$' does not hold
    // either (nfa.mojo:998-1006 compares with the length), which is what bt_match_at() makes of it
    vlen[i] = beyond_neg && s0 > len ? len - s0 : len - sc;
    vskip[i] = 0x80000000u;   // k_stream_findall VIRT: a whole text (nothing skipped, may end a match at its end)
  }
}
// results of the view -> results of the text.  rule bits: 1 = None for start > 0 ('^' on the DFA /
// OnePass route), 2 = match_first at start > len is the empty match (start, start), 4 = is_match at
// start > len is true, 8 = is_match operation (out_flag), else spans, 16 = start > len was answered by the
// kernel itself (k_view_build's beyond_neg)
__global__ __launch_bounds__(kBlock) void k_view_fix(Layout lay, int64_t n, int32_t start, const int32_t* __restrict__ starts,
                                                     int rules, int32_t* __restrict__ out_s, int32_t* __restrict__ out_e,
                                                     uint8_t* __restrict__ out_flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int len = lay.text(i).len;
    const int s0 = starts ? starts[i] : start;
    const bool none = s0 < 0 || ((rules & 1) && s0 > 0);
    const bool beyond = s0 > len && !(rules & 16);
    if (rules & 8) {
      if (none) out_flag[i] = 0;
      else if (beyond) out_flag[i] = (rules & 4) ? 1 : 0;
    } else {
      if (none || (beyond && !(rules & 2))) { out_s[i] = -1; out_e[i] = -1; }
      else if (beyond) { out_s[i] = s0; out_e[i] = s0; }
      else if (out_s[i] >= 0) { out_s[i] += s0; out_e[i] += s0; }
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_max_len(const int64_t* __restrict__ offsets, int64_t n, int32_t* __restrict__ out) {
  int m = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
    m = max(m, (int)(offsets[t + 1] - offsets[t]));
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  __shared__ int wmax[kBlock / 64];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {   // one atomic per workgroup
    for (int w = 1; w < kBlock / 64; ++w) m = max(m, wmax[w]);
    if (m > 0) atomicMax(out, m);
  }
}

// ---- sub() -------------------------------------------------------------------------
struct SizeSink {
  int64_t n = 0;
  __device__ __forceinline__ void bytes(const uint8_t*, int k) { n += k; }
};
struct WriteSink {
  uint8_t* out;
  int64_t pos, cap;
  __device__ __forceinline__ void bytes(const uint8_t* src, int k) {
    if (pos + k > cap || k < 24) {   // clipped at the buffer's end, or short: byte by byte
      for (int j = 0; j < k; ++j)
        if (pos + j < cap) out[pos + j] = src[j];
      pos += k;
      return;
    }
    // A long piece (the text between two matches): aligned 8-byte stores; the source word for each is put
    // together from the two aligned words that hold its bytes (every word read holds at least one byte of the piece)
    uint8_t* d = out + pos;
    int j = 0;
    while ((uintptr_t)(d + j) & 7) { d[j] = src[j]; ++j; }
    const uintptr_t sa = (uintptr_t)(src + j);
    const uint64_t* sp = (const uint64_t*)(sa & ~(uintptr_t)7);
    const int sh = (int)(sa & 7) * 8;
    if (sh) {
      uint64_t lo = *sp;
      for (; j + 8 <= k; j += 8) {
        const uint64_t hi = *++sp;   // (holds the chunk's last sh / 8 bytes)
        *(uint64_t*)(d + j) = (lo >> sh) | (hi << (64 - sh));
        lo = hi;
      }
    } else {
      for (; j + 8 <= k; j += 8) *(uint64_t*)(d + j) = *sp++;
    }
    for (; j < k; ++j) d[j] = src[j];
    pos += k;
  }
};

// Inclusive scan over the G lanes of a group (16, 32 or 64: groups are aligned rows of 16 lanes) on the
// DPP path: row_shr 1 / 2 / 4 / 8 inside a row, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2
// and 3 -- one VALU instruction per step where __shfl_up costs a ds_bpermute round trip and a select.
template <int G, bool XOR>
__device__ __forceinline__ int group_scan(int x) {
#define MRX_DPP_STEP(CTRL, ROWS)                                                      \
  do {                                                                                \
    const int y = __builtin_amdgcn_update_dpp(0, x, CTRL, ROWS, 0xF, false);          \
    x = XOR ? (x ^ y) : (x + y);                                                      \
  } while (0)
  MRX_DPP_STEP(0x111, 0xF);   // row_shr:1
  MRX_DPP_STEP(0x112, 0xF);   // row_shr:2
  MRX_DPP_STEP(0x114, 0xF);   // row_shr:4
  MRX_DPP_STEP(0x118, 0xF);   // row_shr:8
  if (G >= 32) MRX_DPP_STEP(0x142, 0xA);   // row_bcast:15 -> rows 1, 3
  if (G >= 64) MRX_DPP_STEP(0x143, 0xC);   // row_bcast:31 -> rows 2, 3
#undef MRX_DPP_STEP
  return x;
}

// ---- sub() from findall spans (streamable plans) -----------------------------------------
// For plans whose findall runs on the streaming kernel, regex.sub is assembled from the CSR spans:
// the first `count` matches of a text (all when count == 0) are replaced, matches are never empty,
// and the replacement has a fixed length R (literal, or template over fixed-width groups), so
//   out position of replacement m:  rstart(m) = s_m - cum_m + m * R     (cum_m = matched bytes before m)
// k_subs_sizes: one lane per text -> output length and cum_m per span.
__global__ __launch_bounds__(kBlock) void k_subs_sizes(int64_t n, const int64_t* __restrict__ offsets,
                                                       const int64_t* __restrict__ prefix,
                                                       const int32_t* __restrict__ spans, long long count,
                                                       int R, int64_t* __restrict__ sizes,
                                                       int32_t* __restrict__ cum, int64_t span_cap,
                                                       const int32_t* __restrict__ left) {
  // 16 lanes per text: coalesced span loads, prefix sum of the match lengths inside the group
  if (prefix[n] > span_cap) return;   // the findall in front did not have room for its spans: the host retries
  if (left && *left == 0) return;     // cum[] is wanted for the texts k_subs_wave left over: there are none
  constexpr int G = 16;
  const int lane = threadIdx.x & (G - 1);
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / G);
  for (int64_t i = (int64_t)blockIdx.x * (blockDim.x / G) + (threadIdx.x / G); i < n; i += ngroups) {
    const int64_t a = prefix[i];
    int64_t k = prefix[i + 1] - a;
    if (count > 0 && k > count) k = count;
    int carry = 0;
    for (int64_t m0 = 0; m0 < k; m0 += G) {
      const int64_t m = m0 + lane;
      int len = 0;
      if (m < k) {
        const int2 sp = *(const int2*)(spans + 2 * (a + m));
        len = sp.y - sp.x;
      }
      int incl = len;
#pragma unroll
      for (int d = 1; d < G; d <<= 1) {
        const int v = __shfl_up(incl, d, G);
        if (lane >= d) incl += v;
      }
      if (m < k) cum[a + m] = carry + incl - len;
      carry += __shfl(incl, G - 1, G);
    }
    if (lane == 0 && sizes) sizes[i] = (offsets[i + 1] - offsets[i]) - carry + k * (int64_t)R;
  }
}

// k_subs_sizes_flat (count == 0: every match is replaced): output length of every text without a dependent
// round trip per text.  A wavefront owns 64 consecutive texts, whose spans are one contiguous range of the
// CSR; it streams that range 256 spans at a time, keeps the running sum of the match lengths, and lane t
// picks the sum's value at its text's first and one-past-last span as they pass (two cross-lane reads per
// 64 spans).  cum[] -- matched bytes before each match, which only k_subs_emit reads -- is not written here.
__global__ __launch_bounds__(kBlock) void k_subs_sizes_flat(int64_t n, const int64_t* __restrict__ offsets,
                                                            const int64_t* __restrict__ prefix,
                                                            const int32_t* __restrict__ spans, int R,
                                                            int64_t* __restrict__ sizes, int64_t span_cap) {
  if (prefix[n] > span_cap) return;   // as k_subs_sizes
  const int lane = threadIdx.x & 63;
  const int64_t nw = (n + 63) / 64;
  for (int64_t w = (int64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64; w < nw; w += (int64_t)gridDim.x * (kBlock / 64)) {
    const int64_t i = w * 64 + lane;
    const bool live = i < n;
    const int64_t pa = live ? prefix[i] : 0, pb = live ? prefix[i + 1] : 0;
    const int64_t tbytes = live ? offsets[i + 1] - offsets[i] : 0;
    const int64_t i_last = (w * 64 + 64 < n ? w * 64 + 64 : n);
    const int64_t A = __shfl(pa, 0), B = prefix[i_last];
    int run = 0;                 // matched bytes of spans [A, c), modulo 2^32 (differences are what is used)
    int at_a = 0, at_b = 0;      // the running sum in front of span pa / pb (both 0 while pa == A / pb == A)
    for (int64_t c = A; c < B; c += 256) {
      int len[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t m = c + 64 * j + lane;
        int2 sp = make_int2(0, 0);
        if (m < B) sp = *(const int2*)(spans + 2 * m);
        len[j] = sp.y - sp.x;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t c0 = c + 64 * j;
        const int incl = (int)((uint32_t)run + (uint32_t)group_scan<64, false>(len[j]));   // sum of spans [A, c0 + lane]
        // the sum in front of span x, x in (c0, c0 + 64]: lane x - 1 - c0 holds it
        const int64_t ja = pa - 1 - c0, jb = pb - 1 - c0;
        const int va = __shfl(incl, (int)(ja & 63)), vb = __shfl(incl, (int)(jb & 63));
        if (ja >= 0 && ja < 64) at_a = va;
        if (jb >= 0 && jb < 64) at_b = vb;
        run = __shfl(incl, 63);
      }
    }
    if (live) sizes[i] = tbytes - (int64_t)((uint32_t)at_b - (uint32_t)at_a) + (pb - pa) * (int64_t)R;
  }
}

// k_subs_emit: one wavefront per text, output centric.  Each lane produces one 16-byte block of
// the output (aligned on the output ADDRESS, so full blocks are single coalesced 16-byte stores),
// finds by binary search which replacement precedes its first byte and then walks: replacement
// bytes come from rmap (literal byte, or 0x8000 | offset into the match for a group byte), kept
// bytes from the input through an 8-byte register window.
constexpr int kSubsStage = 128;  // replacements per text staged in LDS (more: read from global)
constexpr int kSubsLanes = 16;   // lanes that share one text in k_subs_emit

// bytes src .. src+15 of a text as two little-endian u64 (bytes past the text are unspecified but
// never fetched from beyond the aligned word that holds the text's last byte)
__device__ __forceinline__ void load16(const uint8_t* base, int len, int src, uint64_t& lo, uint64_t& hi) {
  const uintptr_t addr = (uintptr_t)(base + src);
  const uint64_t* w = (const uint64_t*)(addr & ~(uintptr_t)7);
  const uint64_t* last = (const uint64_t*)(((uintptr_t)(base + len - 1)) & ~(uintptr_t)7);  // len > 0
  const uint64_t w0 = w <= last ? w[0] : 0, w1 = w + 1 <= last ? w[1] : 0, w2 = w + 2 <= last ? w[2] : 0;
  const int sh = (int)(addr & 7) * 8;
  if (sh == 0) { lo = w0; hi = w1; }
  else { lo = (w0 >> sh) | (w1 << (64 - sh)); hi = (w1 >> sh) | (w2 << (64 - sh)); }
}

// G = 16: 16 lanes share a text and its replacements are staged whole (up to kSubsStage, more: read
// from global).  G = kBlock (long texts): a whole workgroup shares a text, and every round -- G output
// blocks = 4 KiB of output -- stages the window of replacements that can touch it (the last one
// starting at or before the round's first byte, and the kSubsWindow - 1 after it).
constexpr int kSubsWindow = 512;
// texts k_subs_wave<G> takes (the rest is k_subs_emit's): frame and output fit the group's LDS tiles
__host__ __device__ inline bool subs_wave_takes(int G, int tlen, int mis, int olen, int head) {
  return tlen + mis <= 32 * G + 16 && olen + head <= 64 * G;
}
template <int G>
__global__ __launch_bounds__(kBlock) void k_subs_emit(int64_t n, const uint8_t* __restrict__ data,
                                                      const int64_t* __restrict__ offsets,
                                                      const int64_t* __restrict__ prefix,
                                                      const int32_t* __restrict__ spans,
                                                      const int32_t* __restrict__ cum, long long count,
                                                      int R, const uint16_t* __restrict__ rmap,
                                                      const int64_t* __restrict__ out_off,
                                                      uint8_t* __restrict__ out, int skip_g,
                                                      const int32_t* __restrict__ left) {
  // `left`: k_subs_wave<skip_g> in front took every text (skip_g != 0), or the host's go-ahead behind the totals
  // (k_subs_gate: the kernel is enqueued before the host has seen them) says no
  if (left && *left == 0) return;
  // kSubsLanes lanes share one text (4 texts per wavefront): a 1 KiB text has ~70 output blocks,
  // which 64 lanes cover in two half-empty rounds; 16 lanes cover them in five full ones, and the
  // four texts' dependent round trips (offsets -> spans -> bytes) overlap.
  constexpr bool WIN = G != kSubsLanes;
  constexpr int STAGE = WIN ? kSubsWindow : kSubsStage;
  static_assert(G == kSubsLanes || G == kBlock, "a group is 16 lanes or the workgroup");
  __shared__ int3 stage_all[kBlock / G][STAGE];  // {rstart, match start, match end}
  extern __shared__ __align__(16) uint8_t subs_dyn[];   // the replacement map (R u16 entries)
  uint16_t* rmap_lds = (uint16_t*)subs_dyn;
  for (int r = threadIdx.x; r < R; r += blockDim.x) rmap_lds[r] = rmap[r];
  __syncthreads();
  auto group_sync = [&]() {
    if (WIN) __syncthreads();
    else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  };
  const int lane = threadIdx.x & (G - 1);   // lane within the group that owns the text
  int3* stage = stage_all[threadIdx.x / G];
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x / G);
  for (int64_t i = (int64_t)blockIdx.x * (blockDim.x / G) + (threadIdx.x / G); i < n; i += ngroups) {
    const int64_t ibase = offsets[i];
    const uint8_t* tptr = data + ibase;
    const int tlen = (int)(offsets[i + 1] - ibase);
    const int64_t a = prefix[i];
    int64_t k64 = prefix[i + 1] - a;
    if (count > 0 && k64 > count) k64 = count;
    const int k = (int)k64;
    const int64_t obase = out_off[i];
    const int olen = (int)(out_off[i + 1] - obase);
    if (olen <= 0) continue;
    if (skip_g && subs_wave_takes(skip_g, tlen, (int)((uintptr_t)tptr & 15), olen, (int)((uintptr_t)(out + obase) & 15)))
      continue;   // k_subs_wave<skip_g> wrote this text
    const int32_t* sp = spans + 2 * a;
    const int32_t* cm = cum + a;
    const bool staged = !WIN && k <= STAGE;
    auto repl_global = [&](int m) {  // {rstart, match start, match end} of replacement m
      const int2 se = *(const int2*)(sp + 2 * m);
      return make_int3(se.x - cm[m] + m * R, se.x, se.y);
    };
    group_sync();
    if (staged) {
      for (int m = lane; m < k; m += G) stage[m] = repl_global(m);
      group_sync();
    }
    int wbase = 0, wcnt = 0;   // WIN: replacements [wbase, wbase + wcnt) are staged
    auto repl_at = [&](int m) {
      if (staged) return stage[m];
      if (WIN && m >= wbase && m < wbase + wcnt) return stage[m - wbase];
      return repl_global(m);
    };
    const int head = (int)((uintptr_t)(out + obase) & 15);  // output starts `head` bytes into its first 16-byte block
    for (int blk = 0; blk * 16 < head + olen; blk += G) {
      if (WIN) {
        group_sync();   // everyone is done with the previous window
        const int w_lo = blk * 16 - head < 0 ? 0 : blk * 16 - head;   // first output position of this round
        int lo = -1, hi = k;
        while (hi - lo > 1) {   // same addresses in every lane: one broadcast load per step
          const int mid = (lo + hi) >> 1;
          if (repl_global(mid).x <= w_lo) lo = mid; else hi = mid;
        }
        wbase = lo < 0 ? 0 : lo;
        wcnt = k - wbase < STAGE ? k - wbase : STAGE;
        for (int m = lane; m < wcnt; m += G) stage[m] = repl_global(wbase + m);
        group_sync();
      }
      const int p_lo = (blk + lane) * 16 - head;   // first output position of my block (may be < 0)
      int p = p_lo < 0 ? 0 : p_lo;
      const int p_hi = p_lo + 16 < olen ? p_lo + 16 : olen;
      if (p >= p_hi) continue;
      // j = last replacement with rstart(j) <= p, or -1
      int lo = -1, hi = k;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (repl_at(mid).x <= p) lo = mid; else hi = mid;
      }
      int j = lo;
      int3 cur = j >= 0 ? repl_at(j) : make_int3(0, 0, 0);
      int nxt = j + 1 < k ? repl_at(j + 1).x : 0x7FFFFFFF;
      uint64_t wlo = 0, whi = 0;   // the 16 output bytes
      // OR `cnt` (1..16) bytes of (vlo, vhi) into the block at byte offset q
      auto put = [&](uint64_t vlo, uint64_t vhi, int q, int cnt) {
        if (cnt < 8) { vlo &= (1ull << (cnt * 8)) - 1; vhi = 0; }
        else if (cnt < 16) { vhi &= (cnt == 8) ? 0ull : ((1ull << ((cnt - 8) * 8)) - 1); }
        if (q == 0) { wlo |= vlo; whi |= vhi; }
        else if (q < 8) { wlo |= vlo << (q * 8); whi |= (vhi << (q * 8)) | (vlo >> (64 - q * 8)); }
        else if (q == 8) { whi |= vlo; }
        else { whi |= vlo << ((q - 8) * 8); }
      };
      while (p < p_hi) {
        while (p == nxt) {   // several replacements can start here when R == 0 and matches touch
          ++j;
          cur = repl_at(j);
          nxt = j + 1 < k ? repl_at(j + 1).x : 0x7FFFFFFF;
        }
        if (j >= 0 && p < cur.x + R) {   // replacement bytes: literal, or group bytes of the match
          int stop = cur.x + R < p_hi ? cur.x + R : p_hi;
          if (nxt < stop) stop = nxt;
          uint64_t mlo, mhi;
          load16(tptr, tlen, cur.y, mlo, mhi);   // the first 16 bytes of the match
          for (; p < stop; ++p) {
            const uint32_t r = rmap_lds[p - cur.x];
            int b = (int)r;
            if (r & 0x8000u) {
              const int o = (int)(r & 0x7FFFu);
              b = o < 16 ? (int)(((o < 8 ? mlo : mhi) >> ((o & 7) * 8)) & 0xFFu) : (int)tptr[cur.y + o];
            }
            put((uint64_t)(uint32_t)b, 0, p - p_lo, 1);
          }
        } else {                          // kept input bytes up to the next replacement: one 16-byte fetch
          const int src = j >= 0 ? p - (cur.x + R) + cur.z : p;
          const int stop = nxt < p_hi ? nxt : p_hi;
          uint64_t vlo, vhi;
          load16(tptr, tlen, src, vlo, vhi);
          put(vlo, vhi, p - p_lo, stop - p);
          p = stop;
        }
      }
      uint8_t* dst = out + (obase + p_lo);
      if (p_lo >= 0 && p_lo + 16 <= olen) {
        *(uint4*)dst = make_uint4((uint32_t)wlo, (uint32_t)(wlo >> 32), (uint32_t)whi, (uint32_t)(whi >> 32));
      } else {  // block shared with a neighbouring text: byte stores only
        for (int q = (p_lo < 0 ? -p_lo : 0); q < p_hi - p_lo; ++q)
          dst[q] = (uint8_t)((q < 8 ? wlo : whi) >> ((q & 7) * 8));
      }
    }
    group_sync();
  }
}

// k_subs_wave<G>: G lanes (a wavefront, half or quarter of one) assemble one text's output in LDS without a
// search and without a per-byte branch.  Text positions are kept in the frame of the 16-byte blocks that
// hold the text (mis = address & 15), the output tile in the frame of the output address (head).
//   matches, one lane each:  the start and end position of every replaced match set a bit in two LDS
//     bitmaps; matched bytes before the match come from a prefix sum over the lanes' match lengths, so the
//     replacement's R bytes go straight to  head + start - matched_before + m R  in the output tile;
//   frame blocks, one lane each:  T = starts ^ ends of the block; "inside a match" at byte t is the parity
//     of the T bits at or below t (matches are non-empty and do not overlap; an end that is the next
//     match's start cancels), carried across lanes by a prefix sum that also gives the kept bytes and the
//     match starts before the block, i.e. where the block's first kept byte goes; the 16 bytes are then
//     placed with predicated byte writes;
//   output blocks, one lane each:  16-byte stores (byte stores where a block is shared with a neighbour).
// Texts whose frame or output exceeds the tiles (32 G + 16 / 64 G bytes) are left to k_subs_emit.
template <int G>
__global__ __launch_bounds__(kBlock) void k_subs_wave(int64_t n, const uint8_t* __restrict__ data,
                                                      const int64_t* __restrict__ offsets,
                                                      const int64_t* __restrict__ prefix,
                                                      const int32_t* __restrict__ spans, long long count,
                                                      int R, const uint16_t* __restrict__ rmap,
                                                      const int64_t* __restrict__ out_off,
                                                      uint8_t* __restrict__ out, int32_t* __restrict__ left,
                                                      const int32_t* __restrict__ go, int dbg) {
  if (*go == 0) return;   // k_subs_gate: the spans or the output do not fit (the host finds out behind this launch)
  constexpr int NG = kBlock / G, F = 32 * G + 16, O = 64 * G, BW = F / 32 + 1;
  static_assert(G == 256 || G == 64 || G == 32 || G == 16, "group = workgroup, wavefront, half or quarter of one");
  constexpr bool BLK = G > 64;   // the workgroup shares one text: barriers instead of wavefront order, prefix
                                 // sums carried across its four wavefronts through LDS
  static_assert(!BLK || G == kBlock, "a group beyond a wavefront is the whole workgroup");
  __shared__ __align__(16) uint8_t text_all[NG][F];
  __shared__ __align__(16) uint8_t out_all[NG][O];
  __shared__ uint32_t sbits_all[NG][BW], ebits_all[NG][BW];
  __shared__ uint32_t spare_all[kBlock];
  __shared__ int xw_all[3][BLK ? kBlock / 64 : 1];   // BLK: per-wavefront totals of the three prefix sums
  extern __shared__ __align__(16) uint8_t subs_dyn[];   // the replacement map (R u16 entries)
  uint16_t* rmap_lds = (uint16_t*)subs_dyn;
  for (int r = threadIdx.x; r < R + 8; r += blockDim.x) rmap_lds[r] = r < R ? rmap[r] : (uint16_t)0;
  __syncthreads();
  auto group_sync = [&]() {
    if (BLK) { __syncthreads(); return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // inclusive prefix over the group's lanes and the group's total (xw: one of the three LDS rows; a row is
  // used once per round, and the rounds are separated by the barriers of the other two)
  auto scan_add = [&](int x, int* xw, int& total) {
    if constexpr (!BLK) {
      const int incl = group_scan<BLK ? 64 : G, false>(x);
      total = __shfl(incl, G - 1, G);
      return incl;
    }
    int incl = group_scan<64, false>(x);
    const int wv = threadIdx.x >> 6;
    __syncthreads();   // the row's previous readers are done
    if ((threadIdx.x & 63) == 63) xw[wv] = incl;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) { const int v = xw[q]; tot += v; if (q < wv) before += v; }
    total = tot;
    return incl + before;
  };
  auto scan_xor = [&](int x, int* xw, int& total) {
    if constexpr (!BLK) {
      const int incl = group_scan<BLK ? 64 : G, true>(x);
      total = __shfl(incl, G - 1, G);
      return incl;
    }
    int incl = group_scan<64, true>(x);
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) xw[wv] = incl;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) { const int v = xw[q]; tot ^= v; if (q < wv) before ^= v; }
    total = tot;
    return incl ^ before;
  };
  const int lane = threadIdx.x & (G - 1), grp = threadIdx.x / G;
  uint8_t* text = text_all[grp];
  uint8_t* otile = out_all[grp];
  uint32_t* sbits = sbits_all[grp];
  uint32_t* ebits = ebits_all[grp];
  const int64_t ngroups = (int64_t)gridDim.x * NG;
  // Two texts ahead: the descriptor (offsets, prefix, out_off: one round trip); one text ahead: its frame
  // blocks and its first G spans, in registers -- the global round trips of text i + 1 run under the LDS
  // phases of text i.
  struct Desc { int64_t ibase, a, obase; int tlen, k, olen; };
  auto load_desc = [&](int64_t i) {
    Desc d{0, 0, 0, 0, 0, 0};
    if (i < n) {
      d.ibase = offsets[i];
      d.tlen = (int)(offsets[i + 1] - d.ibase);
      d.a = prefix[i];
      int64_t k64 = prefix[i + 1] - d.a;
      if (count > 0 && k64 > count) k64 = count;
      d.k = (int)k64;
      d.obase = out_off[i];
      d.olen = (int)(out_off[i + 1] - d.obase);
    }
    return d;
  };
  uint4 tx[3];
  int2 sp_first;
  auto issue = [&](const Desc& d) {
    const uint8_t* tptr = data + d.ibase;
    const int mis = (int)((uintptr_t)tptr & 15);
    const uint8_t* fptr = tptr - mis;
    int nfb = d.olen > 0 ? (mis + d.tlen + 15) >> 4 : 0;
    if (nfb > 2 * G + 1) nfb = 2 * G + 1;   // a longer text is not this kernel's
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int b = lane + r * G;
      tx[r] = b < nfb ? MRX_LDG((const uint4*)(fptr + 16 * b)) : make_uint4(0, 0, 0, 0);
    }
    sp_first = (d.olen > 0 && lane < d.k) ? *(const int2*)(spans + 2 * (d.a + lane)) : make_int2(0, 0);
  };
  int64_t i = (int64_t)blockIdx.x * NG + grp;
  Desc d0 = load_desc(i), d1 = load_desc(i + ngroups);
  issue(d0);
  for (; i < n; i += ngroups) {
    const Desc d = d0;
    d0 = d1;
    d1 = load_desc(i + 2 * ngroups);
    const uint8_t* tptr = data + d.ibase;
    const int tlen = d.tlen, k = d.k, olen = d.olen;
    const int64_t a = d.a, obase = d.obase;
    const int mis = (int)((uintptr_t)tptr & 15), head = (int)((uintptr_t)(out + obase) & 15);
    if (olen <= 0 || !subs_wave_takes(G, tlen, mis, olen, head)) {
      if (olen > 0 && lane == 0) *left = 1;   // k_subs_emit, launched behind this kernel, looks
      issue(d0);
      continue;
    }
    const int nfb = (mis + tlen + 15) >> 4;
    group_sync();   // the previous text's tiles are free
    for (int w = lane; w < BW; w += G) { sbits[w] = 0; ebits[w] = 0; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (lane + r * G < nfb) *(uint4*)(text + 16 * (lane + r * G)) = tx[r];
    const int2 sp_cur = sp_first;
    issue(d0);
    group_sync();
    // ---- matches: boundary bits, replacement bytes
    int carry = 0;
    for (int m0 = 0; m0 < ((dbg & 16) ? 0 : k); m0 += G) {
      const int m = m0 + lane;
      int ms = 0, me = 0;
      if (m < k) {
        const int2 sp = m0 == 0 ? sp_cur : *(const int2*)(spans + 2 * (a + m));
        ms = sp.x; me = sp.y;
      }
      const int len = me - ms;
      int round_total;
      const int incl = scan_add(len, xw_all[0], round_total);
      const int before = carry + incl - len;
      carry += round_total;
      if (m < k) {
        atomicOr(&sbits[(mis + ms) >> 5], 1u << ((mis + ms) & 31));
        atomicOr(&ebits[(mis + me) >> 5], 1u << ((mis + me) & 31));
        uint8_t* dst = otile + (head + ms - before + m * R);
        const uint8_t* msrc = text + mis + ms;
        // four bytes a round: the map entries (one address for every lane) and the match bytes they name are read
        // together -- a byte at a time the loop waits for two dependent LDS reads per replacement byte
        for (int t = 0; t < ((dbg & 1) ? 0 : R); t += 4) {
          uint32_t r[4], v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = rmap_lds[t + q];   // (the map is padded with eight zero entries)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = msrc[(r[q] & 0x8000u) ? (r[q] & 0x7FFFu) : 0u];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (t + q < R) dst[t + q] = (r[q] & 0x8000u) ? (uint8_t)v[q] : (uint8_t)r[q];
        }
      }
    }
    group_sync();
    // ---- frame blocks: kept bytes
    int kept_carry = 0, starts_carry = 0, inside_carry = 0;
    for (int b0 = 0; b0 < ((dbg & 8) ? 0 : nfb); b0 += G) {
      const int b = b0 + lane;
      uint32_t S = 0, T = 0, valid = 0;
      uint4 bx = make_uint4(0, 0, 0, 0);
      if (b < nfb) {
        const int sh = (b & 1) * 16;
        S = (sbits[b >> 1] >> sh) & 0xFFFFu;
        T = S ^ ((ebits[b >> 1] >> sh) & 0xFFFFu);
        const int lo = b == 0 ? mis : 0;
        const int hi = mis + tlen - 16 * b < 16 ? mis + tlen - 16 * b : 16;
        valid = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        bx = *(const uint4*)(text + 16 * b);
      }
      uint32_t P = T;   // P bit t = parity of the T bits at or below t
      P ^= P << 1; P ^= P << 2; P ^= P << 4; P ^= P << 8;
      P &= 0xFFFFu;
      const uint32_t keep0 = valid & ~P, keep1 = valid & P;   // kept bytes if the block begins outside / inside a match
      // kept bytes | match starts << 12 | parity << 24, for both entry states the counts differ: scan the
      // parity first (it decides which), then the counts
      const int par = (int)(P >> 15);
      int par_total;
      const int pin = scan_xor(par, xw_all[1], par_total);
      const int inside = inside_carry ^ pin ^ par;   // state on entry to my block
      inside_carry ^= par_total;
      const uint32_t keep = inside ? keep1 : keep0;
      const int cnt = __popc(keep) | (__popc(S) << 16);   // kept bytes (<= 16 G) | match starts
      int tot;
      const int cin = scan_add(cnt, xw_all[2], tot);
      const int ex = cin - cnt;
      int po = head + kept_carry + (ex & 0xFFFF) + (starts_carry + (ex >> 16)) * R;
      kept_carry += tot & 0xFFFF;
      starts_carry += tot >> 16;
      // 16 unconditional byte writes: a byte that is not kept goes to the lane's own spare word (a select costs
      // less than an execution-mask round trip per byte); a wavefront without any kept byte skips them
      if (__builtin_amdgcn_ballot_w64(keep != 0) == 0 || (dbg & 2)) continue;
      const uint32_t wv[4] = {bx.x, bx.y, bx.z, bx.w};
      uint8_t* const spare = (uint8_t*)&spare_all[threadIdx.x];
      const int Rs = S ? R : 0;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        po += (int)((S >> t) & 1u) * Rs;
        const uint32_t kb = (keep >> t) & 1u;
        uint8_t* const wp = kb ? otile + po : spare;
        *wp = (uint8_t)(wv[t >> 2] >> ((t & 3) * 8));
        po += (int)kb;
      }
    }
    group_sync();
    // ---- output blocks
    uint8_t* dst0 = out + obase - head;
    const int nob = (dbg & 4) ? 0 : (head + olen + 15) >> 4;
    for (int b = lane; b < nob; b += G) {
      if (16 * b >= head && 16 * b + 16 <= head + olen) {
        *(uint4*)(dst0 + 16 * b) = *(const uint4*)(otile + 16 * b);
      } else {   // block shared with a neighbouring text
        const int q1 = 16 * b + 16 < head + olen ? 16 * b + 16 : head + olen;
        for (int q = 16 * b < head ? head : 16 * b; q < q1; ++q) dst0[q] = otile[q];
      }
    }
  }
}

enum { SUB_SIZE = 0, SUB_EMIT = 1 };

template <int MODE, int BT = 0>
__global__ __launch_bounds__(kBlock) void k_sub(DevPlan p, const uint8_t* __restrict__ blob,
                                                Layout lay, int64_t n,
                                                const uint8_t* __restrict__ repl, int repl_len,
                                                int use_groups, const ReplSeg* __restrict__ tpl,
                                                int ntpl, long long count,
                                                int64_t* __restrict__ sizes,
                                                const int64_t* __restrict__ out_off,
                                                uint8_t* __restrict__ out, int64_t out_cap) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Ctx c = stage_tables(p, blob, lds);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const Text t = lay.text(i);
    if (MODE == SUB_SIZE) {
      SizeSink s;
      sub_text<BT>(c, t, repl, repl_len, use_groups, tpl, ntpl, count, s);
      sizes[i] = s.n;
    } else {
      WriteSink s{out, out_off[i], out_cap};
      sub_text<BT>(c, t, repl, repl_len, use_groups, tpl, ntpl, count, s);
    }
  }
}

// ---- exclusive scan (counts -> CSR offsets) ---------------------------------------
template <class T>
__global__ __launch_bounds__(kScanBlock) void k_scan_local(const T* __restrict__ in, int64_t n,
                                                           int64_t* __restrict__ out,
                                                           int64_t* __restrict__ block_sums) {
  __shared__ int64_t wsum[kScanBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int64_t v[kScanItems];
  int64_t local = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t idx = base + k;
    const int64_t x = idx < n ? (int64_t)in[idx] : 0;
    v[k] = local;
    local += x;
  }
  // wave inclusive scan of `local`
  int64_t inc = local;
  const int lane = threadIdx.x & 63;
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t y = __shfl_up(inc, off);
    if (lane >= off) inc += y;
  }
  const int wave = threadIdx.x >> 6;
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int64_t wave_off = 0;
  for (int w2 = 0; w2 < wave; ++w2) wave_off += wsum[w2];
  const int64_t excl = wave_off + inc - local;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t idx = base + k;
    if (idx < n) out[idx] = excl + v[k];
  }
  if (threadIdx.x == kScanBlock - 1) block_sums[blockIdx.x] = wave_off + inc;
}

__global__ void k_scan_blocks(int64_t* __restrict__ block_sums, int64_t nblocks,
                              int64_t* __restrict__ total) {
  // single workgroup, sequential over tiles of 256: nblocks <= n/2048
  __shared__ int64_t carry;
  __shared__ int64_t wsum[4];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nblocks; base += blockDim.x) {
    const int64_t idx = base + threadIdx.x;
    const int64_t x = idx < nblocks ? block_sums[idx] : 0;
    int64_t inc = x;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
      const int64_t y = __shfl_up(inc, off);
      if (lane >= off) inc += y;
    }
    const int wave = threadIdx.x >> 6;
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int64_t wave_off = carry;
    for (int w2 = 0; w2 < wave; ++w2) wave_off += wsum[w2];
    if (idx < nblocks) block_sums[idx] = wave_off + inc - x;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry = wave_off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_add(int64_t* __restrict__ out, int64_t n,
                                                         const int64_t* __restrict__ block_sums,
                                                         const int64_t* __restrict__ total) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  const int64_t add = block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t idx = base + k;
    if (idx < n) out[idx] += add;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

}  // namespace

// ============================================================================
// host side
// ============================================================================
// The compiled tables live once per device that has used the handle (uploaded on first use there,
// never freed or replaced before mrx_free): concurrent calls on one handle from threads bound to
// different GPUs each see their own device's copy.
constexpr int kMaxDevices = 64;
struct mrx_handle {
  HostPlan hp;
  std::atomic<uint8_t*> d_blobs[kMaxDevices];
  std::mutex mu;   // serialises the first upload per device
  std::string describe_cache;
  // sub: matches per KiB of input the last batch held (0 = nothing seen yet): sizes the span buffer of the next
  // call so that a dense batch (more than one match per eight bytes) does not scan twice every time
  mutable std::atomic<int64_t> sub_matches_per_kib{0};
  // findall of a required-byte plan on long texts: the wavefront-per-text kernel or pieces on the multi-walk kernel?
  // Neither candidate density nor batch shape predicts it (profiles/r04_suite_routes.md), so the first two calls of a
  // batch shape try each route (see ReqTune) and the faster one is kept.
  // Four measured calls per batch shape: each route once untimed (its scratch gets allocated), then each route timed --
  // two HIP events on the caller's stream around the call's work, read by a LATER call of the shape once the last pair
  // has completed: no call ever waits for the tuner, and callers that enqueue back to back are served as well.
  struct ReqTune {
    int issued = 0, choice = 0;   // calls measured so far (0..4); choice: 1 wavefront kernel, 2 pieces
    bool probed = false;
    float ms_wave = 0, ms_pieces = 0;
    hipEvent_t ev[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    int dev = 0;                  // device the events belong to (part of the key)
  };
  mutable std::mutex tune_mu;
  mutable std::map<uint32_t, ReqTune> req_tune;
  // findall on texts FULL of matches (config 5: a match every six bytes): event rows instead of records (ST_ROWS,
  // k_decode_rows).  How full a batch is, the handle's previous eligible call tells: its total travels to pinned host
  // memory behind the call's work and is looked at by the next call if it has arrived -- no call waits for it.
  struct DenseProbe {
    int64_t* host_total = nullptr;   // pinned
    hipEvent_t ev = nullptr;
    bool pending = false;
    int64_t bytes = 0;               // of the batch the pending total belongs to
    int dev = 0, dense = 0;          // dense: 1 = the last batch seen held at least one match per kDenseBytesPerSpan bytes
  };
  mutable DenseProbe dense_probe;    // (under tune_mu)
  mrx_handle() { for (auto& b : d_blobs) b.store(nullptr, std::memory_order_relaxed); }
};

namespace {

thread_local std::string g_err;
thread_local bool g_timing = false;
thread_local double g_scan_ms = 0.0;
thread_local int64_t g_scan_launches = 0;
thread_local const char* g_last_kernel = "";
// set while a search runs on the view behind each text's first prefilter-literal occurrence: the memchr
// prefilter of HybridMatcher.match_next has been consulted, the engine's own search is what is left
thread_local bool t_prefilter_done = false;
// mrx_debug_force_generic(): route every call to the generic lane-per-text kernels (tests compare
// the two implementations; never set in production)
// (process-wide switches of include/mrx_testing.h; relaxed atomics: set while no call is in flight)
std::atomic<int> g_pair_tables{1};      // MRX_NO_PAIR_TABLES=1 in the environment: measure the one-byte class table
std::atomic<int> g_long_text_mode{0};   // mrx_debug_long_text_kernels(): 0 by average length, 1 always, 2 never,
                                        // 3 = as 1 but stepper plans on the wavefront-per-text kernel instead of pieces
std::atomic<int> g_force_generic{0};   // 0 best kernel, 1 no streaming kernel, 2 literal restatement (mrx_device.hpp) only
std::atomic<int> g_litscan_pieces{2};   // mrx_debug_litscan_pieces(): 0 one lane per text, 1 always 208-byte pieces, 2 by batch shape

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

}  // namespace
namespace mrx { int internal_fail(int code, const std::string& msg) { return fail(code, msg); } }   // mrx_internal.hpp
namespace {

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(MRX_E_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// ---- per-call scratch ------------------------------------------------------------------
// Counts, records, block sums ... live only for one API call.  hipFreeAsync was measured to block
// the host until the stream reaches it (296 us behind the scan kernel), which serialises host and
// GPU; so scratch comes from a grow-only arena per (host thread, stream) instead.  Calls on one
// stream run in order, so the next call may reuse the bytes; the arena is rewound when the last
// allocation of a call is released and is only ever freed after synchronising its stream.
struct ScratchArena {
  struct Chunk { uint8_t* base; size_t cap, used; };
  std::vector<Chunk> chunks;
  int live = 0;
  int depth = 0;   // nested ScratchScopes (entry points call each other)
};
// keyed by (device, stream): the null stream of two devices must not share an arena
thread_local std::map<std::pair<int, hipStream_t>, ScratchArena> g_scratch;
static ScratchArena& scratch_arena(hipStream_t s) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return g_scratch[std::make_pair(dev, s)];
}

hipError_t scratch_alloc(void** out, size_t bytes, hipStream_t s) {
  ScratchArena& a = scratch_arena(s);
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  if (a.live == 0 && a.chunks.size() > 1) {  // grew during the previous call: one chunk from now on
    size_t total = 0;
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    for (auto& c : a.chunks) { total += c.cap; (void)hipFree(c.base); }
    a.chunks.clear();
    uint8_t* p = nullptr;
    e = hipMalloc((void**)&p, total);
    if (e != hipSuccess) return e;
    a.chunks.push_back({p, total, 0});
  }
  for (auto& c : a.chunks)
    if (c.cap - c.used >= bytes) {
      *out = c.base + c.used;
      c.used += bytes;
      ++a.live;
      return hipSuccess;
    }
  size_t cap = bytes;
  if (!a.chunks.empty() && a.chunks.back().cap / 2 > cap) cap = a.chunks.back().cap / 2;
  uint8_t* p = nullptr;
  const hipError_t e = hipMalloc((void**)&p, cap);
  if (e != hipSuccess) return e;
  a.chunks.push_back({p, cap, bytes});
  *out = p;
  ++a.live;
  return hipSuccess;
}
hipError_t scratch_free(void* p, hipStream_t s) {
  if (!p) return hipSuccess;
  ScratchArena& a = scratch_arena(s);
  if (a.live > 0 && --a.live == 0)
    for (auto& c : a.chunks) c.used = 0;
  return hipSuccess;
}
// One per API call (entry points nest: sub -> findall): whatever exit path the outermost call takes,
// bad-argument and HIP-error returns included, its allocations are returned to the arena, so the
// next call reuses the same bytes instead of growing the arena.
struct ScratchScope {
  ScratchArena& a;
  explicit ScratchScope(hipStream_t s) : a(scratch_arena(s)) { ++a.depth; }
  ~ScratchScope() {
    if (--a.depth == 0 && a.live != 0) {
      a.live = 0;
      for (auto& c : a.chunks) c.used = 0;
    }
  }
  ScratchScope(const ScratchScope&) = delete;
  ScratchScope& operator=(const ScratchScope&) = delete;
};
size_t scratch_bytes_reserved() {   // testing: total bytes held by the calling thread's arenas
  size_t t = 0;
  for (auto& kv : g_scratch) for (auto& c : kv.second.chunks) t += c.cap;
  return t;
}
void scratch_release_all() {
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto& kv : g_scratch) {
    (void)hipSetDevice(kv.first.first);
    (void)hipStreamSynchronize(kv.first.second);
    for (auto& c : kv.second.chunks) (void)hipFree(c.base);
  }
  (void)hipSetDevice(cur);
  g_scratch.clear();
}

// The calling thread's current device, set by ensure_device() at the start of every entry point;
// H_BLOB(h) is the handle's table copy on that device.
thread_local int t_dev = 0;
#define H_BLOB(h) ((h)->d_blobs[t_dev].load(std::memory_order_acquire))

int ensure_device(const mrx_handle* hc) {
  mrx_handle* h = const_cast<mrx_handle*>(hc);
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(MRX_E_NO_DEVICE, "device ordinal out of range");
  t_dev = dev;
  if (h->d_blobs[dev].load(std::memory_order_acquire)) return MRX_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->d_blobs[dev].load(std::memory_order_acquire)) return MRX_OK;
  uint8_t* p = nullptr;
  HIP_TRY(hipMalloc((void**)&p, h->hp.blob.size()));
  const hipError_t e = hipMemcpy(p, h->hp.blob.data(), h->hp.blob.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(p); return fail(MRX_E_NO_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
  h->d_blobs[dev].store(p, std::memory_order_release);
  return MRX_OK;
}

// workgroups that fill the current device: 8 per CU (CU count queried once per device)
int grid_cap() {
  static std::atomic<int> cus[kMaxDevices];
  int c = cus[t_dev].load(std::memory_order_relaxed);
  if (c <= 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, t_dev) != hipSuccess || v <= 0) v = 256;
    cus[t_dev].store(v, std::memory_order_relaxed);
    c = v;
  }
  return c * 8;
}

int grid_for(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  const int64_t cap = grid_cap();
  return (int)(g < cap ? g : cap);
}

// k_wstep for findall / count: plain route or required-byte route (a bool `use_req_route` in scope)
// (and a DevPlan `p` or handle `h` whose flags say whether the bitset form runs: `wstep_bits`)
// (and a bool `wstep_mwalk`: the plan's multi-walk form, k_mwalk, takes the plain route's place)
// k_mwalk for a plan of `kw` walk slots; pk: every text is shorter than 64 KiB (the packed-start form, modes that
// keep start registers)
std::atomic<int> g_mwalk_pk{1};   // MRX_NO_MWALK_PK=1: never the packed form (A/B, and a parity test compares the two)
bool mwalk_pk_ok(const Layout& lay, int64_t known_max) {
  static const bool off = getenv("MRX_NO_MWALK_PK") && getenv("MRX_NO_MWALK_PK")[0] == '1';
  if (off || !g_mwalk_pk || lay.vlen) return false;
  const int64_t m = lay.offsets ? known_max : (lay.lens ? lay.stride : (int64_t)lay.len);
  return m >= 0 && m <= 65535;
}
template <int MODE, class... Args>
void mwalk_launch(int kw, bool pk, dim3 g, dim3 b, size_t lds_bytes, hipStream_t s, Args... args) {
  if constexpr (MODE == STEP_SEARCH) {
    if (kw == -3) {   // (PF_MW_TRIES: match_next is the first report of the same walk)
      hipLaunchKernelGGL((k_mwalk<MODE, 2, 0, 3>), g, b, lds_bytes, s, args...);
      return;
    }
  }
  if constexpr (MODE == STEP_COUNT || MODE == STEP_EMIT) {
    if (kw == -2) {   // (PF_MW_EMPTY, walks that read beyond their match: build_emptywalk2())
      hipLaunchKernelGGL((k_mwalk<MODE, 2, 0, 2>), g, b, lds_bytes, s, args...);
      return;
    }
    if (kw == -3) {   // (PF_MW_TRIES: the same table form for a plan without empty matches)
      hipLaunchKernelGGL((k_mwalk<MODE, 2, 0, 3>), g, b, lds_bytes, s, args...);
      return;
    }
    if (kw < 0) {   // (DevPlan::mw_k == -1: the empty-match walk of a PF_MW_EMPTY plan; 0 is a table no walk ever enters)
      hipLaunchKernelGGL((k_mwalk<MODE, 2, 0, 1>), g, b, lds_bytes, s, args...);
      return;
    }
  }
  if constexpr (MODE == STEP_COUNT || MODE == STEP_ANY) {
    if (kw <= 2) hipLaunchKernelGGL((k_mwalk<MODE, 2>), g, b, lds_bytes, s, args...);
    else if (kw == 3) hipLaunchKernelGGL((k_mwalk<MODE, 3>), g, b, lds_bytes, s, args...);
    else hipLaunchKernelGGL((k_mwalk<MODE, 4>), g, b, lds_bytes, s, args...);
  } else {
    if (pk && kw <= 2) hipLaunchKernelGGL((k_mwalk<MODE, 2, 1>), g, b, lds_bytes, s, args...);
    else if (pk) hipLaunchKernelGGL((k_mwalk<MODE, 4, 1>), g, b, lds_bytes, s, args...);
    else if (kw <= 2) hipLaunchKernelGGL((k_mwalk<MODE, 2>), g, b, lds_bytes, s, args...);
    else if (kw == 3) hipLaunchKernelGGL((k_mwalk<MODE, 3>), g, b, lds_bytes, s, args...);
    else hipLaunchKernelGGL((k_mwalk<MODE, 4>), g, b, lds_bytes, s, args...);
  }
}
thread_local int64_t t_csr_max_len = -1;   // longest text of the CSR batch req_wave_pays() last looked at (-1: not known)
#define MRX_WSTEP_LAUNCH(MODE, ...)                                                        \
  do {                                                                                     \
    if (wstep_mwalk) mwalk_launch<MODE>(wstep_mwalk_k, wstep_mwalk_pk, __VA_ARGS__);         \
    else if (wstep_bm && wstep_bm_big) hipLaunchKernelGGL((k_wstep<MODE, 0, 0, 0, 1, 1>), __VA_ARGS__); \
    else if (wstep_bm) hipLaunchKernelGGL((k_wstep<MODE, 0, 0, 0, 1>), __VA_ARGS__);       \
    else if (wstep_bits) hipLaunchKernelGGL((k_wstep<MODE, 0, 1>), __VA_ARGS__);           \
    else if (wstep_empty) hipLaunchKernelGGL((k_wstep<MODE, 0, 0, 1>), __VA_ARGS__);       \
    else if (use_req_route) hipLaunchKernelGGL((k_wstep<MODE, 1>), __VA_ARGS__);         \
    else if (wstep_lz) hipLaunchKernelGGL((k_wstep<MODE, 0, 0, 0, 0, 0, 1>), __VA_ARGS__); \
    else hipLaunchKernelGGL((k_wstep<MODE, 0>), __VA_ARGS__);                              \
  } while (0)
// dynamic LDS of k_wstep for this plan
size_t wstep_lds(const DevPlan& p, bool mwalk = false, bool bm_big = false) {
  if (mwalk) return mwalk_table_bytes(p);
  if (bm_big) return 256 + (size_t)(p.nstates + 1) * p.ncls * 2 + 16;
  return (p.flags & PF_BSTEP) ? bstep_table_bytes(p.bs_npos) : wstep_table_bytes(p.nstates + ((p.flags & PF_LAZY_END) ? 1 : 0));
}
// PF_MWALK plans: several walks in one pass (k_mwalk) instead of the stepper's restart-per-position loop.
// mrx_debug_multiwalk(2) / MRX_NO_MWALK=1: never (A/B runs, and the parity tests compare the two text by text).
std::atomic<int> g_mwalk_mode{0};
bool mwalk_enabled() {
  static const bool off = getenv("MRX_NO_MWALK") && getenv("MRX_NO_MWALK")[0] == '1';
  return !off && g_mwalk_mode != 2 && g_force_generic == 0;   // (level 1 = the stepper and nothing newer)
}
bool mwalk_on(const DevPlan& p) { return (p.flags & PF_MWALK) && mwalk_enabled(); }
// MRX_TRIES_ALWAYS=1 / mrx_debug_tries_always(1): PF_MW_TRIES plans take the pending-tries walk without the route
// tuner's measurement (tests, A/B runs)
std::atomic<int> g_tries_always{(getenv("MRX_TRIES_ALWAYS") && getenv("MRX_TRIES_ALWAYS")[0] == '1') ? 1 : 0};
// PF_MW_TRIES plans on k_mwalk (MRX_NO_TRIES=1: their marks / stepper route instead -- A/B runs)
bool mw_tries_on(const DevPlan& p) {
  static const bool off = getenv("MRX_NO_TRIES") && getenv("MRX_NO_TRIES")[0] == '1';
  return (p.flags & PF_MW_TRIES) && !off && mwalk_enabled();
}
// PF_BACKSET plans: mark where matches begin (k_backscan), then the stepper only starts walks that succeed.
// MRX_NO_BACKSET=1 / mrx_debug_multiwalk(2): off.
bool backset_on(const DevPlan& p) {
  static const bool off = getenv("MRX_NO_BACKSET") && getenv("MRX_NO_BACKSET")[0] == '1';
  return (p.flags & PF_BACKSET) && !off && g_mwalk_mode != 2 && g_force_generic == 0;
}
// the plan as k_mwalk sees it on the required-byte route: its table in the place of the plain route's
DevPlan mwalk_req_plan(const DevPlan& p) {
  DevPlan q = p;
  q.off_mw_cls = p.off_mwr_cls; q.off_mw_tab = p.off_mwr_cls + 256;
  q.mw_ncfg = p.mwr_ncfg; q.mw_cshift = p.mwr_cshift; q.mw_bytes = p.mwr_bytes; q.mw_k = p.mwr_k;
  return q;
}
// Table plans of the stepper's plain route get the same first pass when their state sets fit 32 bits and the
// per-class follow tables fit LDS next to the text tiles (MRX_NO_UNION_PASS=1: off, for measurement).
// search: match_next always takes the plain route (PF_STEP_SEARCH); findall / count only without a required byte.
bool union_pass_for_table_plan(const DevPlan& p, bool search) {
  static const bool off = getenv("MRX_NO_UNION_PASS") && getenv("MRX_NO_UNION_PASS")[0] == '1';
  const bool plain = search ? (p.flags & PF_STEP_SEARCH) != 0 : ((p.flags & PF_STEPPABLE) && !(p.flags & PF_STEP_REQ));
  return !off && plain && !(p.flags & (PF_BSTEP | PF_STEP_BIG | PF_LAZY_END)) && p.nstates <= 32 &&
         bscan_dfa_table_bytes(p.nstates, p.ncls) <= 26 * 1024;
}
// Bitset NFA, first pass (k_bscan): on return *out is `lay` with every text cut to what the second pass
// has to look at (mode 0 search, 1 count / findall); *d_limit is scratch the caller frees.
// Right-to-left pass of a PF_BACKSET plan: on return *out is `lay` with the marks (bm, bm_cnt) attached (scratch of
// the calling scope).  A CSR batch's byte count is read back once to size the bitmap.
int backscan_marks(const mrx_handle* h, const Layout& lay, int64_t n, hipStream_t s, Layout* out) {
  const DevPlan& p = h->hp.dev;
  int64_t words;
  if (lay.offsets) {
    // rows are placed by the texts' byte offsets (Layout::bm_row), which never decrease: where the last text ends
    int64_t total = 0;
    if (lay.vlen) {   // views / pieces: offsets[n] need not exist
      int32_t last_len = 0;
      HIP_TRY(hipMemcpyAsync(&total, lay.offsets + (n - 1), sizeof total, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(&last_len, lay.vlen + (n - 1), sizeof last_len, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      total += last_len;
    } else {
      HIP_TRY(hipMemcpyAsync(&total, lay.offsets + n, sizeof total, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    words = 4 * ((total >> 7) + 2 * n + 2);
  } else {
    words = 4 * (n * ((lay.stride >> 7) + 2) + 2);
  }
  uint32_t* d_bm = nullptr;
  int32_t* d_cnt = nullptr;
  HIP_TRY(scratch_alloc((void**)&d_bm, sizeof(uint32_t) * (size_t)words, s));
  HIP_TRY(scratch_alloc((void**)&d_cnt, sizeof(int32_t) * (n > 0 ? n : 1), s));
  const int64_t nw = (n + 63) / 64;
  int64_t g = (nw + kWsWaves - 1) / kWsWaves;
  if (g < 1) g = 1;
  if (g > grid_cap()) g = grid_cap();
  hipLaunchKernelGGL(k_backscan, dim3((unsigned)g), dim3(64 * kWsWaves), (size_t)p.bk_bytes + 16, s, p, H_BLOB(h), lay, n, d_bm, d_cnt);
  HIP_TRY(hipGetLastError());
  *out = lay;
  out->bm = d_bm;
  out->bm_cnt = d_cnt;
  return MRX_OK;
}

int bscan_limits(const mrx_handle* h, const Layout& lay, int64_t n, int mode, hipStream_t s, Layout* out,
                 int32_t** d_limit) {
  const DevPlan& p = h->hp.dev;
  HIP_TRY(scratch_alloc((void**)d_limit, sizeof(int32_t) * (n > 0 ? n : 1), s));
  const int64_t nw = (n + 63) / 64;
  int64_t g = (nw + kWsWaves - 1) / kWsWaves;
  if (g < 1) g = 1;
  if (g > grid_cap()) g = grid_cap();
  if (!(p.flags & PF_BSTEP))   // a table plan: sets of DFA states
    hipLaunchKernelGGL((k_bscan<1, 1>), dim3((unsigned)g), dim3(64 * kWsWaves), bscan_dfa_table_bytes(p.nstates, p.ncls), s, p,
                       H_BLOB(h), lay, n, mode, *d_limit);
  else if (p.bs_npos <= 11)
    hipLaunchKernelGGL((k_bscan<1, 0, 1>), dim3((unsigned)g), dim3(64 * kWsWaves), bscan_table_bytes(p.bs_npos), s, p, H_BLOB(h),
                       lay, n, mode, *d_limit);
  else if (p.bs_npos <= 22)
    hipLaunchKernelGGL((k_bscan<1, 0, 2>), dim3((unsigned)g), dim3(64 * kWsWaves), bscan_table_bytes(p.bs_npos), s, p, H_BLOB(h),
                       lay, n, mode, *d_limit);
  else if (p.bs_npos <= 32)
    hipLaunchKernelGGL((k_bscan<1, 0, 3>), dim3((unsigned)g), dim3(64 * kWsWaves), bscan_table_bytes(p.bs_npos), s, p, H_BLOB(h),
                       lay, n, mode, *d_limit);
  else
    hipLaunchKernelGGL((k_bscan<0>), dim3((unsigned)g), dim3(64 * kWsWaves), bscan_table_bytes(p.bs_npos), s, p, H_BLOB(h),
                       lay, n, mode, *d_limit);
  HIP_TRY(hipGetLastError());
  *out = lay;
  if (lay.offsets) out->vlen = *d_limit; else out->lens = *d_limit;
  return MRX_OK;
}

// Bitset program whose matches all have one length: the union pass answers count (mode 2), findall's emit (3:
// counts = mode 2's, prefix = their prefix sums) and search (4: a = start, b = end) by itself.
bool bits_fixed_on(const DevPlan& p) {
  return (p.flags & PF_BSTEP) && p.bs_fixed_len > 0 && p.bs_nw == 1 && g_force_generic == 0 && mwalk_enabled();
}
int bscan_fixed(const mrx_handle* h, const Layout& lay, int64_t n, int mode, hipStream_t s, int32_t* a, int32_t* b,
                const int64_t* prefix, int64_t span_cap) {
  const DevPlan& p = h->hp.dev;
  const int64_t nw = (n + 63) / 64;
  int64_t g = (nw + kWsWaves - 1) / kWsWaves;
  if (g < 1) g = 1;
  if (g > grid_cap()) g = grid_cap();
  static const int ch = getenv("MRX_BSCAN_CH") ? atoi(getenv("MRX_BSCAN_CH")) : 64;
#define MRX_BSF(NCH_)                                                                                                       \
  do {                                                                                                                      \
    if (ch == 64 && mode != 5)   /* (the slot-row form measured better on the 128-byte tile: 0.95 against 1.07 ms) */       \
      hipLaunchKernelGGL((k_bscan<1, 0, NCH_, 64>), dim3((unsigned)g), dim3(64 * kWsWaves), bscan_table_bytes(p.bs_npos), s, p, \
                         H_BLOB(h), lay, n, mode, a, b, prefix, span_cap);                                                  \
    else                                                                                                                    \
      hipLaunchKernelGGL((k_bscan<1, 0, NCH_>), dim3((unsigned)g), dim3(64 * kWsWaves), bscan_table_bytes(p.bs_npos), s, p,   \
                         H_BLOB(h), lay, n, mode, a, b, prefix, span_cap);                                                  \
  } while (0)
  if (p.bs_npos <= 11) MRX_BSF(1);
  else if (p.bs_npos <= 22) MRX_BSF(2);
  else if (p.bs_npos <= 32) MRX_BSF(3);
#undef MRX_BSF
  else
    hipLaunchKernelGGL((k_bscan<0>), dim3((unsigned)g), dim3(64 * kWsWaves), bscan_table_bytes(p.bs_npos), s, p, H_BLOB(h),
                       lay, n, mode, a, b, prefix, span_cap);
  HIP_TRY(hipGetLastError());
  return MRX_OK;
}

// The stepper's routes: one wavefront per text (k_req_wave) when the texts are long or too few to
// fill the device with one lane each, one lane per text (k_wstep) otherwise.  The average length
// decides; a CSR batch's byte count lives on the device, so that costs one 8-byte read-back.
// big: the plan's table only exists in the wavefront kernel's class-indexed form (PF_STEP_BIG); the
// alternative is the literal restatement (50 GB/s), so the wavefront form takes every batch of texts of
// half a KiB and more
int req_wave_pays(const Layout& lay, int64_t n, bool req_route, hipStream_t s, bool* out, int* split = nullptr,
                  bool big = false, bool mwalk = false, bool marks_big = false) {
  *out = false;
  if (split) *split = 0;
  t_csr_max_len = -1;
  if (g_long_text_mode) { *out = g_long_text_mode == 1 || g_long_text_mode == 3; return MRX_OK; }
  if (n <= 0) return MRX_OK;
  int64_t total = 0, max_len = 0;
  if (lay.offsets) {
    int32_t* d_max = nullptr;
    int32_t m = 0;
    HIP_TRY(scratch_alloc((void**)&d_max, sizeof(int32_t), s));
    HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_max_len, dim3(grid_for(n, kBlock * 8)), dim3(kBlock), 0, s, lay.offsets, n, d_max);
    HIP_TRY(hipMemcpyAsync(&m, d_max, sizeof m, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&total, lay.offsets + n, sizeof total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(scratch_free(d_max, s));
    max_len = m;
    if (!lay.vlen) t_csr_max_len = m;
  } else {
    total = n * (lay.lens ? lay.stride : (int64_t)lay.len);
  }
  const int64_t avg = total / n;
  // Measured on the reference's benchmark texts (tools/bench_suite.py): with sparse candidates (the
  // required byte) the wavefront form wins from 2 KiB per text whatever the batch size; with dense
  // candidates (every byte a walk may start on) it does several times the work of the serial loop and
  // only pays while one lane per text would leave most of the device idle.
  *out = req_route ? (avg >= 2048 || (avg >= 512 && n <= 32768)) : (avg >= 1024 && n <= 65536);
  if (big && avg >= 512) *out = true;
  // plans with a multi-walk table: a lane scans its text once whatever the text holds, so from 32 Ki texts on one
  // lane per text beats the wavefront kernel's per-candidate walks (measured on the reference's list:
  // range_quantifiers 474 -> 853 GB/s, dual_quantifiers 218 -> 156: profiles/r03_multiwalk.md)
  if (mwalk && !req_route && !big && n >= 32768) *out = false;   // (required-byte route: sparse hits stay the wavefront kernel's best case)
  // big tables with a backward table: marks + one lane per text (class-indexed walk) once the batch has the lanes
  if (marks_big && big && !req_route && n >= 32768) *out = false;
  // a ragged batch with a few texts far longer than the rest: those go to the wavefront kernel, the
  // others keep one lane each
  if (!*out && split && max_len >= 32768 && max_len >= 8 * avg) *split = 16384;
  return MRX_OK;
}
int reqwave_grid(int64_t n) {
  int64_t g = (n + kRqWaves - 1) / kRqWaves;
  if (g < 1) g = 1;
  return (int)(g < grid_cap() ? g : grid_cap());
}
// (a bool `use_req_route` in scope, as for MRX_WSTEP_LAUNCH)
#define MRX_REQWAVE_LAUNCH(MODE, H, LAY, N, COUNTS, PREFIX, SPANS, CAP, S)                                       \
  do {                                                                                                           \
    if (use_req_route)                                                                                           \
      hipLaunchKernelGGL((k_req_wave<MODE, 1>), dim3(reqwave_grid(N)), dim3(64 * kRqWaves),                      \
                         reqwave_table_bytes((H)->hp.dev.nstates), S, (H)->hp.dev, H_BLOB(H), LAY, N, COUNTS,  \
                         PREFIX, SPANS, CAP, (int32_t*)nullptr, (int32_t*)nullptr);                              \
    else if ((H)->hp.dev.flags & PF_STEP_BIG)                                                                    \
      hipLaunchKernelGGL((k_req_wave<MODE, 0, 1>), dim3(reqwave_grid(N)), dim3(64 * kRqWaves),                   \
                         reqwave_big_bytes((H)->hp.dev.nstates, (H)->hp.dev.ncls), S, (H)->hp.dev, H_BLOB(H),  \
                         LAY, N, COUNTS, PREFIX, SPANS, CAP, (int32_t*)nullptr, (int32_t*)nullptr);              \
    else                                                                                                         \
      hipLaunchKernelGGL((k_req_wave<MODE, 0>), dim3(reqwave_grid(N)), dim3(64 * kRqWaves),                      \
                         reqwave_table_bytes((H)->hp.dev.nstates), S, (H)->hp.dev, H_BLOB(H), LAY, N, COUNTS,  \
                         PREFIX, SPANS, CAP, (int32_t*)nullptr, (int32_t*)nullptr);                              \
  } while (0)

int wstep_grid(int64_t n) {
  const int64_t nw = (n + 63) / 64;
  int64_t g = (nw + kWsWaves - 1) / kWsWaves;
  if (g < 1) g = 1;
  return (int)(g < grid_cap() ? g : grid_cap());
}

struct ScanTimer {  // HIP events around the dominant scan kernel, on its own stream
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s;
  bool on;
  explicit ScanTimer(hipStream_t st) : s(st), on(g_timing) {
    if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
  }
  void stop() {
    if (!on) return;
    (void)hipEventRecord(b, s);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    g_scan_ms += ms;
    g_scan_launches += 1;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    on = false;
  }
};

int check_search_supported(const mrx_handle* h) {
  if (!h->hp.why_no_search.empty()) return fail(MRX_E_UNSUPPORTED, h->hp.why_no_search);
  return MRX_OK;
}

// exclusive scan of n counts into prefix[n+1]; *d_total receives the sum
template <class T>
int device_scan(const T* d_in, int64_t n, int64_t* d_prefix, int64_t* d_total, hipStream_t s) {
  const int64_t nblocks = n > 0 ? (n + kScanTile - 1) / kScanTile : 1;
  int64_t* d_bs = nullptr;
  HIP_TRY(scratch_alloc((void**)&d_bs, sizeof(int64_t) * nblocks, s));
  hipLaunchKernelGGL(k_scan_local<T>, dim3((unsigned)nblocks), dim3(kScanBlock), 0, s, d_in, n,
                     d_prefix, d_bs);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, s, d_bs, nblocks, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nblocks), dim3(kScanBlock), 0, s, d_prefix, n,
                     d_bs, d_total);
  HIP_TRY(hipGetLastError());
  HIP_TRY(scratch_free(d_bs, s));
  return MRX_OK;
}

size_t lds_for(const mrx_handle* h) { return (size_t)h->hp.dev.blob_bytes; }

int check_lds(const mrx_handle* h) {
  if (h->hp.dev.blob_bytes > 60 * 1024)
    return fail(MRX_E_UNSUPPORTED, "compiled tables exceed the 60 KiB LDS staging budget");
  return MRX_OK;
}

// plans with a backtracker route take the kernel instantiations that carry its interpreter (k_match<., true> ...)
static bool plan_uses_backtracker(const mrx_handle* h) {
  return (h->hp.dev.flags & (PF_BT_FIRST | PF_BT_SEARCH)) != 0 && h->hp.dev.bt_nitems > 0;
}
// ... 2: the lean instantiation for deterministic chains (DevPlan::bt_flags bit 5: no choice stack), 1: the full one
static int bt_kernel_kind(const mrx_handle* h, bool wanted) {
  static const bool full_only = getenv("MRX_BT_FULL") && getenv("MRX_BT_FULL")[0] == '1';   // A/B: chains on the full interpreter
  if (!wanted) return 0;
  return ((h->hp.dev.bt_flags & 32) && !full_only) ? 2 : 1;
}
#define MRX_BT_DISPATCH(KIND, LAUNCH)        \
  do {                                       \
    if ((KIND) == 2) { LAUNCH(2); }          \
    else if ((KIND) == 1) { LAUNCH(1); }     \
    else { LAUNCH(0); }                      \
  } while (0)
// NFAEngine's literal prefilter (nfa.mojo:86-143, 391-498, 169-340) in front of the lane-per-text kernels:
// every backtracker-routed search starts with String.find(literal) -- and, on the '.*' fast paths, with a
// look for a newline and String.rfind(literal) -- over the WHOLE text, per lane and byte by byte in the
// literal restatement.  k_litscan answers all three for the whole batch in one coalesced pass (shift-and on
// the LDS text tile); the lanes then read their text's answers (Layout::pre -> Text::pre_*): a text without
// the literal is done at once, a '.*' pattern on a text without newlines needs no byte of it.
// mrx_debug_force_generic(2) runs without it (the parity tests compare the two).
int bt_prepass(const mrx_handle* h, const Layout& lay, int64_t n, hipStream_t s, Layout* out) {
  *out = lay;
  const DevPlan& p = h->hp.dev;
  if (g_force_generic == 2 || n <= 0 || p.bt_nitems <= 0 || !(p.bt_flags & 1) || p.bt_lit_len < 1 || p.bt_lit_len > 32)
    return MRX_OK;
  // HybridMatcher's own memchr prefilter (matcher.mojo:784-796) reads the same answers: only when its literal
  // is this one
  if ((p.flags & PF_PREFILTER) && h->hp.prefilter_literal != h->hp.nfa_literal) return MRX_OK;
  int2* d_pre = nullptr;
  HIP_TRY(scratch_alloc((void**)&d_pre, sizeof(int2) * n, s));
  const int64_t nw = (n + 63) / 64;
  int64_t g = (nw + kWsWaves - 1) / kWsWaves;
  if (g > grid_cap()) g = grid_cap();
  // few texts: one lane per text cannot fill the device, and long ones (4549 texts of 59 KB: 71 wavefronts) take
  // as long as one lane needs for its text.  Then the pass runs over 2 KiB pieces instead; their number stays on
  // the device.  (The early exit of the first-occurrence-only form is worth more than that, so only FULL.)
  const int64_t one_len = !lay.offsets && !lay.lens ? (int64_t)lay.len : -1;   // CSR / ragged: unknown here
  const bool in_pieces = (p.bt_flags & (4 | 8)) && g_litscan_pieces != 0 &&
                         (g_litscan_pieces == 1 || (n <= 32768 && (one_len < 0 || one_len > 4096)));
  if (in_pieces) {
    const int C = g_litscan_pieces == 1 ? 208 : 2048;   // tests: cuts that are not multiples of 16
    int32_t* d_cnt = nullptr;
    int64_t* d_vfirst = nullptr;
    int4* d_acc = nullptr;
    int64_t* d_tot = nullptr;
    HIP_TRY(scratch_alloc((void**)&d_tot, sizeof(int64_t), s));
    HIP_TRY(scratch_alloc((void**)&d_cnt, sizeof(int32_t) * n, s));
    HIP_TRY(scratch_alloc((void**)&d_vfirst, sizeof(int64_t) * (n + 1), s));
    HIP_TRY(scratch_alloc((void**)&d_acc, sizeof(int4) * n, s));
    hipLaunchKernelGGL(k_litscan_pieces, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, lay, n, C, d_cnt, d_acc);
    if (int rc = device_scan<int32_t>(d_cnt, n, d_vfirst, d_tot, s)) return rc;
    hipLaunchKernelGGL((k_litscan<true, true>), dim3((unsigned)grid_cap()), dim3(64 * kWsWaves), 0, s, H_BLOB(h) + p.off_bt_lit,
                       p.bt_lit_len, H_BLOB(h), lay, n, (int32_t*)nullptr, (int2*)nullptr, d_vfirst, C, d_acc);
    hipLaunchKernelGGL(k_litscan_join, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, d_acc, d_pre);
    HIP_TRY(scratch_free(d_cnt, s));
    HIP_TRY(scratch_free(d_vfirst, s));
    HIP_TRY(scratch_free(d_acc, s));
    HIP_TRY(scratch_free(d_tot, s));
  } else if (p.bt_flags & (4 | 8))
    hipLaunchKernelGGL(k_litscan<true>, dim3((unsigned)g), dim3(64 * kWsWaves), 0, s, H_BLOB(h) + p.off_bt_lit, p.bt_lit_len,
                       H_BLOB(h), lay, n, (int32_t*)nullptr, d_pre);
  else
    hipLaunchKernelGGL(k_litscan<false>, dim3((unsigned)g), dim3(64 * kWsWaves), 0, s, H_BLOB(h) + p.off_bt_lit, p.bt_lit_len,
                       H_BLOB(h), lay, n, (int32_t*)nullptr, d_pre);
  HIP_TRY(hipGetLastError());
  out->pre = d_pre;
  return MRX_OK;
}

template <int OP>
int run_match(const mrx_handle* h, const Layout& lay, int64_t n, int32_t* d_s, int32_t* d_e,
              uint8_t* d_flag, void* stream) {
  ScratchScope scratch_scope_((hipStream_t)stream);
  if (!h) return fail(MRX_E_ARGUMENT, "null handle");
  if (n < 0) return fail(MRX_E_ARGUMENT, "negative batch size");
  if (OP == OP_MATCH_FIRST || OP == OP_IS_MATCH) {
    if (!h->hp.why_no_match_first.empty()) return fail(MRX_E_UNSUPPORTED, h->hp.why_no_match_first);
    if (h->hp.first_onepass)  // only the streaming kernel carries the OnePass tables
      return fail(MRX_E_UNSUPPORTED, "internal: OnePass plans have no generic kernel");
  } else if (OP == OP_CAPTURES && h->hp.fixed_total < 0) {
    // general groups: NFAEngine.match_next_with_groups on the flat program, whatever engine searches
    if (!h->hp.bt.ok)
      return fail(MRX_E_UNSUPPORTED,
                  "capture groups of this pattern need the reference's recursive backtracking matcher "
                  "(nfa.mojo:500-574, 1057-1156); its flat-program form does not cover: " +
                      (h->hp.bt.why_not.empty() ? std::string("'.*'") : h->hp.bt.why_not));
  } else {
    if (int rc = check_search_supported(h)) return rc;
  }
  if (int rc = check_lds(h)) return rc;
  if (int rc = ensure_device(h)) return rc;
  if (n == 0) return MRX_OK;
  hipStream_t s = (hipStream_t)stream;
  ScanTimer tm(s);
  if (OP == OP_SEARCH && g_force_generic < 2 && (h->hp.dev.flags & PF_STEP_SEARCH) &&
      (!(h->hp.dev.flags & PF_PREFILTER) || t_prefilter_done)) {   // (the memchr prefilter changes match_next, matcher.mojo:784-796)
    bool wave = false;
    const bool big = (h->hp.dev.flags & PF_STEP_BIG) != 0;   // only the wavefront kernel has its table form
    const bool bits = (h->hp.dev.flags & PF_BSTEP) != 0;     // bitset NFA: the lane-per-text stepper only
    int split = 0;
    const bool lz = (h->hp.dev.flags & PF_LAZY_END) != 0;   // '$' on the LazyDFA search: one lane per text (the cache is the text's)
    if (!bits && !lz)
      if (int rc = req_wave_pays(lay, n, false, s, &wave, big ? nullptr : &split, false, mwalk_on(h->hp.dev) || mw_tries_on(h->hp.dev))) return rc;   // (not the `big` rule: search stops at the first match)
    Layout lay2 = lay;
    lay2.split = split;
    if (bits && bits_fixed_on(h->hp.dev)) {   // one match length: the first match end of the union pass is the answer
      if (int rc = bscan_fixed(h, lay, n, 4, s, d_s, d_e, nullptr, 0)) return rc;
      g_last_kernel = "k_bscan_fixed_search";
    } else
    if (bits) {   // union automaton first: texts without any match end are not searched at all
      int32_t* d_limit = nullptr;
      if (int rc = bscan_limits(h, lay, n, 0, s, &lay2, &d_limit)) return rc;
      hipLaunchKernelGGL((k_wstep<STEP_SEARCH, 0, 1>), dim3(wstep_grid(n)), dim3(64 * kWsWaves), wstep_lds(h->hp.dev), s,
                         h->hp.dev, H_BLOB(h), lay2, n, (int32_t*)nullptr, (const int64_t*)nullptr, (int32_t*)nullptr,
                         (int64_t)0, d_s, d_e);
      HIP_TRY(scratch_free(d_limit, s));
      g_last_kernel = "k_bstep_search";
    } else
    if (!wave && !big && (mwalk_on(h->hp.dev) || mw_tries_on(h->hp.dev))) {   // several walks in one pass; the few very long texts keep their kernel
      // (measured, profiles/r03_multiwalk.md: an "is there a match" pass in front -- STEP_ANY, no start registers,
      // 3 TB/s -- doubles the rate on texts without a match and halves it where the first match lies deep in
      // the text; one pass at 0.9-1.3 TB/s whatever the text holds is the default)
      mwalk_launch<STEP_SEARCH>(h->hp.dev.mw_k, mwalk_pk_ok(lay, t_csr_max_len), dim3(wstep_grid(n)), dim3(64 * kWsWaves),
                                mwalk_table_bytes(h->hp.dev), s, h->hp.dev, H_BLOB(h), lay2, n, (int32_t*)nullptr,
                                (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0, d_s, d_e);
      g_last_kernel = "k_mwalk_search";
      if (split > 0) {
        hipLaunchKernelGGL((k_req_wave<STEP_SEARCH, 0>), dim3(reqwave_grid(n)), dim3(64 * kRqWaves),
                           reqwave_table_bytes(h->hp.dev.nstates), s, h->hp.dev, H_BLOB(h), lay2, n, (int32_t*)nullptr,
                           (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0, d_s, d_e);
        g_last_kernel = "k_mwalk_search+k_req_wave_search";
      }
    } else
    if (big && !wave) {   // many short texts: the literal restatement, one lane per text
      hipLaunchKernelGGL((k_match<OP, 0>), dim3(grid_for(n, kBlock)), dim3(kBlock), lds_for(h), s, h->hp.dev,
                         H_BLOB(h), lay, n, d_s, d_e, d_flag);   // (a stepper plan: no backtracker route)
      g_last_kernel = "k_match";
    } else if (big) {
      hipLaunchKernelGGL((k_req_wave<STEP_SEARCH, 0, 1>), dim3(reqwave_grid(n)), dim3(64 * kRqWaves),
                         reqwave_big_bytes(h->hp.dev.nstates, h->hp.dev.ncls), s, h->hp.dev, H_BLOB(h), lay, n,
                         (int32_t*)nullptr, (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0, d_s, d_e);
      g_last_kernel = "k_req_wave_search";
    } else if (wave) {
      hipLaunchKernelGGL((k_req_wave<STEP_SEARCH, 0>), dim3(reqwave_grid(n)), dim3(64 * kRqWaves),
                         reqwave_table_bytes(h->hp.dev.nstates), s, h->hp.dev, H_BLOB(h), lay, n, (int32_t*)nullptr,
                         (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0, d_s, d_e);
      g_last_kernel = "k_req_wave_search";
    } else {
    int32_t* d_limit = nullptr;
    if (split == 0 && union_pass_for_table_plan(h->hp.dev, true))   // texts in which no walk from any start reaches an accepting state are not searched
      if (int rc = bscan_limits(h, lay, n, 0, s, &lay2, &d_limit)) return rc;
    if (lz)
      hipLaunchKernelGGL((k_wstep<STEP_SEARCH, 0, 0, 0, 0, 0, 1>), dim3(wstep_grid(n)), dim3(64 * kWsWaves), wstep_lds(h->hp.dev), s,
                         h->hp.dev, H_BLOB(h), lay2, n, (int32_t*)nullptr, (const int64_t*)nullptr, (int32_t*)nullptr,
                         (int64_t)0, d_s, d_e);
    else
    hipLaunchKernelGGL((k_wstep<STEP_SEARCH, 0>), dim3(wstep_grid(n)), dim3(64 * kWsWaves), wstep_lds(h->hp.dev), s, h->hp.dev,
                       H_BLOB(h), lay2, n, (int32_t*)nullptr, (const int64_t*)nullptr, (int32_t*)nullptr,
                       (int64_t)0, d_s, d_e);
    if (d_limit) HIP_TRY(scratch_free(d_limit, s));
    g_last_kernel = "k_step_search";
    if (split > 0) {   // the few very long texts of the batch
      hipLaunchKernelGGL((k_req_wave<STEP_SEARCH, 0>), dim3(reqwave_grid(n)), dim3(64 * kRqWaves),
                         reqwave_table_bytes(h->hp.dev.nstates), s, h->hp.dev, H_BLOB(h), lay2, n, (int32_t*)nullptr,
                         (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0, d_s, d_e);
      g_last_kernel = "k_step_search+k_req_wave_search";
    }
    }
  } else {
    Layout layp = lay;
    if ((OP == OP_SEARCH && (h->hp.dev.flags & PF_BT_SEARCH)) || (OP == OP_CAPTURES && h->hp.fixed_total < 0))
      if (int rc = bt_prepass(h, lay, n, s, &layp)) return rc;
#define MRX_L(B) hipLaunchKernelGGL((k_match<OP, B>), dim3(grid_for(n, kBlock)), dim3(kBlock), lds_for(h), s, h->hp.dev, \
                                   H_BLOB(h), layp, n, d_s, d_e, d_flag)
    MRX_BT_DISPATCH(bt_kernel_kind(h, plan_uses_backtracker(h) || (OP == OP_CAPTURES && h->hp.fixed_total < 0)), MRX_L);
#undef MRX_L
    g_last_kernel = "k_match";
  }
  HIP_TRY(hipGetLastError());
  tm.stop();
  return MRX_OK;
}

// ST_FUSED launch shape.  Grid: the workgroups that are resident at once (4 per CU at the kernel's
// ~118 VGPRs / 37 KiB LDS; more would only queue behind them -- harmless, tasks are handed out by
// ticket -- but each wavefront of the grid owns a record region).
// Off by default: measured on the headline workload (profiles/r02_fused_findall.md) the one launch takes
// 0.35 ms against 0.32 ms for scan -> sums -> decode.  The scan loop is bound by instruction issue as much
// as by HBM, so the record expansion costs the same issue slots whether it runs in the scan's launch or in
// its own, and what the fusion saves (0.46 GB of record traffic, two launch boundaries) is less than what
// the per-task hand-offs add.  MRX_FUSED=1 / mrx_debug_fused_findall(1): on for tasks of >= 32 KiB, 2: always.
std::atomic<int> g_fused{0};
std::atomic<int64_t> g_rec_skew{0};   // mrx_debug_rec_skew(): bytes (a multiple of 16) the record stream begins behind its allocation
std::atomic<int> g_fused_bpc{0};      // MRX_FUSED_BPC: workgroups per CU (measurement)
int64_t fused_grid_cap() {
  const int bpc = g_fused_bpc.load() > 0 ? g_fused_bpc.load() : 4;
  return (int64_t)(grid_cap() / 8) * bpc;
}
// Every 64-text task takes one ticket from ONE counter word (about 88 atomics per microsecond at best):
// tasks of less than 32 KiB would queue up there, so batches of short texts keep the three-launch form.
constexpr int64_t kFusedMinTaskBytes = 32768;
// '^'-anchored DFA plan whose search / findall / count are the anchored automaton's run from byte 0 (not the
// pure-literal case: simd_search is not anchored; with '$' only where the automaton carries end-of-text flags;
// not behind the exact-literal / prefilter paths)
bool anchored_at_zero(const mrx_handle* h) {
  const DevPlan& p = h->hp.dev;
  return !g_force_generic && (p.flags & PF_START_ANCHOR) && !(p.flags & PF_PURE_LITERAL) &&
         (!(p.flags & PF_END_ANCHOR) || p.off_fa_end >= 0) &&   // '^...$': the anchored automaton with end-of-text flags
         p.kind == PLAN_DFA && p.fa_bytes > 0 && !(p.flags & (PF_EXACT_LITERAL | PF_PREFILTER)) && h->hp.why_no_search.empty();
}
// longest text the event records of the streaming findall can describe (see run_findall)
constexpr int64_t kStreamMaxText = int64_t(1) << 26;
bool stream_text_too_long(int64_t max_text) { return max_text >= kStreamMaxText; }
// can this batch layout go through the streaming kernel?
bool stream_layout_ok(const Layout&, int64_t n) { return n > 0; }  // every layout has a streaming form
// fixed pitch, 16-byte aligned, 64 rows within 32-bit offsets: the fast strided form; otherwise the
// frame form (CSR, or a fixed pitch at arbitrary alignment)
bool strided_fast(const Layout& lay) {
  return !lay.offsets && (lay.stride % 16 == 0) && (((uintptr_t)lay.data) % 16 == 0) &&
         lay.stride * 64 < (int64_t(1) << 31);
}

// code columns (DevPlan::off_stcol32) on fixed-pitch batches; MRX_NO_CODE_COLUMNS=1 keeps the 4-bit columns (A/B runs)
static bool code_columns_on() {
  static const bool on = [] { const char* e = getenv("MRX_NO_CODE_COLUMNS"); return !(e && e[0] == '1'); }();
  return on;
}

template <int MODE>
void launch_stream(const mrx_handle* h, const Layout& lay, int64_t n, int32_t* d_counts, int32_t* d_nrecs,
                   EvRec* d_recs, int64_t rec_row, int32_t* d_s, int32_t* d_e, hipStream_t s,
                   const int32_t* d_vlen = nullptr, const uint32_t* d_vskip = nullptr, bool rec32 = false,
                   const FusedArgs* fused = nullptr, int fused_grid = 0) {
  const DevPlan& p = h->hp.dev;
  const int64_t nw = (n + 63) / 64;
  int64_t g = (nw + kStreamWaves - 1) / kStreamWaves;
  if (g > grid_cap()) g = grid_cap();
  if (MODE == ST_FUSED) g = fused_grid;   // one record region per wavefront of this grid
  const FusedArgs* fzv = fused;   // device copy (k_fused_init)
  const dim3 grid((unsigned)g), block(64 * kStreamWaves);
  const int kind = MODE == ST_FIRST ? p.fa_kind : p.st_kind;   // automaton form of this mode
  const bool table = kind == 2;
  const bool wide = kind == 3;
  // class table walked two bytes per lookup (plans with a reset byte; never the anchored automaton)
  const bool pairs = MODE != ST_FIRST && (table || wide) && p.off_stg_pair >= 0 && g_pair_tables;
  const size_t lds = pairs ? (size_t)p.stg_bytes
                           : wide ? 2048 : !table ? 0 : (size_t)(MODE == ST_FIRST ? p.fa_bytes : p.stg_bytes);
#define MRX_LAUNCH_R(AUTO, CSR, R32)                                                              \
  hipLaunchKernelGGL((k_stream_findall<MODE, (AUTO == 2 ? MRX_STREAM_CHUNK_TABLE : MRX_STREAM_CHUNK), AUTO, CSR, 0, R32>), grid, block, lds, s, p, \
                     H_BLOB(h), lay.data, lay.stride, lay.lens, lay.len, lay.offsets, n, d_counts, \
                     d_nrecs, d_recs, rec_row, d_s, d_e, (const int32_t*)nullptr, (const uint32_t*)nullptr, fzv)
#define MRX_LAUNCH(AUTO, CSR)                                                                     \
  do {                                                                                            \
    if constexpr (MODE == ST_RECORDS || MODE == ST_FUSED) {                                       \
      if (rec32) MRX_LAUNCH_R(AUTO, CSR, 1); else MRX_LAUNCH_R(AUTO, CSR, 0);                     \
    } else MRX_LAUNCH_R(AUTO, CSR, 0);                                                            \
  } while (0)
  if (d_vlen) {   // pieces of long texts / views: lay.offsets = their start offsets, n = how many
    if constexpr (MODE == ST_RECORDS || MODE == ST_COUNT || MODE == ST_SEARCH || MODE == ST_FIRST) {
#define MRX_LAUNCH_V(AUTO)                                                                        \
  hipLaunchKernelGGL((k_stream_findall<MODE, (AUTO == 2 ? MRX_STREAM_CHUNK_TABLE : MRX_STREAM_CHUNK), AUTO, 1, 1>), grid, block, lds, s, p, \
                     H_BLOB(h), lay.data, lay.stride, lay.lens, lay.len, lay.offsets, n, d_counts, \
                     d_nrecs, d_recs, rec_row, d_s, d_e, d_vlen, d_vskip)
      if (pairs) { if constexpr (MODE != ST_FIRST) MRX_LAUNCH_V(4); }
      else if (table) MRX_LAUNCH_V(2);
      else if (wide) MRX_LAUNCH_V(3);
      else MRX_LAUNCH_V(1);
#undef MRX_LAUNCH_V
    }
  } else if (!strided_fast(lay)) {
    if constexpr (MODE != ST_ROWS) {   // (event rows: the aligned fixed pitch only, the caller has checked)
    if (pairs) { if constexpr (MODE != ST_FIRST) MRX_LAUNCH(4, 1); }
    else if (table) MRX_LAUNCH(2, 1);
    else if (wide) MRX_LAUNCH(3, 1);
    else MRX_LAUNCH(1, 1);
    }
  } else {
    if (pairs) { if constexpr (MODE != ST_FIRST) MRX_LAUNCH(4, 0); }
    else if (table) MRX_LAUNCH(2, 0);
    else if (wide) MRX_LAUNCH(3, 0);
    // (records / search / fused: -16 % VALU instructions, findall step -1 %, search +6 %; the count kernel measured 4 %
    // slower in this form -- profiles/r03_scan_forms.md -- and keeps the 4-bit columns)
    else if (MODE != ST_FIRST && MODE != ST_COUNT && p.off_stcol32 >= 0 && code_columns_on()) { if constexpr (MODE != ST_FIRST && MODE != ST_COUNT) MRX_LAUNCH(5, 0); }
    else MRX_LAUNCH(1, 0);
  }
#undef MRX_LAUNCH
#undef MRX_LAUNCH_R
}

// Ragged CSR batches on k_stream_dyn (texts handed to lanes as they fall free): plans with a reset byte, and
// batches whose 256-text tasks fill at least three quarters of the device's wavefront slots (16 per CU) --
// the lanes are busy 71 % of the time instead of 51 % (lengths U[64, 1024]), which only pays while there
// are enough tasks; below that the 64-text wavefronts of k_stream_findall keep more of the device busy.
// mrx_debug_dynamic_texts(): 1 always, 2 never.
std::atomic<int> g_dyn_mode{0};
bool dyn_ok(const mrx_handle* h, const Layout& lay, int64_t n) {
  const DevPlan& p = h->hp.dev;
  if (g_dyn_mode == 2 || !lay.offsets || p.st_reset_byte < 0 || n <= 0) return false;
  return g_dyn_mode == 1 || n >= (int64_t)kDynTexts * (grid_cap() / 8) * 16 * 3 / 4;
}
template <int MODE>
void launch_stream_dyn(const mrx_handle* h, const Layout& lay, int64_t n, int32_t* d_counts, int32_t* d_nrecs, EvRec* d_recs,
                       int32_t* d_s, int32_t* d_e, hipStream_t s, bool rec32) {
  const DevPlan& p = h->hp.dev;
  const int64_t ntasks = (n + kDynTexts - 1) / kDynTexts;
  int64_t g = (ntasks + kStreamWaves - 1) / kStreamWaves;
  if (g > grid_cap()) g = grid_cap();
  const dim3 grid((unsigned)g), block(64 * kStreamWaves);
  const bool table = p.st_kind == 2, wide = p.st_kind == 3;
  const bool pairs = (table || wide) && p.off_stg_pair >= 0 && g_pair_tables;
  const size_t lds = pairs ? (size_t)p.stg_bytes : wide ? 2048 : !table ? 0 : (size_t)p.stg_bytes;
#define MRX_DYN_R(AUTO, R32)                                                                          \
  hipLaunchKernelGGL((k_stream_dyn<MODE, AUTO, R32>), grid, block, lds, s, p, H_BLOB(h), lay.data, lay.offsets, lay.vlen, n, \
                     d_counts, d_nrecs, d_recs, d_s, d_e)
#define MRX_DYN(AUTO)                                                                                  \
  do {                                                                                                 \
    if constexpr (MODE == ST_RECORDS) { if (rec32) MRX_DYN_R(AUTO, 1); else MRX_DYN_R(AUTO, 0); }      \
    else MRX_DYN_R(AUTO, 0);                                                                           \
  } while (0)
  if (pairs) MRX_DYN(4);
  else if (table) MRX_DYN(2);
  else if (wide) MRX_DYN(3);
  else MRX_DYN(1);
#undef MRX_DYN
#undef MRX_DYN_R
}

// (a named function, not a lambda: hipcc gave two namespace-scope lambdas of this shape ONE body -- the second
// variable was initialised by the first one's getenv -- see env_int's other user, g_subs_group)
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// Measurement knobs of the call path, read from the environment ONCE (at the first call that looks): no getenv per call.
struct EnvKnobs {
  int decode_grid, decode_reverse, piece_c, fused_skew, fused_debug;
  int subc_debug;   // k_subc_sizes / k_subc_emit with phases left out (timing only): 1 no walk, 2 no bitmaps, 4 no per-match store; 8 no gap copies, 32 no replacement bytes, 64 no stores
  int subs_debug;   // k_subs_wave with phases left out (timing only, the output is wrong): 1 replacement bytes, 2 kept bytes, 4 stores, 8 frame phase, 16 match phase
  bool piece_c_set;
};
static const EnvKnobs& env_knobs() {
  static const EnvKnobs k = [] {
    EnvKnobs e;
    e.decode_grid = env_int("MRX_DECODE_GRID", 0);
    e.decode_reverse = env_int("MRX_DECODE_REVERSE", 0);
    e.piece_c_set = getenv("MRX_PIECE_C") != nullptr;
    e.piece_c = env_int("MRX_PIECE_C", 0);
    e.fused_skew = env_int("MRX_FUSED_SKEW", 0);
    e.fused_debug = env_int("MRX_FUSED_DEBUG", 0);
    e.subs_debug = env_int("MRX_SUBS_DEBUG", 0);
    e.subc_debug = env_int("MRX_SUBC_DEBUG", 0);
    return e;
  }();
  return k;
}

// Long texts on the streaming kernels: cut into pieces at synchronising bytes when one lane per text
// would leave most of the device idle.
struct Pieces {
  bool on = false;
  int C = 0;
  int64_t nv = 0, data_bytes = 0;
  int64_t* vfirst = nullptr;   // [n + 1]: first piece of every text
  int64_t* vstart = nullptr;
  int32_t* vlen = nullptr;
  uint32_t* vskip = nullptr;
  int32_t* vbase = nullptr;
  int32_t* back = nullptr;
  Layout lay{};   // the pieces as a batch: offsets = vstart
};
// byte count and longest text of a CSR batch: both live on the device (one small kernel, one sync)
int csr_stats(const Layout& lay, int64_t n, hipStream_t s, int64_t* total, int64_t* max_len) {
  int32_t* d_max = nullptr;
  int32_t m = 0;
  HIP_TRY(scratch_alloc((void**)&d_max, sizeof(int32_t), s));
  HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(k_max_len, dim3(grid_for(n, kBlock * 8)), dim3(kBlock), 0, s, lay.offsets, n, d_max);
  HIP_TRY(hipMemcpyAsync(&m, d_max, sizeof m, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(total, lay.offsets + n, sizeof *total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(scratch_free(d_max, s));
  *max_len = m;
  return MRX_OK;
}
// known_total / known_max: csr_stats() of the batch when the caller has them already (< 0: not)
int pieces_prepare(const mrx_handle* h, const Layout& lay, int64_t n, hipStream_t s, Pieces* pc,
                   int64_t known_total = -1, int64_t known_max = -1, bool disjoint = false, bool mwalk = false) {
  const DevPlan& p = h->hp.dev;
  pc->on = false;
  if (p.st_nsync <= 0 || g_long_text_mode == 2 || (g_long_text_mode == 3 && disjoint) || n <= 0) return MRX_OK;
  // pays when one lane per text leaves the device mostly idle, or for outliers of a ragged batch; a
  // fixed-length batch of many texts is decided before anything is launched
  const bool env_pieces = env_knobs().piece_c_set;   // measurement
  if (g_long_text_mode == 0 && n > 131072 && !lay.offsets && !lay.lens && !env_pieces) return MRX_OK;
  // count / search of a large CSR batch stay free of any host synchronisation: its outliers are only
  // looked for where the caller has the batch's statistics anyway (findall)
  if (g_long_text_mode == 0 && n > 131072 && lay.offsets && known_total < 0 && !env_pieces) return MRX_OK;
  int64_t total = 0, max_len = 0;
  if (lay.offsets && known_total >= 0) {
    total = known_total; max_len = known_max;
  } else if (lay.offsets) {
    if (int rc = csr_stats(lay, n, s, &total, &max_len)) return rc;
  } else {
    max_len = lay.lens ? lay.stride : (int64_t)lay.len;
    total = n * lay.stride;
  }
  int C;
  const int env_c = env_knobs().piece_c;   // measurement: piece size
  if (env_c > 0 && g_long_text_mode == 0) {
    C = env_c;
    if (max_len <= C) return MRX_OK;
  } else
  if (g_long_text_mode == 1 || g_long_text_mode == 3) {
    C = 200;   // tests: cut even short texts, at positions that are not multiples of 16
    if (max_len <= C) return MRX_OK;
  } else {
    if (max_len < 4096) return MRX_OK;   // nothing long enough to cut
    // batches that fill the device with one lane per text are cut only for the sake of texts far
    // longer than the rest (ragged batches: one lane would still be walking long after the others)
    const int64_t avg = total / n;
    // stepper plans (disjoint pieces): measured on the reference's benchmark texts against the wavefront-per-text
    // kernel -- pieces win on 11-74 KB texts of the plain route (dense candidates: flexible_phone 1.57 -> 0.84 ms,
    // multi_format_phone 2.42 -> 1.57, alternation_quantifiers 2.84 -> 1.87) and lose below that and on every
    // required-byte plan (sparse candidates are the wavefront kernel's best case)
    // (multi-walk plans: a piece is scanned once whatever it holds, so pieces pay as soon as one lane per text
    // leaves the device short of lanes)
    if (disjoint && avg < (mwalk ? 2048 : 10000)) return MRX_OK;
    if (n > 131072 && !(max_len >= 32768 && max_len >= 8 * avg && (lay.offsets || lay.lens))) return MRX_OK;
    const int64_t want = (total + 262143) / 262144;   // about 2^18 pieces
    C = (int)((want + 255) / 256 * 256);
    if (C < 2048) C = 2048;
    if (C >= max_len) return MRX_OK;
  }
  // pieces per text -> prefix sums -> how many there are
  HIP_TRY(scratch_alloc((void**)&pc->vfirst, sizeof(int64_t) * (n + 1), s));
  int64_t nv = 0;
  if (!lay.offsets && !lay.lens) {   // one length: nothing to count, nothing to read back
    const int64_t cpt = (max_len + C - 1) / C;
    nv = n * cpt;
    hipLaunchKernelGGL(k_virt_uniform, dim3(grid_for(n + 1, kBlock)), dim3(kBlock), 0, s, n, cpt, pc->vfirst);
  } else {
    int32_t* d_cnt = nullptr;
    int64_t* d_tot = nullptr;
    HIP_TRY(scratch_alloc((void**)&d_cnt, sizeof(int32_t) * n, s));
    HIP_TRY(scratch_alloc((void**)&d_tot, sizeof(int64_t), s));
    hipLaunchKernelGGL(k_virt_count, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, lay, n, C, d_cnt);
    if (int rc = device_scan<int32_t>(d_cnt, n, pc->vfirst, d_tot, s)) return rc;
    HIP_TRY(hipMemcpyAsync(&nv, d_tot, sizeof nv, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(scratch_free(d_cnt, s));
    HIP_TRY(scratch_free(d_tot, s));
  }
  if (nv <= n || nv > (int64_t(1) << 24)) {   // nothing to cut after all, or an absurd number of pieces
    HIP_TRY(scratch_free(pc->vfirst, s));
    pc->vfirst = nullptr;
    return MRX_OK;
  }
  pc->C = C; pc->nv = nv; pc->data_bytes = total;
  HIP_TRY(scratch_alloc((void**)&pc->vstart, sizeof(int64_t) * (pc->nv + 1), s));
  HIP_TRY(scratch_alloc((void**)&pc->vlen, sizeof(int32_t) * pc->nv, s));
  HIP_TRY(scratch_alloc((void**)&pc->vskip, sizeof(uint32_t) * pc->nv, s));
  HIP_TRY(scratch_alloc((void**)&pc->vbase, sizeof(int32_t) * pc->nv, s));
  HIP_TRY(scratch_alloc((void**)&pc->back, sizeof(int32_t) * pc->nv, s));
  const uint8_t* sync = H_BLOB(h) + p.off_st_sync;
  hipLaunchKernelGGL(k_virt_check, dim3(grid_for(pc->nv, kBlock)), dim3(kBlock), 0, s, lay, n, pc->vfirst, pc->nv, C, sync,
                     pc->back);
  hipLaunchKernelGGL(k_virt_fill, dim3(grid_for(pc->nv, kBlock)), dim3(kBlock), 0, s, lay, n, pc->vfirst, pc->nv, C,
                     pc->back, pc->vstart, pc->vlen, pc->vskip, pc->vbase, disjoint ? 1 : 0);
  HIP_TRY(hipGetLastError());
  pc->lay = Layout{lay.data, pc->vstart, 0, nullptr, 0};
  if (disjoint) pc->lay.vlen = pc->vlen;   // a view: the stepper's kernels read the pieces through Layout::text()
  pc->on = true;
  return MRX_OK;
}
int pieces_release(Pieces* pc, hipStream_t s) {
  if (!pc->on) return MRX_OK;
  HIP_TRY(scratch_free(pc->vfirst, s));
  HIP_TRY(scratch_free(pc->vstart, s));
  HIP_TRY(scratch_free(pc->vlen, s));
  HIP_TRY(scratch_free(pc->vskip, s));
  HIP_TRY(scratch_free(pc->vbase, s));
  HIP_TRY(scratch_free(pc->back, s));
  pc->on = false;
  return MRX_OK;
}

// findall over the pieces: scan, prefix sums and decode as for any CSR batch, then the texts'
// entries of the per-piece prefix sums
int findall_pieces(const mrx_handle* h, const Pieces& pc, int64_t n, int64_t* d_prefix, int32_t* d_spans,
                   int64_t span_cap, int64_t* d_total, hipStream_t s) {
  const DevPlan& p = h->hp.dev;
  const int64_t nv = pc.nv, nw = (nv + 63) / 64;
  const size_t nrec = (size_t)(pc.data_bytes / 16 + 256 * nw + 256);
  int32_t* d_vcounts = nullptr;
  EvRec* d_recs = nullptr;
  int32_t* d_nrecs = nullptr;
  int64_t* d_wbase = nullptr;
  int64_t* d_vprefix = nullptr;
  int64_t* d_tsum = nullptr;
  const int64_t ntiles = (nw + kScanTile - 1) / kScanTile;
  HIP_TRY(scratch_alloc((void**)&d_vcounts, sizeof(int32_t) * nv, s));
  HIP_TRY(scratch_alloc((void**)&d_recs, sizeof(EvRec) * nrec, s));
  HIP_TRY(scratch_alloc((void**)&d_nrecs, sizeof(int32_t) * 2 * nw, s));
  HIP_TRY(scratch_alloc((void**)&d_wbase, sizeof(int64_t) * (nw + 1), s));
  HIP_TRY(scratch_alloc((void**)&d_vprefix, sizeof(int64_t) * (nv + 1), s));
  HIP_TRY(scratch_alloc((void**)&d_tsum, sizeof(int64_t) * ntiles, s));
  {
    ScanTimer tm(s);
    launch_stream<ST_RECORDS>(h, pc.lay, nv, d_vcounts, d_nrecs, d_recs, 0, nullptr, nullptr, s, pc.vlen, pc.vskip);
    g_last_kernel = "k_stream_findall_pieces";
    HIP_TRY(hipGetLastError());
    tm.stop();
  }
  hipLaunchKernelGGL(k_scan_local<int32_t>, dim3((unsigned)ntiles), dim3(kScanBlock), 0, s, d_nrecs + nw, nw, d_wbase,
                     d_tsum);
  hipLaunchKernelGGL((k_decode<false, true>), dim3(grid_for(nv, kBlock)), dim3(kBlock), 0, s, nv, d_nrecs, d_recs,
                     (int64_t)0, pc.lay.offsets, d_vcounts, d_wbase, d_tsum, d_vprefix, d_spans, span_cap, p.st_fixed_len,
                     d_total, pc.vbase);
  hipLaunchKernelGGL(k_virt_prefix, dim3(grid_for(n + 1, kBlock)), dim3(kBlock), 0, s, n, pc.vfirst, d_vprefix, d_prefix);
  HIP_TRY(hipGetLastError());
  HIP_TRY(scratch_free(d_vcounts, s));
  HIP_TRY(scratch_free(d_recs, s));
  HIP_TRY(scratch_free(d_nrecs, s));
  HIP_TRY(scratch_free(d_wbase, s));
  HIP_TRY(scratch_free(d_vprefix, s));
  HIP_TRY(scratch_free(d_tsum, s));
  return MRX_OK;
}

// findall of a large fixed-pitch batch in two halves on two streams, so that the record decode of the first
// half runs under the scan of the second (the scan is bound by instruction issue as much as by HBM, the decode
// by memory latency: side by side they take less than one after the other):
//   caller's stream:  scan(A) | scan(B), sums(B) ............ | decode(B, base = spans of A)
//   side stream:                sums(A), decode(A) ...........^ (joined before decode(B))
// Results are those of the one-batch form: decode(B) adds A's total to its CSR offsets.
// OFF by default -- measured on the headline workload (one box, bench.py): 0.354 ms per step against 0.336 ms for
// the one-batch form; each half-size scan takes 0.150 ms where the whole one takes 0.243 (half the wavefronts
// per launch: the ramp-up and the tail of a launch weigh twice), which is more than the overlap returns.  What
// does pay is overlapping WHOLE calls on two caller streams (bench.py --streams 2: 0.287 ms on the same box).
// MRX_FINDALL_SPLIT=1 / mrx_debug_split_findall(1): on.
std::atomic<int> g_split_findall{env_int("MRX_FINDALL_SPLIT", 0)};
// findall by event rows (ST_ROWS, k_decode_rows): 0 = when the handle's last batch was full of matches, 1 = whenever the
// batch has the shape, 2 = never (MRX_DENSE_ROWS / mrx_debug_dense_rows)
std::atomic<int> g_dense_rows{env_int("MRX_DENSE_ROWS", 0)};
constexpr int64_t kSplitMinTexts = 1 << 18;
struct SideStream {
  hipStream_t side = nullptr;
  hipEvent_t scanned = nullptr, decoded = nullptr;
};
thread_local std::map<int, SideStream> g_side;
static int side_stream(SideStream** out) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  SideStream& ss = g_side[dev];
  if (!ss.side) {
    HIP_TRY(hipStreamCreateWithFlags(&ss.side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ss.scanned, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ss.decoded, hipEventDisableTiming));
  }
  *out = &ss;
  return MRX_OK;
}
static int findall_split(const mrx_handle* h, const Layout& lay, int64_t n, int32_t* d_counts, int64_t* d_prefix,
                         int32_t* d_spans, int64_t span_cap, int64_t* d_total, int64_t rec_row, bool rec32, bool pack16,
                         hipStream_t s) {
  const DevPlan& p = h->hp.dev;
  SideStream* ss = nullptr;
  if (int rc = side_stream(&ss)) return rc;
  const int64_t nA = (n / 2) & ~int64_t(63), nB = n - nA;
  const int64_t nwA = nA / 64, nwB = (nB + 63) / 64;
  const int64_t ntA = (nwA + kScanTile - 1) / kScanTile, ntB = (nwB + kScanTile - 1) / kScanTile;
  EvRec* d_recs = nullptr;
  int32_t *d_nrA = nullptr, *d_nrB = nullptr;
  int64_t *d_wbA = nullptr, *d_wbB = nullptr, *d_tsA = nullptr, *d_tsB = nullptr, *d_totA = nullptr;
  HIP_TRY(scratch_alloc((void**)&d_recs, sizeof(EvRec) * (size_t)rec_row * n, s));
  HIP_TRY(scratch_alloc((void**)&d_nrA, sizeof(int32_t) * 2 * nwA, s));
  HIP_TRY(scratch_alloc((void**)&d_nrB, sizeof(int32_t) * 2 * nwB, s));
  HIP_TRY(scratch_alloc((void**)&d_wbA, sizeof(int64_t) * (nwA + 1), s));
  HIP_TRY(scratch_alloc((void**)&d_wbB, sizeof(int64_t) * (nwB + 1), s));
  HIP_TRY(scratch_alloc((void**)&d_tsA, sizeof(int64_t) * ntA, s));
  HIP_TRY(scratch_alloc((void**)&d_tsB, sizeof(int64_t) * ntB, s));
  HIP_TRY(scratch_alloc((void**)&d_totA, sizeof(int64_t), s));
  Layout layB = lay;
  layB.data = lay.data + nA * lay.stride;
  if (lay.lens) layB.lens = lay.lens + nA;
  EvRec* d_recsB = d_recs + (size_t)rec_row * nA;
#define MRX_DECODE_HALF(STREAM, N, NRECS, RECS, COUNTS, WBASE, TSUM, PREFIX, TOTAL, BASE)                                   \
  do {                                                                                                                     \
    const dim3 dg_((unsigned)grid_for((N), kBlock) * (pack16 ? 2 : 1)), db_(kBlock);                                        \
    if (pack16 && rec32)                                                                                                   \
      hipLaunchKernelGGL((k_decode<true, false, true>), dg_, db_, 0, STREAM, (N), NRECS, RECS, rec_row, (const int64_t*)nullptr, \
                         COUNTS, WBASE, TSUM, PREFIX, d_spans, span_cap, p.st_fixed_len, TOTAL, (const int32_t*)nullptr, BASE); \
    else if (pack16)                                                                                                       \
      hipLaunchKernelGGL(k_decode<true>, dg_, db_, 0, STREAM, (N), NRECS, RECS, rec_row, (const int64_t*)nullptr,            \
                         COUNTS, WBASE, TSUM, PREFIX, d_spans, span_cap, p.st_fixed_len, TOTAL, (const int32_t*)nullptr, BASE); \
    else                                                                                                                   \
      hipLaunchKernelGGL(k_decode<false>, dg_, db_, 0, STREAM, (N), NRECS, RECS, rec_row, (const int64_t*)nullptr,           \
                         COUNTS, WBASE, TSUM, PREFIX, d_spans, span_cap, p.st_fixed_len, TOTAL, (const int32_t*)nullptr, BASE); \
  } while (0)
  {
    ScanTimer tm(s);
    launch_stream<ST_RECORDS>(h, lay, nA, d_counts, d_nrA, d_recs, rec_row, nullptr, nullptr, s, nullptr, nullptr, rec32);
    HIP_TRY(hipGetLastError());
    tm.stop();
  }
  HIP_TRY(hipEventRecord(ss->scanned, s));
  HIP_TRY(hipStreamWaitEvent(ss->side, ss->scanned, 0));
  hipLaunchKernelGGL(k_scan_local<int32_t>, dim3((unsigned)ntA), dim3(kScanBlock), 0, ss->side, d_nrA + nwA, nwA, d_wbA, d_tsA);
  MRX_DECODE_HALF(ss->side, nA, d_nrA, d_recs, d_counts, d_wbA, d_tsA, d_prefix, d_totA, (const int64_t*)nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ss->decoded, ss->side));
  {
    ScanTimer tm(s);
    launch_stream<ST_RECORDS>(h, layB, nB, d_counts + nA, d_nrB, d_recsB, rec_row, nullptr, nullptr, s, nullptr, nullptr, rec32);
    HIP_TRY(hipGetLastError());
    tm.stop();
  }
  hipLaunchKernelGGL(k_scan_local<int32_t>, dim3((unsigned)ntB), dim3(kScanBlock), 0, s, d_nrB + nwB, nwB, d_wbB, d_tsB);
  HIP_TRY(hipStreamWaitEvent(s, ss->decoded, 0));
  MRX_DECODE_HALF(s, nB, d_nrB, d_recsB, d_counts + nA, d_wbB, d_tsB, d_prefix + nA, d_total, (const int64_t*)d_totA);
#undef MRX_DECODE_HALF
  HIP_TRY(hipGetLastError());
  g_last_kernel = "k_stream_findall";
  HIP_TRY(scratch_free(d_recs, s)); HIP_TRY(scratch_free(d_nrA, s)); HIP_TRY(scratch_free(d_nrB, s));
  HIP_TRY(scratch_free(d_wbA, s)); HIP_TRY(scratch_free(d_wbB, s)); HIP_TRY(scratch_free(d_tsA, s));
  HIP_TRY(scratch_free(d_tsB, s)); HIP_TRY(scratch_free(d_totA, s));
  return MRX_OK;
}

// Long texts on the stepper's routes (findall / count of plans that do not stream): cut at bytes on which every
// walk dies and none begins (DevPlan::off_st_sync of such plans, mrx_plan.cpp), the disjoint pieces searched as
// texts of their own with one lane each -- where the wavefront-per-text kernel pays for its busiest lane and for the
// walks it repeats.  The recursion runs with t_in_pieces set: the batch of pieces is a view, not a CSR batch.
thread_local bool t_in_pieces = false;
// findall by pieces: the pieces' bases, for an inner route whose emit kernel can add them itself (k_mwalk), and whether
// it did -- otherwise k_virt_add_base goes over the spans once more
thread_local int t_piece_tries = -1;   // findall by pieces: the outer call's choice for a PF_MW_TRIES plan (-1: none, 0 marks + stepper, 1 tries)
thread_local const int32_t* t_piece_vbase = nullptr;
thread_local bool t_piece_base_applied = false;
// plain-route stepper plan on long texts: pieces (true) or the wavefront kernel (false)?  Decided by the share of
// bytes that can begin a walk in the first MiB of the batch (one small kernel, one 4-byte read-back): measured on the
// reference's list, dense candidates (phone numbers everywhere: 30 %) are where one lane per piece wins (1.57 -> 0.84
// ms), sparse ones (a number every few hundred bytes: 1-2 %) where the wavefront kernel does (0.52 against 1.0 ms).
static int dense_candidates(const mrx_handle* h, const Layout& lay, int64_t n, hipStream_t s, bool* dense) {
  *dense = true;
  if (g_long_text_mode == 1) return MRX_OK;
  int64_t bytes = 0;
  if (lay.offsets) {
    HIP_TRY(hipMemcpyAsync(&bytes, lay.offsets + n, sizeof bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  } else {
    bytes = n * lay.stride;
  }
  if (bytes > (1 << 20)) bytes = 1 << 20;
  if (bytes <= 0) return MRX_OK;
  unsigned int* d_hits = nullptr;
  unsigned int hits = 0;
  HIP_TRY(scratch_alloc((void**)&d_hits, sizeof(unsigned int), s));
  HIP_TRY(hipMemsetAsync(d_hits, 0, sizeof(unsigned int), s));
  hipLaunchKernelGGL(k_candidate_density, dim3(256), dim3(kBlock), 0, s, h->hp.dev, H_BLOB(h), lay.data, bytes, d_hits);
  HIP_TRY(hipMemcpyAsync(&hits, d_hits, sizeof hits, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(scratch_free(d_hits, s));
  *dense = (double)hits >= 0.08 * (double)bytes;
  return MRX_OK;
}
// share of the bytes of the batch's first MiB that are synchronising bytes of the plan (DevPlan::off_st_sync): pieces are
// cut at those, and a text without them (timestamps for a timestamp pattern) is cut nowhere -- every piece then reaches
// back to the text's start (measured: datetime_quantifiers 1.1 ms on the wavefront kernel, 17 ms in "pieces")
__global__ __launch_bounds__(kBlock) void k_sync_density(const uint8_t* __restrict__ sync, const uint8_t* __restrict__ data,
                                                         int64_t nbytes, unsigned int* __restrict__ hits) {
  __shared__ uint8_t tab[256];
  for (int b = threadIdx.x; b < 256; b += blockDim.x) tab[b] = sync[b] ? 1 : 0;
  __syncthreads();
  unsigned int k = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * blockDim.x) k += tab[data[i]];
  for (int off = 32; off > 0; off >>= 1) k += __shfl_xor(k, off);
  if ((threadIdx.x & 63) == 0 && k) atomicAdd(hits, k);
}
static int sync_bytes_frequent(const mrx_handle* h, const Layout& lay, int64_t n, hipStream_t s, bool* frequent) {
  *frequent = false;
  const DevPlan& p = h->hp.dev;
  if (p.st_nsync <= 0 || p.off_st_sync < 0) return MRX_OK;
  int64_t bytes = 0;
  if (lay.offsets) {
    HIP_TRY(hipMemcpyAsync(&bytes, lay.offsets + n, sizeof bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  } else {
    bytes = n * lay.stride;
  }
  if (bytes > (1 << 20)) bytes = 1 << 20;
  if (bytes <= 0) return MRX_OK;
  unsigned int* d_hits = nullptr;
  unsigned int hits = 0;
  HIP_TRY(scratch_alloc((void**)&d_hits, sizeof(unsigned int), s));
  HIP_TRY(hipMemsetAsync(d_hits, 0, sizeof(unsigned int), s));
  hipLaunchKernelGGL(k_sync_density, dim3(256), dim3(kBlock), 0, s, H_BLOB(h) + p.off_st_sync, lay.data, bytes, d_hits);
  HIP_TRY(hipMemcpyAsync(&hits, d_hits, sizeof hits, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *frequent = (double)hits >= 0.015 * (double)bytes;   // one at least every 64 bytes or so
  return MRX_OK;
}

int run_findall(const mrx_handle* h, const Layout& lay, int64_t n, int64_t* d_prefix, int32_t* d_spans, int64_t span_cap,
                int64_t* total, void* stream, bool match_next_sequence = false, int64_t known_total = -1, int64_t known_max = -1);

// One findall call.  The members are the state the routes share (the launch macros of this file read the route flags by
// name); one member function per route -- round 3's run_findall was a single 420-line function steering ~25 flags.
// Two routes for one plan and batch shape, measured by the handle (mrx_handle::req_tune[key]): the first four eligible calls
// take route 1, 2, 1, 2, the last two between HIP events on the caller's stream; a later call that finds the last event
// complete keeps the faster route.  Returns the route this call takes (1 or 2); *slot >= 0 when the call is one of the
// four (ab_tuner_end records its closing event once the call's work is enqueued).  No call waits; a stream under graph
// capture is left alone (route 1).
static int ab_tuner_begin(const mrx_handle* h, uint32_t key, hipStream_t s, const char* what, int* slot) {
  *slot = -1;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 1;
  std::lock_guard<std::mutex> lk(h->tune_mu);
  mrx_handle::ReqTune& t = h->req_tune[key];
  if (t.choice) return t.choice;
  if (t.issued == 4) {
    if (hipEventQuery(t.ev[3][1]) != hipSuccess) return 1;   // not through yet: the default route, unmeasured
    if (hipEventElapsedTime(&t.ms_wave, t.ev[2][0], t.ev[2][1]) != hipSuccess ||
        hipEventElapsedTime(&t.ms_pieces, t.ev[3][0], t.ev[3][1]) != hipSuccess) { t.choice = 1; return 1; }
    t.choice = t.ms_pieces < t.ms_wave ? 2 : 1;
    static const bool verbose = getenv("MRX_TUNE_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "mrx: %s of '%s' (key %08x): pending tries %.3f ms, marks + stepper %.3f ms -> %s\n", what,
                         h->hp.pattern.c_str(), key, t.ms_wave, t.ms_pieces, t.choice == 2 ? "marks" : "tries");
    return t.choice;
  }
  if (!t.ev[3][1]) {
    t.dev = t_dev;
    for (auto& pr : t.ev)
      for (auto& e : pr)
        if (hipEventCreate(&e) != hipSuccess) { t.choice = 1; return 1; }
  }
  if (t.dev != t_dev) return 1;
  const int sl = t.issued++;
  if (hipEventRecord(t.ev[sl][0], s) != hipSuccess) { t.choice = 1; return 1; }
  *slot = sl;
  return (sl & 1) ? 2 : 1;
}
static void ab_tuner_end(const mrx_handle* h, uint32_t key, int slot, hipStream_t s) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(h->tune_mu);
  mrx_handle::ReqTune& t = h->req_tune[key];
  if (t.ev[slot][1]) (void)hipEventRecord(t.ev[slot][1], s);
}

struct FindallJob {
  const mrx_handle* h;
  const Layout& lay;
  int64_t n;
  int64_t* d_prefix;
  int32_t* d_spans;
  int64_t span_cap;
  int64_t* total;
  hipStream_t s;
  bool match_next_sequence;
  int64_t known_total, known_max;
  const DevPlan& p;
  int32_t* d_counts = nullptr;
  int64_t* d_total = nullptr;
  // ---- route (chosen by choose_route(); the launch macros of this file read these by name) ----
  bool stream_ok = false, use_req_route = false, wstep_bits = false, wstep_lz = false, wstep_empty = false, mw_empty = false, mw_tries = false;
  bool mwalk_req = false, wstep_mwalk = false;
  DevPlan pk;                      // what the lane kernels are launched with
  int wstep_mwalk_k = 0;
  bool wstep_mwalk_pk = false;     // k_mwalk's packed-start form (every text below 64 KiB)
  bool step_ok = false;
  bool bits_fixed = false;         // bitset program whose matches all have one length: the union pass counts and emits by itself
  bool bits_fixed_slots = false;   // ... with the spans in slot rows (one pass)
  bool fused = false;              // streaming path: scan, CSR offsets and spans in one launch (ST_FUSED / k_stream_bits)
  bool split_done = false;         // streaming path: two halves on two streams (findall_split)
  bool dyn = false;                // streaming path: ragged CSR batch on k_stream_dyn (256-text tasks)
  unsigned long long* d_ctrl = nullptr;   // its ticket word, error word and descriptors
  int32_t* d_blimit = nullptr;            // bitset NFA: per-text limits of the first pass
  bool rec32 = false;              // streaming path: one record per two groups (positions fit 16 bits)
  int64_t max_text = int64_t(1) << 40;    // longest text of the batch, where known
  bool req_wave = false;           // the stepper's route on the wavefront-per-text kernel
  bool mwalk_two_pass = false;     // multi-walk plan: count pass + emit pass instead of slot rows
  bool wstep_bm = false;           // stepper behind the right-to-left pass that marks where matches begin (PF_BACKSET)
  bool wstep_bm_big = false;       // ... with its table class indexed (more than 96 states)
  int step_split = 0;              // > 0: lane kernel for texts below this length AND wavefront kernel for the rest
  Layout lay2;                     // lay + that split
  EvRec* d_recs = nullptr;
  EvRec* d_recs_alloc = nullptr;   // what d_recs was cut from when it is skewed (mrx_debug_rec_skew)
  int32_t* d_nrecs = nullptr;
  int64_t* d_wbase = nullptr;
  int32_t* d_slots = nullptr;
  int64_t rec_row = 0;
  Pieces pc;
  Layout lay_pre;                  // literal restatement of a backtracker-routed plan: lay + bt_prepass()
  int64_t csr_total = -1, csr_max = -1;   // CSR batches on the streaming path: byte count and longest text
  bool by_pieces = false;
  bool finished = false;           // a route that answered the whole call by itself (the stepper's pieces)
  // event rows (ST_ROWS) instead of records: a batch the handle's last call found full of matches
  bool rows = false, rows_shape = false;
  uint2* d_rows = nullptr;
  int64_t row_pairs = 0;
  // required-byte plan on long texts: the route being timed for the handle's tuner (0: none), and when the call began
  int tune_route = 0, tune_slot = 0;
  uint32_t tune_key = 0;

  FindallJob(const mrx_handle* h_, const Layout& lay_, int64_t n_, int64_t* d_prefix_, int32_t* d_spans_, int64_t span_cap_,
             int64_t* total_, hipStream_t s_, bool mns, int64_t kt, int64_t km)
      : h(h_), lay(lay_), n(n_), d_prefix(d_prefix_), d_spans(d_spans_), span_cap(span_cap_), total(total_), s(s_),
        match_next_sequence(mns), known_total(kt), known_max(km), p(h_->hp.dev), pk(h_->hp.dev), lay2(lay_), lay_pre(lay_) {}

  // Which kernels serve this call.  Streamable plans: the streaming scan (records -> prefix sums -> decode, or one of
  // its one-launch forms); everything else: the stepper family (multi-walk table, backward marks, bitset union pass,
  // wavefront per text, required-byte route) or the literal restatement.
  void choose_route() {
    stream_ok = !g_force_generic && (p.flags & PF_STREAMABLE) && stream_layout_ok(lay, n);
    // match_next_sequence: the caller (sub) wants the matches that iterating match_next from each
    // match end visits -- the plain walk even on plans whose findall takes the required-byte route
    use_req_route = (p.flags & PF_STEP_REQ) && !match_next_sequence;
    wstep_bits = (p.flags & PF_BSTEP) != 0;   // bitset NFA on the lane-per-text stepper
    wstep_lz = (p.flags & PF_LAZY_END) != 0;  // '$' on the LazyDFA search: lane per text only (the cache is the text's)
    // plans with empty matches: count, then emit (nearly every text has more matches than a slot row holds)
    // ... in one pass on k_mwalk when no walk of the plan ever reads beyond its match (PF_MW_EMPTY)
    mw_empty = (p.flags & PF_MW_EMPTY) != 0 && mwalk_enabled() && !match_next_sequence && g_force_generic < 2;
    wstep_empty = (p.flags & PF_STEP_EMPTY) != 0 && !match_next_sequence && g_force_generic < 2 && !mw_empty;
    // several walks in one pass instead of the restart-per-position loop (plain route; sub's match_next sequence is
    // the same list of matches, but a memchr-prefiltered match_next is not the plain search)
    mwalk_req = use_req_route && (p.flags & PF_MWALK_REQ) && mwalk_enabled();
    // ... and plain-route plans outside the multi-walk proofs whose walks stay within seven bytes of their match (PF_MW_TRIES)
    mw_tries = mw_tries_on(p) && !use_req_route && g_force_generic < 2;
    if (t_in_pieces && t_piece_tries >= 0) mw_tries = mw_tries && t_piece_tries == 1;   // (the pieces follow the call they belong to)
    if (mw_tries && !t_in_pieces && n >= 256 && backset_on(p) && !g_tries_always) {   // (marks + stepper is the other candidate)
      bool use_tries = true;
      tries_route_tuner(&use_tries);
      mw_tries = use_tries;
    }
    wstep_mwalk = (mwalk_req || (mwalk_on(p) && !use_req_route) || mw_empty || mw_tries) && !wstep_bits && !wstep_empty &&
                  !(match_next_sequence && (p.flags & PF_PREFILTER));
    pk = mwalk_req ? mwalk_req_plan(p) : p;
    wstep_mwalk_k = pk.mw_k;
    step_ok = g_force_generic < 2 &&
              (wstep_mwalk ||
               (match_next_sequence ? ((p.flags & PF_STEP_SEARCH) && !(p.flags & PF_PREFILTER))
                                    : (p.flags & (PF_STEPPABLE | PF_STEP_REQ | PF_STEP_EMPTY)) != 0));
    // bitset program whose matches all have one length: the union pass counts and emits by itself (k_bscan modes 2, 3)
    bits_fixed = step_ok && wstep_bits && bits_fixed_on(p) && !(match_next_sequence && (p.flags & PF_PREFILTER));
    lay2 = lay;
    lay_pre = lay;
  }

  // '^'-anchored DFA plan: the anchored automaton's run from byte 0 is the whole answer
  int anchored() {
  int32_t* d_se = nullptr;
    HIP_TRY(scratch_alloc((void**)&d_se, sizeof(int32_t) * 2 * n, s));
    {
      ScanTimer tm(s);
      launch_stream<ST_FIRST>(h, lay, n, nullptr, nullptr, nullptr, 0, d_se, d_se + n, s, lay.vlen, lay.vskip);
      HIP_TRY(hipGetLastError());
      tm.stop();
    }
    hipLaunchKernelGGL(k_first_to_counts, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, d_se, d_counts);
    if (int rc = device_scan<int32_t>(d_counts, n, d_prefix, d_total, s)) return rc;
    if (span_cap > 0)
      hipLaunchKernelGGL(k_first_to_spans, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, d_se, d_se + n, d_prefix, d_spans,
                         span_cap);
    HIP_TRY(hipGetLastError());
    g_last_kernel = "k_stream_first_findall";
    int rc = MRX_OK;
    if (total) {
      int64_t tot = 0;
      HIP_TRY(hipMemcpyAsync(&tot, d_total, sizeof tot, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      *total = tot;
      if (tot > span_cap) rc = fail(MRX_E_CAPACITY, "span buffer too small: need " + std::to_string(tot));
    }
    return rc;
  }

  // streamable plan: the scan (records, or one of the one-launch forms)
  int stream_scan() {
  const int64_t nw = (n + 63) / 64;
    size_t nrec;
    if (lay.offsets) {
      // the record buffer is sized by the batch's byte count, which only the device knows
      // (csr_stats above: the only host synchronisation the CSR path adds)
      nrec = (size_t)(csr_total / 16 + 256 * nw + 256);
    } else {
      // one 16-byte record per 16-byte group at most (+1 for the match that ends at len)
      rec_row = rec_row_len(lay.lens ? lay.stride : lay.len) + (strided_fast(lay) ? 0 : 1);  // frame: one more group
      nrec = (size_t)rec_row * n;
    }
    max_text = lay.offsets ? csr_max : (lay.lens ? lay.stride : (int64_t)lay.len);
    rec32 = max_text <= kRec32MaxLen;
    // (a CSR batch of equal-length texts leaves no lane idle: the 64-text wavefronts and their decode are faster)
    dyn = dyn_ok(h, lay, n) && max_text < (int64_t(1) << kDynShift) &&
          (g_dyn_mode == 1 || csr_total < max_text * n - max_text * n / 8);
    // Event rows: texts of 2 KiB and more at an aligned fixed pitch, one common length that is a multiple of the chunk
    // (1 KiB texts were measured with sixteen lanes per text: config 4 0.462 -> 0.502 ms, config 2 0.363 -> 0.471 -- per
    // text the prefix sums and the decode's fixed work outweigh the smaller intermediate; profiles/r04_config5_rows.md)
    // (no byte behind a text is ever walked: the match that ends with the text is the count's business), 16-bit positions.
    rows_shape = !dyn && !lay.offsets && !lay.lens && strided_fast(lay) && span_cap > 0 && !g_split_findall && rec32 &&
                 max_text >= 2048 && max_text <= 65024 && max_text % MRX_STREAM_CHUNK == 0 && g_dense_rows != 2;
    rows = rows_shape && (g_dense_rows == 1 || (g_fused != 2 && dense_probe_read()));   // (a forced one-launch form stays forced)
    // One launch (ST_FUSED) when a record region per resident wavefront -- sized for the most one
    // 64-text task can produce -- stays within twice the record stream of the three-launch form
    // (ragged batches whose longest text is far above the average do not: they are cut into pieces
    // above, or keep the stream that is sized by the batch's byte count).
    const int64_t fz_per_text = (rec32 ? max_text / 32 : max_text / 16) + 4;
    int64_t fz_grid = (nw + kStreamWaves - 1) / kStreamWaves;
    if (fz_grid > fused_grid_cap()) fz_grid = fused_grid_cap();
    const size_t fz_nrec = (size_t)(64 * fz_per_text + 64) * (size_t)(fz_grid * kStreamWaves) * 2;   // two regions per wavefront
    const int64_t batch_bytes = lay.offsets ? csr_total : n * (lay.lens ? lay.stride : (int64_t)lay.len);
    fused = !dyn && !rows && g_fused && span_cap > 0 && fz_nrec <= 2 * nrec + (size_t(8) << 20) && (g_fused == 2 || batch_bytes >= nw * kFusedMinTaskBytes);
    // texts of at most 1 KiB at a 16-byte aligned pitch, automaton in registers: one launch, no records at all
    // (mrx_stream_bits.hip)
    const bool bits = !dyn && !fused && !rows && !lay.offsets && span_cap > 0 && !g_split_findall &&
                      stream_bits_eligible(p, lay.data, lay.stride, max_text, n);
    if (rows) {
      row_pairs = ((max_text / 32) + 1) & ~int64_t(1);
      HIP_TRY(scratch_alloc((void**)&d_rows, sizeof(uint2) * (size_t)row_pairs * (size_t)n, s));
      ScanTimer tm(s);
      launch_stream<ST_ROWS>(h, lay, n, d_counts, nullptr, (EvRec*)d_rows, row_pairs, nullptr, nullptr, s);
      g_last_kernel = "k_stream_findall_rows";
      HIP_TRY(hipGetLastError());
      tm.stop();
    } else
    if (bits) {
      void* d_args = nullptr;
      HIP_TRY(scratch_alloc((void**)&d_ctrl, sizeof(unsigned long long) * stream_bits_ctrl_words(n), s));
      HIP_TRY(scratch_alloc(&d_args, stream_bits_args_bytes(), s));
      if (int rc = stream_bits_init(n, max_text, d_prefix, d_spans, span_cap, d_total, d_ctrl, d_args, s)) return rc;
      ScanTimer tm(s);
      if (int rc = stream_bits_scan(p, H_BLOB(h), lay.data, lay.stride, lay.lens, lay.len, max_text, n, d_args, s)) return rc;
      g_last_kernel = "k_stream_bits";
      tm.stop();
      fused = true;   // (offsets and spans are complete: nothing is left for the launches below)
    } else
    if (fused) {
      // ticket | error | one descriptor per task | two words per group of 64 tasks; a 16-byte multiple
      const size_t ctrl_words = (size_t)((2 + nw + 2 * ((nw + 63) / 64) + 1) & ~int64_t(1));
      HIP_TRY(scratch_alloc((void**)&d_ctrl, sizeof(unsigned long long) * ctrl_words, s));   // first: starts its own 256-byte block
      HIP_TRY(scratch_alloc((void**)&d_recs, sizeof(EvRec) * fz_nrec, s));
      FusedArgs* d_fz = nullptr;
      HIP_TRY(scratch_alloc((void**)&d_fz, sizeof(FusedArgs), s));
      FusedArgs fz;
      fz.ctrl = d_ctrl; fz.prefix = d_prefix; fz.spans = d_spans; fz.span_cap = span_cap; fz.total_out = d_total;
      fz.rec_cap = 64 * fz_per_text + env_knobs().fused_skew;
      fz.debug = env_knobs().fused_debug;
      hipLaunchKernelGGL(k_fused_init, dim3((unsigned)((ctrl_words + kBlock * 8 - 1) / (kBlock * 8))), dim3(kBlock), 0, s,
                         d_ctrl, (int64_t)ctrl_words, d_fz, fz);
      ScanTimer tm(s);
      launch_stream<ST_FUSED>(h, lay, n, nullptr, nullptr, d_recs, 0, nullptr, nullptr, s, nullptr, nullptr, rec32, d_fz,
                              (int)fz_grid);
      g_last_kernel = "k_stream_findall_fused";
      HIP_TRY(hipGetLastError());
      tm.stop();
    } else if (dyn) {
      const int64_t nt = (n + kDynTexts - 1) / kDynTexts;
      HIP_TRY(scratch_alloc((void**)&d_recs, sizeof(EvRec) * (size_t)(csr_total / 16 + 4 * kDynTexts * (nt + 1)), s));
      HIP_TRY(scratch_alloc((void**)&d_nrecs, sizeof(int32_t) * 2 * nt, s));   // records | matches per task
      HIP_TRY(scratch_alloc((void**)&d_wbase, sizeof(int64_t) * (nt + 1), s));
      ScanTimer tm(s);
      launch_stream_dyn<ST_RECORDS>(h, lay, n, d_counts, d_nrecs, d_recs, nullptr, nullptr, s, rec32);
      g_last_kernel = "k_stream_findall_dyn";
      HIP_TRY(hipGetLastError());
      tm.stop();
    } else if (g_split_findall && !lay.offsets && strided_fast(lay) && span_cap > 0 && n >= kSplitMinTexts) {
      split_done = true;
      if (int rc = findall_split(h, lay, n, d_counts, d_prefix, d_spans, span_cap, d_total, rec_row, rec32, max_text <= 65535, s))
        return rc;
    } else {
    const int64_t skew = g_rec_skew.load(std::memory_order_relaxed);   // (mrx_debug_rec_skew: placement experiments)
    HIP_TRY(scratch_alloc((void**)&d_recs_alloc, sizeof(EvRec) * nrec + (size_t)skew, s));
    d_recs = (EvRec*)((uint8_t*)d_recs_alloc + skew);
    HIP_TRY(scratch_alloc((void**)&d_nrecs, sizeof(int32_t) * 2 * nw, s));  // records | matches per wavefront
    HIP_TRY(scratch_alloc((void**)&d_wbase, sizeof(int64_t) * (nw + 1), s));
    ScanTimer tm(s);
    launch_stream<ST_RECORDS>(h, lay, n, d_counts, d_nrecs, d_recs, rec_row, nullptr, nullptr, s, nullptr, nullptr, rec32);
    g_last_kernel = "k_stream_findall";
    HIP_TRY(hipGetLastError());
    tm.stop();
    }
    return MRX_OK;
  }

  // ... and what turns its records into the CSR: prefix sums over the wavefronts' totals, k_decode
  int stream_finish() {
    if (rows) {   // offsets from the scan's counts, then a wavefront per text (k_decode_rows)
      if (int rc = device_scan<int32_t>(d_counts, n, d_prefix, d_total, s)) return rc;
      // lanes per text by length: a round of 2 G pairs covers 64 G bytes
      const int G = max_text <= 1024 ? 16 : max_text <= 2048 ? 32 : 64;
      const int64_t per_block = (kBlock / 64) * (64 / G);
      const int64_t blocks = (n + per_block - 1) / per_block;
      const dim3 dg((unsigned)(blocks < 8 * grid_cap() ? blocks : 8 * grid_cap())), db(kBlock);
#define MRX_DR(GG) hipLaunchKernelGGL(k_decode_rows<GG>, dg, db, 0, s, n, (const uint2*)d_rows, row_pairs, (int)(max_text / 32), \
                                      (int)max_text, (const int32_t*)d_counts, (const int64_t*)d_prefix, d_spans, span_cap,      \
                                      p.st_fixed_len)
      if (G == 16) MRX_DR(16); else if (G == 32) MRX_DR(32); else MRX_DR(64);
#undef MRX_DR
      HIP_TRY(hipGetLastError());
      return MRX_OK;
    }
  // prefix sums over the wavefronts' totals only (n/64 values); k_decode derives the per-text
    // offsets from its 64 counts and writes them along with the spans
    const int64_t nw = dyn ? (n + kDynTexts - 1) / kDynTexts : (n + 63) / 64;
    // one launch: tile-local exclusive sums of the wavefront totals + one sum per tile
    const int64_t ntiles = (nw + kScanTile - 1) / kScanTile;
    int64_t* d_tsum = nullptr;
    HIP_TRY(scratch_alloc((void**)&d_tsum, sizeof(int64_t) * ntiles, s));
    hipLaunchKernelGGL(k_scan_local<int32_t>, dim3((unsigned)ntiles), dim3(kScanBlock), 0, s, d_nrecs + nw, nw,
                       d_wbase, d_tsum);
    const bool pack16 = max_text <= 65535;
    if (dyn) {
      const dim3 dg((unsigned)grid_for(nw * 64, kBlock) * 2), db(kBlock);
      if (pack16 && rec32)
        hipLaunchKernelGGL((k_decode<true, false, true, true, MRX_DYN_DECODE_TILE>), dg, db, 0, s, n, d_nrecs, d_recs, rec_row, lay.offsets, d_counts,
                           d_wbase, d_tsum, d_prefix, d_spans, span_cap, p.st_fixed_len, d_total);
      else if (pack16)
        hipLaunchKernelGGL((k_decode<true, false, false, true>), dg, db, 0, s, n, d_nrecs, d_recs, rec_row, lay.offsets, d_counts,
                           d_wbase, d_tsum, d_prefix, d_spans, span_cap, p.st_fixed_len, d_total);
      else
        hipLaunchKernelGGL((k_decode<false, false, false, true>), dg, db, 0, s, n, d_nrecs, d_recs, rec_row, lay.offsets, d_counts,
                           d_wbase, d_tsum, d_prefix, d_spans, span_cap, p.st_fixed_len, d_total);
    } else
    if (pack16 && rec32 && max_text >= 768)
      hipLaunchKernelGGL((k_decode<true, false, true, false, 3072>), dim3(env_knobs().decode_grid > 0 ? env_knobs().decode_grid : grid_for(n, kBlock) * 2), dim3(kBlock), 0, s, n, d_nrecs, d_recs,
                         rec_row, lay.offsets, d_counts, d_wbase, d_tsum, d_prefix, d_spans, span_cap, p.st_fixed_len,
                         d_total, (const int32_t*)nullptr, (const int64_t*)nullptr, env_knobs().decode_reverse);
    else if (pack16 && rec32)
      hipLaunchKernelGGL((k_decode<true, false, true>), dim3(grid_for(n, kBlock) * 2), dim3(kBlock), 0, s, n, d_nrecs, d_recs,
                         rec_row, lay.offsets, d_counts, d_wbase, d_tsum, d_prefix, d_spans, span_cap, p.st_fixed_len,
                         d_total);
    else if (pack16)
      hipLaunchKernelGGL(k_decode<true>, dim3(grid_for(n, kBlock) * 2), dim3(kBlock), 0, s, n, d_nrecs, d_recs,
                         rec_row, lay.offsets, d_counts, d_wbase, d_tsum, d_prefix, d_spans, span_cap, p.st_fixed_len,
                         d_total);
    else
      hipLaunchKernelGGL(k_decode<false>, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, d_nrecs, d_recs,
                         rec_row, lay.offsets, d_counts, d_wbase, d_tsum, d_prefix, d_spans, span_cap, p.st_fixed_len,
                         d_total);
    HIP_TRY(hipGetLastError());
    HIP_TRY(scratch_free(d_tsum, s));
    return MRX_OK;
  }

  // ---- PF_MW_TRIES plan: the pending-tries walk or round 3's marks + stepper?  (mrx_handle::req_tune, kind bit 30) ----
  // The one-pass walk wins where matches are found (3-10 x), marks alone are faster where nothing matches, and bigger
  // tables sit in between (tools/r04_tries_cap.py): as for the required-byte routes the handle measures -- the first four
  // eligible calls of a batch shape take the routes alternately, the last two timed; no call waits.
  static uint32_t tries_tune_key(const Layout& lay, int64_t n) {
    const int64_t bytes_per_text = lay.offsets ? 0 : (lay.lens ? lay.stride : (int64_t)lay.len);
    uint32_t lb = 0, nb = 0;
    for (int64_t v = bytes_per_text; v > 1; v >>= 1) ++lb;
    for (int64_t v = n; v > 1; v >>= 1) ++nb;
    return (1u << 30) | (lay.offsets ? 1u << 31 : 0u) | ((uint32_t)(t_dev & 63) << 16) | (lb << 8) | nb;
  }
  void tries_route_tuner(bool* use_tries) {
    int slot = -1;
    const uint32_t key = tries_tune_key(lay, n);
    *use_tries = ab_tuner_begin(h, key, s, "findall", &slot) == 1;
    if (slot >= 0) { tune_key = key; tune_slot = slot; tune_route = *use_tries ? 1 : 2; }
  }

  // ---- required-byte plan on long texts: which route?  (mrx_handle::req_tune) ----
  // First call of a batch shape: are the plan's synchronising bytes frequent in the text at all (one small kernel, one
  // 4-byte read-back; else: the wavefront kernel, for good).  Then four measured calls (mrx_handle::ReqTune), and the
  // faster route from the first call that finds the last measurement complete.
  int req_route_tuner(bool* pieces) {
    *pieces = false;
    const int64_t bytes_per_text = lay.offsets ? 0 : (lay.lens ? lay.stride : (int64_t)lay.len);
    uint32_t lb = 0, nb = 0;
    for (int64_t v = bytes_per_text; v > 1; v >>= 1) ++lb;
    for (int64_t v = n; v > 1; v >>= 1) ++nb;
    tune_key = (lay.offsets ? 1u << 31 : 0u) | ((uint32_t)(t_dev & 63) << 16) | (lb << 8) | nb;
    bool probe = false;
    {
      std::lock_guard<std::mutex> lk(h->tune_mu);
      mrx_handle::ReqTune& t = h->req_tune[tune_key];
      if (t.choice) { *pieces = t.choice == 2; return MRX_OK; }
      if (t.issued == 4) {   // all four are enqueued: through?
        if (hipEventQuery(t.ev[3][1]) != hipSuccess) return MRX_OK;   // not yet: the default route, unmeasured
        if (hipEventElapsedTime(&t.ms_wave, t.ev[2][0], t.ev[2][1]) != hipSuccess ||
            hipEventElapsedTime(&t.ms_pieces, t.ev[3][0], t.ev[3][1]) != hipSuccess) { t.choice = 1; return MRX_OK; }
        t.choice = t.ms_pieces < t.ms_wave ? 2 : 1;
        static const bool verbose = getenv("MRX_TUNE_VERBOSE") != nullptr;
        if (verbose) fprintf(stderr, "mrx: required-byte route of '%s' (key %08x): wavefront %.3f ms, pieces %.3f ms -> %s\n",
                             h->hp.pattern.c_str(), tune_key, t.ms_wave, t.ms_pieces, t.choice == 2 ? "pieces" : "wavefront");
        *pieces = t.choice == 2;
        return MRX_OK;
      }
      probe = !t.probed;
      t.probed = true;
    }
    if (probe) {
      bool freq = false;
      if (int rc = sync_bytes_frequent(h, lay, n, s, &freq)) return rc;
      std::lock_guard<std::mutex> lk(h->tune_mu);
      mrx_handle::ReqTune& t = h->req_tune[tune_key];
      if (!freq) { t.choice = 1; return MRX_OK; }
      t.dev = t_dev;
      for (auto& pr : t.ev)
        for (auto& e : pr)
          if (hipEventCreate(&e) != hipSuccess) { t.choice = 1; return MRX_OK; }
    }
    std::lock_guard<std::mutex> lk(h->tune_mu);
    mrx_handle::ReqTune& t = h->req_tune[tune_key];
    if (t.choice || t.issued >= 4 || !t.ev[3][1]) return MRX_OK;   // (another thread got here first, or is still probing)
    tune_slot = t.issued++;
    tune_route = (tune_slot & 1) ? 2 : 1;
    *pieces = tune_route == 2;
    HIP_TRY(hipEventRecord(t.ev[tune_slot][0], s));
    return MRX_OK;
  }
  void req_tuner_unavailable() {   // nothing to cut: the wavefront kernel it is
    std::lock_guard<std::mutex> lk(h->tune_mu);
    h->req_tune[tune_key].choice = 1;
    tune_route = 0;
  }
  void req_tuner_report() {   // the call's work is enqueued: close the measurement
    if (!tune_route) return;
    std::lock_guard<std::mutex> lk(h->tune_mu);
    mrx_handle::ReqTune& t = h->req_tune[tune_key];
    if (t.ev[tune_slot][1]) (void)hipEventRecord(t.ev[tune_slot][1], s);
    tune_route = 0;
  }

  // every other plan, first stage: counts (+ slot rows) on the stepper family / the literal restatement
  int step_scan() {
  bool req_pieces = false;   // required-byte plan: pieces on the multi-walk kernel instead of the wavefront-per-text kernel
    if (use_req_route && mwalk_req && g_long_text_mode == 0 && !t_in_pieces && p.st_nsync > 0 && span_cap > 0 &&
        !(p.flags & (PF_STEP_BIG | PF_STREAMABLE)))
      if (int rc = req_route_tuner(&req_pieces)) return rc;
    if (step_ok && !wstep_bits && !wstep_empty && !t_in_pieces && !(p.flags & PF_STEP_BIG) && p.st_nsync > 0 &&
        !(p.flags & PF_STREAMABLE) && span_cap > 0 && (!use_req_route || g_long_text_mode == 1 || req_pieces)) {
      Pieces spc;
      if (int rc = pieces_prepare(h, lay, n, s, &spc, -1, -1, /*disjoint=*/true, wstep_mwalk)) return rc;
      if (req_pieces && !spc.on) req_tuner_unavailable();   // nothing to cut (texts too short, too many of them): the wavefront kernel it is
      if (spc.on && !wstep_mwalk) {   // (a multi-walk plan scans every piece once, dense candidates or not)
        bool dense = true;
        if (int rc = dense_candidates(h, lay, n, s, &dense)) return rc;
        if (!dense)
          if (int rc = pieces_release(&spc, s)) return rc;
      }
      if (spc.on) {
        int64_t* d_vprefix = nullptr;
        HIP_TRY(scratch_alloc((void**)&d_vprefix, sizeof(int64_t) * (spc.nv + 1), s));
        t_in_pieces = true;
        t_piece_vbase = spc.vbase;
        t_piece_base_applied = false;
        t_piece_tries = (p.flags & PF_MW_TRIES) ? (mw_tries ? 1 : 0) : -1;
        const int rc = run_findall(h, spc.lay, spc.nv, d_vprefix, d_spans, span_cap, nullptr, s, match_next_sequence);
        t_in_pieces = false;
        t_piece_vbase = nullptr;
        t_piece_tries = -1;
        if (rc != MRX_OK) return rc;
        const std::string inner = g_last_kernel;
        hipLaunchKernelGGL(k_virt_prefix, dim3(grid_for(n + 1, kBlock)), dim3(kBlock), 0, s, n, spc.vfirst, d_vprefix, d_prefix);
        if (!t_piece_base_applied)
          hipLaunchKernelGGL(k_virt_add_base, dim3(grid_cap()), dim3(kBlock), 0, s, spc.nv, d_vprefix, spc.vbase,
                             d_spans, span_cap);
        HIP_TRY(hipGetLastError());
        static thread_local std::string piece_name;
        piece_name = inner + "_pieces";
        g_last_kernel = piece_name.c_str();
        req_tuner_report();
        int rc2 = MRX_OK;
        if (total) {
          int64_t tot = 0;
          HIP_TRY(hipMemcpyAsync(&tot, d_vprefix + spc.nv, sizeof tot, hipMemcpyDeviceToHost, s));
          HIP_TRY(hipStreamSynchronize(s));
          *total = tot;
          if (tot > span_cap) rc2 = fail(MRX_E_CAPACITY, "span buffer too small: need " + std::to_string(tot));
        }
        HIP_TRY(scratch_free(d_vprefix, s));
        if (int rc3 = pieces_release(&spc, s)) return rc3;
        finished = true;   // the pieces answered the whole call
        finished_rc = rc2;
        return MRX_OK;
      }
    }
    if (step_ok && !wstep_bits && !wstep_empty && !mw_empty && !mw_tries && !t_in_pieces && !wstep_lz) {
      if (int rc = req_wave_pays(lay, n, use_req_route, s, &req_wave, (p.flags & PF_STEP_BIG) ? nullptr : &step_split,
                                 (p.flags & PF_STEP_BIG) != 0, wstep_mwalk, backset_on(p) && !wstep_mwalk))
        return rc;
      wstep_mwalk_pk = wstep_mwalk && mwalk_pk_ok(lay, t_csr_max_len);
    }
    lay2.split = step_split;
    wstep_bm = step_ok && backset_on(p) && !wstep_mwalk && !wstep_bits && !wstep_empty && !use_req_route && !req_wave &&
               step_split == 0;
    wstep_bm_big = wstep_bm && (p.flags & PF_STEP_BIG) != 0;
    if (wstep_bm) {
      const int32_t split_keep = lay2.split;
      if (int rc = backscan_marks(h, lay, n, s, &lay2)) return rc;
      lay2.split = split_keep;
    }
    if (step_ok && !bits_fixed && !wstep_empty && !wstep_mwalk && !wstep_bm && (wstep_bits || (!req_wave && step_split == 0 && !use_req_route && union_pass_for_table_plan(p, false)))) {
      // union automaton first: texts in which no walk from any start reaches MATCH are not walked at all
      // (mode 0: a wavefront stops as soon as each of its texts has shown one match end, so on texts full
      // of matches the pass costs next to nothing; cutting tails -- mode 1 -- would scan everything)
      if (int rc = bscan_limits(h, lay, n, 0, s, &lay2, &d_blimit)) return rc;
    }
    // big tables: only the wavefront kernel has their form; many short texts stay on the literal restatement
    if ((p.flags & PF_STEP_BIG) && !req_wave && !wstep_mwalk && !wstep_bm) step_ok = false;
    ScanTimer tm(s);
    // multi-walk plans: count, prefix sums, emit -- two one-pass scans whatever the match density (the count pass
    // keeps no start registers and runs at 3 TB/s; slot rows + a second walk for overflowing texts would be three)
    mwalk_two_pass = wstep_mwalk && !req_wave && step_split == 0;
    if (mwalk_two_pass && t_in_pieces && t_piece_vbase) {   // every span of this call leaves k_mwalk<STEP_EMIT>
      lay2.vbase = t_piece_vbase;
      t_piece_base_applied = true;
    }
    if (bits_fixed) {
      // matches of at least four bytes: one pass, spans into slot rows of len / 4 + 32 (they cannot overflow),
      // gathered behind the prefix sums; shorter ones: count, prefix sums, the pass once more to emit
      bits_fixed_slots = p.bs_fixed_len >= 4 && span_cap > 0 && !lay.vlen;
      if (bits_fixed_slots) {
        int64_t bytes = 0;
        if (lay.offsets) {
          HIP_TRY(hipMemcpyAsync(&bytes, lay.offsets + n, sizeof bytes, hipMemcpyDeviceToHost, s));
          HIP_TRY(hipStreamSynchronize(s));
        } else {
          bytes = n * (lay.lens ? lay.stride : (int64_t)lay.len);
        }
        lay2.wide_slots = 1;
        HIP_TRY(scratch_alloc((void**)&d_slots, sizeof(int32_t) * 2 * (size_t)(bytes / 4 + 32 * n + 64), s));
      }
      if (int rc = bscan_fixed(h, lay, n, bits_fixed_slots ? 5 : 2, s, d_counts, d_slots, nullptr, 0)) return rc;
    } else if (step_ok && span_cap > 0 && !wstep_empty && !mwalk_two_pass) {
      if (req_wave) {
        // long texts: rows of len / 4 + 32 slots (twice the bytes of the batch) -- the second walk
        // is then only for texts with a match every 4 bytes
        int64_t bytes = 0;
        if (lay.offsets) {
          HIP_TRY(hipMemcpyAsync(&bytes, lay.offsets + n, sizeof bytes, hipMemcpyDeviceToHost, s));
          HIP_TRY(hipStreamSynchronize(s));
        } else {
          bytes = n * (lay.lens ? lay.stride : (int64_t)lay.len);
        }
        lay2.wide_slots = 1;
        HIP_TRY(scratch_alloc((void**)&d_slots, sizeof(int32_t) * 2 * (size_t)(bytes / 4 + 32 * n + 64), s));
      } else if (!lay.offsets && step_split == 0 && (lay.lens ? lay.stride : (int64_t)lay.len) >= 2048) {
        // one lane per text, but texts long enough to hold more than kStepSlots matches as a rule
        // (rows sized as Layout::slot_row sizes them for lay2 -- the bitset first pass gives it per-text lengths)
        lay2.wide_slots = 1;
        HIP_TRY(scratch_alloc((void**)&d_slots,
                              sizeof(int32_t) * 2 * (size_t)(n * ((lay2.lens ? lay2.stride : (int64_t)lay2.len) / 4 + 32) + 64), s));
      } else
      HIP_TRY(scratch_alloc((void**)&d_slots, sizeof(int32_t) * 2 * kStepSlots * (size_t)n, s));
      if (req_wave)
        MRX_REQWAVE_LAUNCH(STEP_SLOTS, h, lay2, n, d_counts, (const int64_t*)nullptr, d_slots, (int64_t)0, s);
      else
      {
      MRX_WSTEP_LAUNCH(STEP_SLOTS, dim3(wstep_grid(n)), dim3(64 * kWsWaves), wstep_lds(pk, wstep_mwalk, wstep_bm_big), s, pk,
                         H_BLOB(h), lay2, n, d_counts, (const int64_t*)nullptr, d_slots, (int64_t)0,
                         (int32_t*)nullptr, (int32_t*)nullptr);
      if (step_split > 0)
        MRX_REQWAVE_LAUNCH(STEP_SLOTS, h, lay2, n, d_counts, (const int64_t*)nullptr, d_slots, (int64_t)0, s);
      }
    } else if (step_ok && req_wave)
      MRX_REQWAVE_LAUNCH(STEP_COUNT, h, lay, n, d_counts, (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0, s);
    else if (step_ok) {
      MRX_WSTEP_LAUNCH(STEP_COUNT, dim3(wstep_grid(n)), dim3(64 * kWsWaves), wstep_lds(pk, wstep_mwalk, wstep_bm_big), s, pk,
                         H_BLOB(h), lay2, n, d_counts, (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0,
                         (int32_t*)nullptr, (int32_t*)nullptr);
      if (step_split > 0)
        MRX_REQWAVE_LAUNCH(STEP_COUNT, h, lay2, n, d_counts, (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0, s);
    } else {
      if (p.flags & PF_BT_SEARCH)
        if (int rc = bt_prepass(h, lay, n, s, &lay_pre)) return rc;
  #define MRX_L(B) hipLaunchKernelGGL((k_findall<FA_COUNT, B>), dim3(grid_for(n, kBlock)), dim3(kBlock), lds_for(h), s, \
                                 p, H_BLOB(h), lay_pre, n, d_counts, (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0)
      MRX_BT_DISPATCH(bt_kernel_kind(h, plan_uses_backtracker(h)), MRX_L);
  #undef MRX_L
    }
    g_last_kernel = bits_fixed ? "k_bscan_fixed" : req_wave ? "k_req_wave" : step_ok ? (wstep_mwalk ? (step_split > 0 ? "k_mwalk+k_req_wave" : "k_mwalk") : wstep_bm ? "k_backscan+k_step_count" : wstep_bits ? "k_bstep_count" : wstep_empty ? "k_estep_count" : step_split > 0 ? "k_step_count+k_req_wave" : "k_step_count")
                                                      : "k_findall_count";
    HIP_TRY(hipGetLastError());
    tm.stop();
    return MRX_OK;
  }

  // ... second stage, behind the prefix sums: spans to their CSR place
  int step_finish() {
  if (bits_fixed && bits_fixed_slots) {
      hipLaunchKernelGGL(k_slots_gather_wide, dim3(grid_for(n * 64, kBlock)), dim3(kBlock), 0, s, lay2, n, d_counts,
                         d_prefix, d_slots, d_spans, span_cap);
    } else if (bits_fixed) {   // the union pass once more, texts that hold a match: spans straight to their CSR place
      if (int rc = bscan_fixed(h, lay, n, 3, s, d_counts, d_spans, d_prefix, span_cap)) return rc;
    } else if (step_ok && mwalk_two_pass) {   // second scan, texts that hold a match: spans straight to their CSR place
      Layout lay_e = lay2;
      lay_e.wide_slots = 2;
      MRX_WSTEP_LAUNCH(STEP_EMIT, dim3(wstep_grid(n)), dim3(64 * kWsWaves), wstep_lds(pk, wstep_mwalk, wstep_bm_big), s, pk, H_BLOB(h),
                       lay_e, n, d_counts, d_prefix, d_spans, span_cap, (int32_t*)nullptr, (int32_t*)nullptr);
    } else if (step_ok && wstep_empty) {   // second walk, every text: spans straight to their CSR place
      MRX_WSTEP_LAUNCH(STEP_EMIT, dim3(wstep_grid(n)), dim3(64 * kWsWaves), wstep_lds(pk, wstep_mwalk, wstep_bm_big), s, pk, H_BLOB(h),
                       lay2, n, (int32_t*)nullptr, d_prefix, d_spans, span_cap, (int32_t*)nullptr, (int32_t*)nullptr);
    } else if (step_ok) {
      if (lay2.wide_slots)
        hipLaunchKernelGGL(k_slots_gather_wide, dim3(grid_for(n * 64, kBlock)), dim3(kBlock), 0, s, lay2, n, d_counts,
                           d_prefix, d_slots, d_spans, span_cap);
      else
      hipLaunchKernelGGL(k_slots_gather, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, d_counts, d_prefix,
                         d_slots, d_spans, span_cap);
      // wavefronts without an overflowing text leave at once
      if (req_wave)
        MRX_REQWAVE_LAUNCH(STEP_EMIT, h, lay2, n, d_counts, d_prefix, d_spans, span_cap, s);
      else {
      MRX_WSTEP_LAUNCH(STEP_EMIT, dim3(wstep_grid(n)), dim3(64 * kWsWaves), wstep_lds(pk, wstep_mwalk, wstep_bm_big), s, pk, H_BLOB(h),
                         lay2, n, d_counts, d_prefix, d_spans, span_cap, (int32_t*)nullptr,
                         (int32_t*)nullptr);
      if (step_split > 0) MRX_REQWAVE_LAUNCH(STEP_EMIT, h, lay2, n, d_counts, d_prefix, d_spans, span_cap, s);
      }
    } else
  #define MRX_L(B) hipLaunchKernelGGL((k_findall<FA_EMIT, B>), dim3(grid_for(n, kBlock)), dim3(kBlock), lds_for(h), s, p, \
                                 H_BLOB(h), lay_pre, n, (int32_t*)nullptr, d_prefix, d_spans, span_cap)
      MRX_BT_DISPATCH(bt_kernel_kind(h, plan_uses_backtracker(h)), MRX_L);
  #undef MRX_L
    HIP_TRY(hipGetLastError());
    return MRX_OK;
  }

  // the total (one stream synchronisation when the caller asked for it)
  // mrx_handle::dense_probe: was the handle's last eligible batch full of matches?  (Takes a total that has arrived.)
  static constexpr int64_t kDenseBytesPerSpan = 20;
  bool dense_probe_read() {
    std::lock_guard<std::mutex> lk(h->tune_mu);
    mrx_handle::DenseProbe& d = h->dense_probe;
    if (d.pending && d.dev == t_dev && hipEventQuery(d.ev) == hipSuccess) {
      d.dense = (*d.host_total > 0 && *d.host_total * kDenseBytesPerSpan >= d.bytes) ? 1 : 0;
      d.pending = false;
    }
    return d.dense == 1;
  }
  // ... and this call's total on its way (nothing in flight and the shape eligible: else the older answer stands)
  void dense_probe_send() {
    if (!rows_shape || g_dense_rows != 0) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;   // (a stream under graph capture: no event of ours in the graph)
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
    std::lock_guard<std::mutex> lk(h->tune_mu);
    mrx_handle::DenseProbe& d = h->dense_probe;
    if (d.pending) return;
    if (!d.host_total) {
      if (hipHostMalloc((void**)&d.host_total, sizeof(int64_t)) != hipSuccess) { d.host_total = nullptr; return; }
      if (hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) != hipSuccess) { d.ev = nullptr; return; }
      d.dev = t_dev;
    }
    if (!d.ev || d.dev != t_dev) return;   // (a handle used on several devices keeps the first one's answer)
    if (hipMemcpyAsync(d.host_total, d_total, sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess) return;
    if (hipEventRecord(d.ev, s) != hipSuccess) return;
    d.bytes = n * max_text;
    d.pending = true;
  }

  int read_total() {
    req_tuner_report();
    dense_probe_send();
  int rc = MRX_OK;
    if (total) {
      int64_t tot = 0;
      unsigned long long fused_err = 0;
      HIP_TRY(hipMemcpyAsync(&tot, d_total, sizeof tot, hipMemcpyDeviceToHost, s));
      if (d_ctrl) HIP_TRY(hipMemcpyAsync(&fused_err, d_ctrl + 1, sizeof fused_err, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      *total = tot;
      if (fused_err) return fail(MRX_E_NO_DEVICE, "internal: a wavefront gave up waiting for its predecessors' span counts");
      if (tot > span_cap) rc = fail(MRX_E_CAPACITY, "span buffer too small: need " + std::to_string(tot));
    }  // total == NULL: fully asynchronous; d_counts_prefix[n] holds the total when the stream drains
    HIP_TRY(scratch_free(d_counts, s));
    HIP_TRY(scratch_free(d_total, s));
    if (d_ctrl) { HIP_TRY(scratch_free(d_ctrl, s)); HIP_TRY(scratch_free(d_ctrl, s)); }   // ctrl block and the argument copy
    if (d_recs_alloc) HIP_TRY(scratch_free(d_recs_alloc, s));
    else if (d_recs) HIP_TRY(scratch_free(d_recs, s));
    if (d_nrecs) HIP_TRY(scratch_free(d_nrecs, s));
    if (d_wbase) HIP_TRY(scratch_free(d_wbase, s));
    if (d_slots) HIP_TRY(scratch_free(d_slots, s));
    if (d_blimit) HIP_TRY(scratch_free(d_blimit, s));
    if (d_rows) HIP_TRY(scratch_free(d_rows, s));
    return rc;
  }

  int run() {
    HIP_TRY(scratch_alloc((void**)&d_counts, sizeof(int32_t) * (n > 0 ? n : 1), s));
    HIP_TRY(scratch_alloc((void**)&d_total, sizeof(int64_t), s));
    if (n > 0 && anchored_at_zero(h) && stream_layout_ok(lay, n)) return anchored();
    choose_route();
  if (n > 0 && stream_ok && lay.offsets) {
      if (known_total >= 0) { csr_total = known_total; csr_max = known_max; }   // the caller (sub) has read them
      else if (int rc = csr_stats(lay, n, s, &csr_total, &csr_max)) return rc;
      if (csr_total < 0) return fail(MRX_E_ARGUMENT, "offsets[n] is negative");
    }
    // An event record counts the matches of its text in front of it in 26 bits (kRecBeforeMask), and a
    // text of 2^26 bytes can hold that many (one-byte matches, no synchronising byte to cut at): such
    // texts take the lane-per-text kernels, whose span cursor is 64 bits wide.
    if (stream_ok && stream_text_too_long(lay.offsets ? csr_max : (lay.lens ? lay.stride : (int64_t)lay.len)))
      stream_ok = false;
    if (n > 0 && stream_ok)
      if (int rc = pieces_prepare(h, lay, n, s, &pc, csr_total, csr_max)) return rc;
    by_pieces = pc.on;
    if (pc.on) {
      if (int rc = findall_pieces(h, pc, n, d_prefix, d_spans, span_cap, d_total, s)) return rc;
      if (int rc = pieces_release(&pc, s)) return rc;
    } else if (n > 0) {
      if (int rc = stream_ok ? stream_scan() : step_scan()) return rc;
      if (finished) return finished_rc;
    }
    if (by_pieces || fused || split_done) {
      // done over the pieces above / by the one launch / in two halves (findall_split)
    } else if (stream_ok) {
      if (int rc = stream_finish()) return rc;
    } else if (int rc = device_scan<int32_t>(d_counts, n, d_prefix, d_total, s)) return rc;
    // Second stage is enqueued before the total is known on the host: both kernels clip
    // at span_cap, so a too-small buffer is reported (MRX_E_CAPACITY) without overrun and
    // the whole call needs a single stream synchronisation.
    if (n > 0 && span_cap > 0 && !stream_ok)
      if (int rc = step_finish()) return rc;
    return read_total();
  }
  int finished_rc = MRX_OK;
};

int run_findall(const mrx_handle* h, const Layout& lay, int64_t n, int64_t* d_prefix, int32_t* d_spans, int64_t span_cap,
                int64_t* total, void* stream, bool match_next_sequence, int64_t known_total, int64_t known_max) {
  ScratchScope scratch_scope_((hipStream_t)stream);
  if (!h) return fail(MRX_E_ARGUMENT, "null handle");
  if (n < 0 || span_cap < 0) return fail(MRX_E_ARGUMENT, "negative size");
  if ((uintptr_t)d_spans & 7) return fail(MRX_E_ARGUMENT, "d_spans must be 8-byte aligned");
  if (int rc = check_search_supported(h)) return rc;
  if (int rc = check_lds(h)) return rc;
  if (int rc = ensure_device(h)) return rc;
  FindallJob job(h, lay, n, d_prefix, d_spans, span_cap, total, (hipStream_t)stream, match_next_sequence, known_total, known_max);
  return job.run();
}

}  // namespace

namespace {
// mrx_debug_subs_group(): lanes per text in k_subs_wave (16 / 32 / 64), 0 = k_subs_emit only, -1 = by text length
std::atomic<int> g_subs_group{env_int("MRX_SUBS_G", -1)};
// regex.sub for streamable plans: streaming findall -> sizes -> prefix sums -> emit (see k_subs_*)
// A replacement that copies a fixed-width group copies text[match start + offset ...] whatever the match looked like
// (_apply_template_fixed, matcher.mojo:1592-1621; the widths come from the pattern text, quantifiers ignored:
// 'x(\\d)?' has its group at offset 1 even when the match is "x").  A match close to the end of its text can thus
// reach behind the text -- upstream reads those bytes unchecked; the oracle's slice and sub_text() stop at the end of
// the text, so the replacement is shorter than R bytes there.  The spans route has one R for every match: it
// looks at the LAST match of each text (the one that reaches furthest) and hands the call back
// (kSubsRetryGeneric) when any reaches behind its text.
constexpr int kSubsRetryGeneric = 1 << 20;
__global__ __launch_bounds__(kBlock) void k_subs_reach(int64_t n, const int64_t* __restrict__ offsets,
                                                       const int64_t* __restrict__ prefix, const int32_t* __restrict__ spans,
                                                       int64_t span_cap, int reach, int32_t* __restrict__ over) {
  if (prefix[n] > span_cap) return;   // the spans are incomplete: the caller repeats the findall
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = prefix[i], b = prefix[i + 1];
    if (b > a && (int64_t)spans[2 * (b - 1)] + reach > offsets[i + 1] - offsets[i]) *over = 1;
  }
}
// go = 1 when the assembly may run: every span had room, the output fits, there is output, no group reaches behind its text
__global__ void k_subs_gate(const int64_t* __restrict__ matches, int64_t span_cap, const int64_t* __restrict__ out_bytes,
                            int64_t out_cap, const int32_t* __restrict__ over, int32_t* __restrict__ go) {
  *go = (*matches <= span_cap && *out_bytes <= out_cap && *out_bytes > 0 && !(over && *over)) ? 1 : 0;
}
int sub_from_spans(const mrx_handle* h, const Layout& lay, int64_t n, const std::vector<uint16_t>& rmap,
                   int64_t count, int64_t* out_off, uint8_t* out, int64_t out_cap, int64_t* total_bytes,
                   hipStream_t s, int group_reach = 0, int64_t known_bytes = -1, int64_t known_max = -1) {
  // Two host synchronisations per call: the batch's byte count and longest text (sizes the span buffer
  // and, handed on, spares findall its own look; none when the caller knows them: rows at a fixed pitch), and the
  // two totals -- matches and output bytes -- behind findall, sizes and prefix sums, all enqueued without waiting.
  int64_t in_bytes = known_bytes, max_len = known_max;
  if (known_bytes < 0 || known_max < 0)
    if (int rc0 = csr_stats(lay, n, s, &in_bytes, &max_len)) return rc0;
  if (in_bytes < 0) return fail(MRX_E_ARGUMENT, "offsets[n] is negative");
  const int R = (int)rmap.size();
  int64_t* d_prefix = nullptr;
  int32_t* d_spans = nullptr;
  uint16_t* d_rmap = nullptr;
  int32_t* d_cum = nullptr;
  int64_t *d_sizes = nullptr, *d_total = nullptr;
  int32_t* d_left = nullptr;
  int32_t* d_go = nullptr;
  int32_t* d_over = nullptr;
  int32_t over = 0;
  HIP_TRY(scratch_alloc((void**)&d_over, sizeof(int32_t), s));
  if (group_reach > 0) HIP_TRY(hipMemsetAsync(d_over, 0, sizeof(int32_t), s));
  HIP_TRY(scratch_alloc((void**)&d_prefix, sizeof(int64_t) * (n + 1), s));
  HIP_TRY(scratch_alloc((void**)&d_rmap, sizeof(uint16_t) * (R + 8), s));
  HIP_TRY(scratch_alloc((void**)&d_sizes, sizeof(int64_t) * n, s));
  HIP_TRY(scratch_alloc((void**)&d_total, sizeof(int64_t), s));
  HIP_TRY(scratch_alloc((void**)&d_left, sizeof(int32_t), s));
  HIP_TRY(scratch_alloc((void**)&d_go, sizeof(int32_t), s));
  if (R) HIP_TRY(hipMemcpyAsync(d_rmap, rmap.data(), sizeof(uint16_t) * R, hipMemcpyHostToDevice, s));
  int64_t cap = in_bytes / 8 + n + 64, nm = 0, tot = 0;
  if (const int64_t seen = h->sub_matches_per_kib.load(std::memory_order_relaxed); seen > 128) {
    const int64_t by_hint = (in_bytes >> 10) * (seen + seen / 8 + 1) + n + 64;   // the last batch's density + 1/8
    if (by_hint > cap) cap = by_hint < in_bytes + n + 64 ? by_hint : in_bytes + n + 64;
  }
  int rc = MRX_OK;
  bool cum_later = false;
  int G = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    HIP_TRY(scratch_alloc((void**)&d_spans, sizeof(int32_t) * 2 * (size_t)cap, s));
    HIP_TRY(scratch_alloc((void**)&d_cum, sizeof(int32_t) * (size_t)(cap + 1), s));
    rc = run_findall(h, lay, n, d_prefix, d_spans, cap, nullptr, s, /*match_next_sequence=*/true, in_bytes, max_len);
    if (rc != MRX_OK) return rc;
    const int64_t avg0 = in_bytes / (n > 0 ? n : 1);
    // lanes per text in k_subs_wave by average text length (1 KiB texts: 32 lanes 1.41 ms, 64 lanes 1.56 ms);
    // 0 = k_subs_emit alone: long replacement templates (one lane writes a replacement), forced, or texts
    // beyond the workgroup form's tiles, which take k_subs_emit<kBlock>
    const bool long_texts = (g_long_text_mode == 1 || g_long_text_mode == 3 || avg0 > 6144) && g_long_text_mode != 2;
    const int force_g = g_subs_group;
    G = force_g >= 0 ? force_g : (avg0 > 2048 ? 256 : avg0 > 1024 ? 64 : avg0 > 224 ? 32 : 16);
    if (R > 1024 || long_texts) G = 0;
    // cum[] (matched bytes before each match) is k_subs_emit's: when k_subs_wave runs in front of it, it is
    // only computed if that kernel leaves texts over (cum_later)
    cum_later = count == 0 && G != 0;
    if (cum_later) {
      const int64_t blocks = ((n + 63) / 64 + (kBlock / 64) - 1) / (kBlock / 64);
      hipLaunchKernelGGL(k_subs_sizes_flat, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(kBlock), 0, s,
                         n, lay.offsets, d_prefix, d_spans, R, d_sizes, cap);
    } else {
      const int64_t blocks = (n + (kBlock / 16) - 1) / (kBlock / 16);
      hipLaunchKernelGGL(k_subs_sizes, dim3((unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4)), dim3(kBlock), 0, s,
                         n, lay.offsets, d_prefix, d_spans, (long long)count, R, d_sizes, d_cum, cap, (const int32_t*)nullptr);
    }
    HIP_TRY(hipGetLastError());
    rc = device_scan<int64_t>(d_sizes, n, out_off, d_total, s);
    if (rc != MRX_OK) return rc;
    if (group_reach > 0)
      hipLaunchKernelGGL(k_subs_reach, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, lay.offsets, d_prefix, d_spans, cap,
                         group_reach, d_over);
    // The assembly is enqueued BEHIND a device-side go-ahead and BEFORE the host has seen the totals: the stream does
    // not run dry while they travel (63 us of 1.7 ms on config 4).  When the go-ahead says no, the kernels return at
    // once and the host, which reads the same numbers below, retries or reports.
    hipLaunchKernelGGL(k_subs_gate, dim3(1), dim3(1), 0, s, d_prefix + n, cap, d_total, out_cap,
                       group_reach > 0 ? (const int32_t*)d_over : (const int32_t*)nullptr, d_go);
    if (G == 0 && (g_long_text_mode == 1 || g_long_text_mode == 3 || in_bytes / n >= 4096) && g_long_text_mode != 2) {   // long texts: a workgroup per text
      hipLaunchKernelGGL(k_subs_emit<kBlock>, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(kBlock),
                         (size_t)(2 * R + 16), s, n, lay.data, lay.offsets, d_prefix, d_spans, d_cum, (long long)count, R,
                         d_rmap, out_off, out, 0, (const int32_t*)d_go);
      g_last_kernel = "k_subs_emit_long";
    } else {
      // short texts: G lanes assemble a text in LDS (k_subs_wave); what does not fit its tiles is left to
      // k_subs_emit, which returns at once when nothing was left
      HIP_TRY(hipMemsetAsync(d_left, 0, sizeof(int32_t), s));
#define MRX_SUBS_WAVE(GG)                                                                                        \
  do {                                                                                                           \
    const int64_t blocks = (n + (kBlock / GG) - 1) / (kBlock / GG);                                              \
    hipLaunchKernelGGL(k_subs_wave<GG>, dim3((unsigned)(blocks < grid_cap() ? blocks : grid_cap())), dim3(kBlock), \
                       (size_t)(2 * R + 16), s, n, lay.data, lay.offsets, d_prefix, d_spans, (long long)count, R,   \
                       d_rmap, out_off, out, d_left, (const int32_t*)d_go, env_knobs().subs_debug);              \
  } while (0)
      if (G == 256) MRX_SUBS_WAVE(256);
      else if (G == 64) MRX_SUBS_WAVE(64);
      else if (G == 32) MRX_SUBS_WAVE(32);
      else if (G == 16) MRX_SUBS_WAVE(16);
#undef MRX_SUBS_WAVE
      HIP_TRY(hipGetLastError());
      const int64_t blocks = (n + (kBlock / kSubsLanes) - 1) / (kBlock / kSubsLanes);
      if (cum_later)
        hipLaunchKernelGGL(k_subs_sizes, dim3((unsigned)(blocks < grid_cap() ? blocks : grid_cap())), dim3(kBlock), 0, s,
                           n, lay.offsets, d_prefix, d_spans, (long long)count, R, (int64_t*)nullptr, d_cum, cap,
                           (const int32_t*)d_left);
      // (behind k_subs_wave: the texts it left over -- none when the go-ahead stopped it; alone: the go-ahead itself)
      hipLaunchKernelGGL(k_subs_emit<kSubsLanes>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kBlock),
                         (size_t)(2 * R + 16), s, n, lay.data, lay.offsets, d_prefix, d_spans, d_cum, (long long)count, R,
                         d_rmap, out_off, out, G, G ? (const int32_t*)d_left : (const int32_t*)d_go);
      g_last_kernel = G ? "k_subs_wave" : "k_subs_emit";
    }
    HIP_TRY(hipGetLastError());   // the output bytes are complete when `s` reaches this point, as with every _dev call
    if (group_reach > 0) HIP_TRY(hipMemcpyAsync(&over, d_over, sizeof over, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&nm, d_prefix + n, sizeof nm, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&tot, d_total, sizeof tot, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    h->sub_matches_per_kib.store(in_bytes >= 1024 ? nm / (in_bytes >> 10) : 0, std::memory_order_relaxed);
    if (nm <= cap) break;
    if (attempt == 1) return fail(MRX_E_NO_DEVICE, "sub: match count changed between two passes");
    HIP_TRY(scratch_free(d_spans, s));
    HIP_TRY(scratch_free(d_cum, s));
    cap = nm;   // more matches than one per eight bytes: once more with room for all of them
  }
  if (over) {
    rc = kSubsRetryGeneric;
  } else {
    if (total_bytes) *total_bytes = tot;
    if (tot > out_cap) rc = fail(MRX_E_CAPACITY, "output buffer too small: need " + std::to_string(tot));
  }
  HIP_TRY(scratch_free(d_prefix, s));
  HIP_TRY(scratch_free(d_spans, s));
  HIP_TRY(scratch_free(d_rmap, s));
  HIP_TRY(scratch_free(d_cum, s));
  HIP_TRY(scratch_free(d_sizes, s));
  HIP_TRY(scratch_free(d_total, s));
  HIP_TRY(scratch_free(d_left, s));
  HIP_TRY(scratch_free(d_go, s));
  HIP_TRY(scratch_free(d_over, s));
  return rc;
}
}  // namespace

static int g_chain_sub_general = 0;   // mrx_debug_chain_sub_general
// ---- sub() with \1..\9 on a deterministic chain (HostPlan::chain) -----------------------------------------
// The matches are the plain search's spans (run_findall, match_next_sequence); a wavefront takes a text: every byte's
// leaf mask (bit l: leaf l takes the byte) in an LDS tile, a lane per match finds the leaves' boundaries as runs of
// mask bits, four bytes a step; from them the groups and the replacement's length.  k_subc_sizes: output length per
// text and, per match, the bytes the output has gained in front of it; k_subc_emit<TILE>: gaps and replacements into
// an output tile, the tile out in 16-byte stores.  A text or an output beyond 4 KiB: the call is the lane-per-text
// interpreter's (k_sub).
namespace {
constexpr int kSubcLeaves = 8, kSubcTpl = 8, kSubcRepl = 256;   // (loops over leaves and segments are unrolled: the kernel
                                                                   // arguments they index stay in scalar registers)
struct ChainDev {
  int nleaf, ntpl;
  int lmax[kSubcLeaves];
  int lfix[kSubcLeaves];   // width of a leaf with a fixed count, else 0
  int need;                // bit l: leaf l's runs are looked at (variable count, not the last leaf): its bitmap is built
  ReplSeg tpl[kSubcTpl];   // (as sub_chain_from_spans rewrites them: a group's segment names its two boundaries)
};
// One text's descriptor, loaded two texts ahead; its blocks, first spans and first gains one text ahead: the global
// round trips of text i + 1 run under the LDS phases of text i (as in k_subs_wave).
struct SubcDesc { int64_t ibase, a, obase; int tlen, k, olen; };
template <bool EMIT>
__device__ __forceinline__ SubcDesc subc_desc(int64_t i, int64_t n, const int64_t* offsets, const int64_t* prefix,
                                              const int64_t* out_off, long long count) {
  SubcDesc d{0, 0, 0, 0, 0, 0};
  if (i < n) {
    d.ibase = offsets[i];
    d.tlen = (int)(offsets[i + 1] - d.ibase);
    d.a = prefix[i];
    int64_t k64 = prefix[i + 1] - d.a;
    if (count > 0 && k64 > count) k64 = count;
    d.k = (int)k64;
    if constexpr (EMIT) {
      d.obase = out_off[i];
      d.olen = (int)(out_off[i + 1] - d.obase);
    }
  }
  return d;
}
// blocks in registers -> per-leaf bitmaps (bm[l * ROW + b] bit j: leaf l takes byte 16 b + j of the frame) and, when wanted,
// the text tile; frame of the blocks: the text begins at [address & 15]
template <int NB, int ROW, bool TEXT>
__device__ __forceinline__ void subc_store(const uint4 (&tx)[NB], uint16_t* bm, uint8_t* tile, const uint8_t* mask, int need,
                                           int nfb, int lane) {
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const int b = lane + 64 * r;
    if (b < nfb) {
      const uint4 v = tx[r];
      if constexpr (TEXT) *(uint4*)(tile + 16 * b) = v;
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        o[q] = (uint32_t)mask[w[q] & 255] | ((uint32_t)mask[(w[q] >> 8) & 255] << 8) |
               ((uint32_t)mask[(w[q] >> 16) & 255] << 16) | ((uint32_t)mask[w[q] >> 24] << 24);
#pragma unroll
      for (int l = 0; l < kSubcLeaves; ++l) {
        if (!((need >> l) & 1)) continue;
        uint32_t bits = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)   // bit l of four bytes -> four adjacent bits (the products' other terms stay below bit 24)
          bits |= (((((o[q] >> l) & 0x01010101u) * 0x01020408u) >> 24) & 15u) << (4 * q);
        bm[l * ROW + b] = (uint16_t)bits;
      }
    }
  }
}
// boundaries of the leaves of the match [ms, me) -> bnd[0 .. nleaf] (the lane's row); positions relative to the text, which
// begins at frame position mis; a leaf's run: trailing ones of its bitmap from the position on, 32 positions a step
template <int ROW>
__device__ __forceinline__ void subc_walk(const ChainDev& cd, const uint16_t* bm, int mis, uint16_t* bnd, int ms, int me) {
  int pos = ms;
#pragma unroll
  for (int l = 0; l < kSubcLeaves; ++l) {
    if (l >= cd.nleaf) break;
    bnd[l] = (uint16_t)pos;
    if (l == cd.nleaf - 1) { pos = me; break; }          // the last leaf ends with the match
    if (cd.lfix[l]) { pos += cd.lfix[l]; continue; }     // a fixed count: nothing to look at (the match is the table walk's)
    const int lm = cd.lmax[l];
    const int stop = lm < 0 || pos + lm > me ? me : pos + lm;
    const uint16_t* row = bm + l * ROW;
    while (pos < stop) {
      const int f = mis + pos;
      const uint16_t* w = row + (f >> 4);
      const uint32_t lo = (uint32_t)w[0] | ((uint32_t)w[1] << 16), hi = w[2];
      const uint32_t miss = ~__funnelshift_r(lo, hi, f & 15);
      int k = miss ? __ffs((int)miss) - 1 : 32;
      if (k > stop - pos) k = stop - pos;
      pos += k;
      if (k < 32) break;
    }
  }
  bnd[cd.nleaf] = (uint16_t)pos;
}
// (ChainDev::tpl as the host rewrote it: group_ref != 0: bytes between the boundaries `start` and `length`; else literal)
__device__ __forceinline__ int subc_repl_len(const ChainDev& cd, const uint16_t* bnd) {
  int rl = 0;
#pragma unroll
  for (int k = 0; k < kSubcTpl; ++k) {
    if (k >= cd.ntpl) break;
    const ReplSeg sg = cd.tpl[k];
    rl += sg.group_ref ? (int)bnd[sg.length] - (int)bnd[sg.start] : sg.length;
  }
  return rl;
}
// n bytes src -> dst (LDS, one lane), four reads in flight
__device__ __forceinline__ void subc_copy(uint8_t* dst, const uint8_t* src, int n) {
  for (int j = 0; j < n; j += 4) {
    const uint8_t b0 = src[j], b1 = src[j + 1], b2 = src[j + 2], b3 = src[j + 3];   // (the tiles are padded)
    dst[j] = b0;
    if (j + 1 < n) dst[j + 1] = b1;
    if (j + 2 < n) dst[j + 2] = b2;
    if (j + 3 < n) dst[j + 3] = b3;
  }
}
#define MRX_SUBC_WAVE_SYNC()                                      \
  do {                                                            \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        \
    __builtin_amdgcn_wave_barrier();                              \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        \
  } while (0)
template <int TILE>
__global__ __launch_bounds__(kBlock) void k_subc_sizes(ChainDev cd, int64_t n, const uint8_t* __restrict__ data,
                                                       const int64_t* __restrict__ offsets,
                                                       const int64_t* __restrict__ prefix, const int32_t* __restrict__ spans,
                                                       long long count, const uint8_t* __restrict__ g_mask,
                                                       int64_t* __restrict__ sizes, int32_t* __restrict__ dcum,
                                                       int32_t* __restrict__ longest, int64_t span_cap, int dbg) {
  if (prefix[n] > span_cap) return;   // the findall in front did not have room for its spans: the host retries
  constexpr int NB = TILE / 1024 + 1;
  constexpr int ROW = TILE / 16 + 4;
  __shared__ uint16_t bm_all[kBlock / 64][kSubcLeaves * ROW];
  __shared__ uint16_t bnd_all[kBlock / 64][64 * (kSubcLeaves + 1)];
  __shared__ uint8_t mask[256];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) mask[c] = g_mask[c];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint16_t* bm = bm_all[wv];
  uint16_t* bnd = bnd_all[wv] + lane * (kSubcLeaves + 1);
  int worst = 0;
  const int64_t nw = (int64_t)gridDim.x * (kBlock / 64);
  uint4 tx[NB];
  int2 sp_first;
  auto issue = [&](const SubcDesc& d) {
    const uint8_t* tptr = data + d.ibase;
    const int mis = (int)((uintptr_t)tptr & 15);
    const uint8_t* fptr = tptr - mis;
    const int nfb = d.k > 0 && d.tlen <= TILE ? (mis + d.tlen + 15) >> 4 : 0;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const int b = lane + 64 * r;
      tx[r] = b < nfb ? *(const uint4*)(fptr + 16 * b) : make_uint4(0, 0, 0, 0);
    }
    sp_first = lane < d.k ? *(const int2*)(spans + 2 * (d.a + lane)) : make_int2(0, 0);
  };
  int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + wv;
  SubcDesc d0 = subc_desc<false>(i, n, offsets, prefix, nullptr, count), d1 = subc_desc<false>(i + nw, n, offsets, prefix, nullptr, count);
  issue(d0);
  for (; i < n; i += nw) {
    const SubcDesc d = d0;
    d0 = d1;
    d1 = subc_desc<false>(i + 2 * nw, n, offsets, prefix, nullptr, count);
    const int tlen = d.tlen, k = d.k;
    const int64_t a = d.a;
    if (tlen > TILE) {   // (the host has looked at the longest text already: not reached)
      if (lane == 0) sizes[i] = 0;
      worst = 0x7FFFFFFF;
      issue(d0);
      continue;
    }
    const int mis = (int)((uintptr_t)(data + d.ibase) & 15);
    MRX_SUBC_WAVE_SYNC();
    if (k > 0 && !(dbg & 2)) subc_store<NB, ROW, false>(tx, bm, nullptr, mask, cd.need, (mis + tlen + 15) >> 4, lane);
    const int2 sp_cur = sp_first;
    issue(d0);
    MRX_SUBC_WAVE_SYNC();
    int carry = 0;
    for (int m0 = 0; m0 < k; m0 += 64) {
      const int m = m0 + lane;
      int delta = 0;
      if (m < k) {
        const int2 sp = m0 == 0 ? sp_cur : *(const int2*)(spans + 2 * (a + m));
        if (!(dbg & 1)) {
          subc_walk<ROW>(cd, bm, mis, bnd, sp.x, sp.y);
          delta = subc_repl_len(cd, bnd) - (sp.y - sp.x);
        }
      }
      const int incl = group_scan<64, false>(delta);
      if (m < k && !(dbg & 4)) dcum[a + m] = carry + incl - delta;
      carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) sizes[i] = (int64_t)tlen + carry;
    if (tlen + carry > worst) worst = tlen + carry;
  }
  if (lane == 0 && worst > 0) atomicMax(longest, worst);
}
// Every match's replacement differs from the match by the same number of bytes (each leaf of variable width lies in
// exactly one referenced group -- a template that reorders the groups): no walk for the sizes, a lane per text.
__global__ __launch_bounds__(kBlock) void k_subc_sizes_const(int64_t n, const int64_t* __restrict__ offsets,
                                                             const int64_t* __restrict__ prefix, long long count, int delta,
                                                             int64_t* __restrict__ sizes, int32_t* __restrict__ longest,
                                                             int64_t span_cap) {
  if (prefix[n] > span_cap) return;
  int worst = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tlen = offsets[i + 1] - offsets[i];
    int64_t k = prefix[i + 1] - prefix[i];
    if (count > 0 && k > count) k = count;
    const int64_t olen = tlen + k * delta;
    sizes[i] = olen;
    const int w = tlen > 0x7FFFFFFF || olen > 0x7FFFFFFF ? 0x7FFFFFFF : (int)(tlen > olen ? tlen : olen);
    if (w > worst) worst = w;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(worst, d, 64); if (o > worst) worst = o; }
  if ((threadIdx.x & 63) == 0 && worst > 0) atomicMax(longest, worst);
}
template <int TILE>
__global__ __launch_bounds__(kBlock) void k_subc_emit(ChainDev cd, int64_t n, const uint8_t* __restrict__ data,
                                                      const int64_t* __restrict__ offsets,
                                                      const int64_t* __restrict__ prefix, const int32_t* __restrict__ spans,
                                                      long long count, const uint8_t* __restrict__ g_mask,
                                                      const uint8_t* __restrict__ g_repl, int repl_len,
                                                      const int32_t* __restrict__ dcum, const int64_t* __restrict__ out_off,
                                                      uint8_t* __restrict__ out, int cdelta, int dbg) {
  // dbg (MRX_SUBC_DEBUG, timing only): 8 no gap copies, 32 no replacement bytes, 64 no stores
  // dcum == nullptr: every match gains cdelta bytes (k_subc_sizes_const)
  constexpr int NB = TILE / 1024 + 1;
  constexpr int ROW = TILE / 16 + 4;
  __shared__ uint16_t bm_all[kBlock / 64][kSubcLeaves * ROW];
  __shared__ __align__(16) uint8_t text_all[kBlock / 64][TILE + 32];
  __shared__ __align__(16) uint8_t otile_all[kBlock / 64][TILE + 32];
  __shared__ uint16_t bnd_all[kBlock / 64][64 * (kSubcLeaves + 1)];
  __shared__ uint8_t mask[256];
  __shared__ uint8_t repl[kSubcRepl + 8];
  for (int c = threadIdx.x; c < 256; c += blockDim.x) mask[c] = g_mask[c];
  for (int c = threadIdx.x; c < repl_len; c += blockDim.x) repl[c] = g_repl[c];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint16_t* bm = bm_all[wv];
  uint8_t* tile = text_all[wv];
  uint8_t* otile = otile_all[wv];
  uint16_t* bnd = bnd_all[wv] + lane * (kSubcLeaves + 1);
  const int64_t nw = (int64_t)gridDim.x * (kBlock / 64);
  uint4 tx[NB];
  int2 sp_first;
  int dc_first;
  auto takes = [&](const SubcDesc& d) { return d.olen > 0 && d.olen <= TILE && d.tlen <= TILE; };   // (else: not launched)
  auto issue = [&](const SubcDesc& d) {
    const uint8_t* tptr = data + d.ibase;
    const int mis = (int)((uintptr_t)tptr & 15);
    const uint8_t* fptr = tptr - mis;
    const int nfb = takes(d) ? (mis + d.tlen + 15) >> 4 : 0;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const int b = lane + 64 * r;
      tx[r] = b < nfb ? *(const uint4*)(fptr + 16 * b) : make_uint4(0, 0, 0, 0);
    }
    const bool have = takes(d) && lane < d.k;
    sp_first = have ? *(const int2*)(spans + 2 * (d.a + lane)) : make_int2(0, 0);
    dc_first = have && dcum ? dcum[d.a + lane] : lane * cdelta;
  };
  int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + wv;
  SubcDesc d0 = subc_desc<true>(i, n, offsets, prefix, out_off, count), d1 = subc_desc<true>(i + nw, n, offsets, prefix, out_off, count);
  issue(d0);
  for (; i < n; i += nw) {
    const SubcDesc d = d0;
    d0 = d1;
    d1 = subc_desc<true>(i + 2 * nw, n, offsets, prefix, out_off, count);
    const int tlen = d.tlen, k = d.k, olen = d.olen;
    const int64_t a = d.a, obase = d.obase;
    if (!takes(d)) { issue(d0); continue; }
    const int mis = (int)((uintptr_t)(data + d.ibase) & 15), head = (int)((uintptr_t)(out + obase) & 15);
    MRX_SUBC_WAVE_SYNC();   // the previous text's tiles are free
    subc_store<NB, ROW, true>(tx, bm, tile, mask, cd.need, (mis + tlen + 15) >> 4, lane);
    const int2 sp_cur = sp_first;
    const int dc_cur = dc_first;
    issue(d0);
    MRX_SUBC_WAVE_SYNC();
    const uint8_t* txt = tile + mis;
    uint8_t* o = otile + head;
    // the whole wavefront copies text[src, src + len) to o[dst ..]
    auto copy_all = [&](int src, int len, int dst) {
      for (int j = lane; j < len; j += 64) o[dst + j] = txt[src + j];
    };
    int prev_end = 0;   // end of the match in front of this round's first
    for (int m0 = 0; m0 < k; m0 += 64) {
      const int m = m0 + lane;
      const bool live = m < k;
      int ms = 0, me = 0, before = 0;
      if (live) {
        const int2 sp = m0 == 0 ? sp_cur : *(const int2*)(spans + 2 * (a + m));
        ms = sp.x; me = sp.y;
        before = m0 == 0 ? dc_cur : dcum ? dcum[a + m] : m * cdelta;
      }
      int pe = __shfl_up(me, 1, 64);
      if (lane == 0) pe = prev_end;
      const int nlive = k - m0 < 64 ? k - m0 : 64;
      prev_end = __shfl(me, nlive - 1, 64);
      // the kept bytes in front of the match: short gaps by the lane, long ones by the wavefront
      const int gap = live ? ms - pe : 0;
      if (gap <= 24 && !(dbg & 8)) subc_copy(o + pe + before, txt + pe, gap);
      uint64_t wide = __ballot(gap > 24);
      while (wide) {
        const int src_lane = __ffsll((unsigned long long)wide) - 1;
        wide &= wide - 1;
        copy_all(__shfl(pe, src_lane, 64), __shfl(gap, src_lane, 64), __shfl(pe + before, src_lane, 64));
      }
      if (live) {
        subc_walk<ROW>(cd, bm, mis, bnd, ms, me);
        uint8_t* dst = o + ms + before;
        if (!(dbg & 32))
#pragma unroll
        for (int q = 0; q < kSubcTpl; ++q) {
          if (q >= cd.ntpl) break;
          const ReplSeg sg = cd.tpl[q];
          if (sg.group_ref) {
            const int gs = bnd[sg.start], gl = (int)bnd[sg.length] - gs;
            subc_copy(dst, txt + gs, gl);
            dst += gl;
          } else {
            subc_copy(dst, repl + sg.start, sg.length);
            dst += sg.length;
          }
        }
      }
    }
    copy_all(prev_end, tlen - prev_end, prev_end + (olen - tlen));   // behind the last replaced match
    MRX_SUBC_WAVE_SYNC();
    // the tile out: whole 16-byte blocks where the block is the text's alone, bytes at the two ends
    uint8_t* oframe = out + obase - head;
    const int nob = (head + olen + 15) >> 4;
    for (int b = lane; b < ((dbg & 64) ? 0 : nob); b += 64) {
      const int lo = 16 * b, hi = lo + 16;
      if (lo >= head && hi <= head + olen) {
        *(uint4*)(oframe + lo) = *(const uint4*)(otile + lo);
      } else {
        const int from = lo > head ? lo : head, to = hi < head + olen ? hi : head + olen;
        for (int j = from; j < to; ++j) oframe[j] = otile[j];
      }
    }
  }
}
#undef MRX_SUBC_WAVE_SYNC

int sub_chain_from_spans(const mrx_handle* h, const Layout& lay, int64_t n, const std::string& r,
                         const std::vector<ReplSeg>& tpl, int64_t count, int64_t* out_off, uint8_t* out, int64_t out_cap,
                         int64_t* total_bytes, hipStream_t s, int64_t known_bytes, int64_t known_max) {
  const ChainGroups& cg = h->hp.chain;
  if (cg.nleaf > kSubcLeaves || (int)tpl.size() > kSubcTpl || (int)r.size() > kSubcRepl) return kSubsRetryGeneric;
  int64_t in_bytes = known_bytes, max_len = known_max;
  if (known_bytes < 0 || known_max < 0)
    if (int rc0 = csr_stats(lay, n, s, &in_bytes, &max_len)) return rc0;
  if (in_bytes < 0) return fail(MRX_E_ARGUMENT, "offsets[n] is negative");
  if (max_len > 4096) return kSubsRetryGeneric;
  ChainDev cd{};
  cd.nleaf = cg.nleaf; cd.ntpl = (int)tpl.size();
  for (int l = 0; l < kSubcLeaves; ++l) { cd.lmax[l] = cg.lmax[l]; cd.lfix[l] = l < cg.nleaf && cg.lmin[l] == cg.lmax[l] ? cg.lmin[l] : 0; }
  for (int l = 0; l + 1 < cg.nleaf; ++l) if (!cd.lfix[l]) cd.need |= 1 << l;
  cd.ntpl = 0;
  for (const ReplSeg& sg : tpl) {   // a group's segment: the two boundaries it lies between (a group the pattern lacks: nothing)
    if (sg.group_ref > 0) {
      if (sg.group_ref <= 9 && cg.gopen[sg.group_ref] >= 0) cd.tpl[cd.ntpl++] = ReplSeg{1, cg.gopen[sg.group_ref], cg.gclose[sg.group_ref]};
    } else if (sg.length > 0) {
      cd.tpl[cd.ntpl++] = sg;
    }
  }
  uint8_t mask8[256];
  for (int c = 0; c < 256; ++c) mask8[c] = (uint8_t)cg.mask[c];
  // does every match gain the same number of bytes?  (literal bytes + fixed-width leaves counted by the groups that hold
  // them - their own width; a leaf of variable width must lie in exactly one referenced group)
  bool const_delta = true;
  int cdelta = 0;
  {
    int covers[kSubcLeaves] = {0};
    for (int k = 0; k < cd.ntpl; ++k) {
      if (cd.tpl[k].group_ref) for (int l = cd.tpl[k].start; l < cd.tpl[k].length; ++l) ++covers[l];
      else cdelta += cd.tpl[k].length;
    }
    for (int l = 0; l < cg.nleaf; ++l) {
      if (cg.lmin[l] == cg.lmax[l]) cdelta += (covers[l] - 1) * cg.lmin[l];
      else if (covers[l] != 1) const_delta = false;
    }
    if (g_chain_sub_general) const_delta = false;   // (mrx_debug_chain_sub_general: the general form, for A/B runs and tests)
  }
  int64_t *d_prefix = nullptr, *d_sizes = nullptr, *d_total = nullptr;
  int32_t *d_spans = nullptr, *d_dcum = nullptr, *d_longest = nullptr;
  uint8_t *d_mask = nullptr, *d_repl = nullptr;
  HIP_TRY(scratch_alloc((void**)&d_prefix, sizeof(int64_t) * (n + 1), s));
  HIP_TRY(scratch_alloc((void**)&d_sizes, sizeof(int64_t) * n, s));
  // (output total, longest output, match total: three words side by side, one copy to the host)
  HIP_TRY(scratch_alloc((void**)&d_total, 3 * sizeof(int64_t), s));
  d_longest = (int32_t*)(d_total + 1);
  HIP_TRY(scratch_alloc((void**)&d_mask, 256, s));
  HIP_TRY(scratch_alloc((void**)&d_repl, r.size() + 16, s));
  HIP_TRY(hipMemcpyAsync(d_mask, mask8, 256, hipMemcpyHostToDevice, s));   // (pageable source: copied before the call returns)
  if (!r.empty()) HIP_TRY(hipMemcpyAsync(d_repl, r.data(), r.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_total, 0, 3 * sizeof(int64_t), s));
  int64_t cap = in_bytes / 8 + n + 64, nm = 0, tot = 0;
  if (const int64_t seen = h->sub_matches_per_kib.load(std::memory_order_relaxed); seen > 128) {
    const int64_t by_hint = (in_bytes >> 10) * (seen + seen / 8 + 1) + n + 64;
    if (by_hint > cap) cap = by_hint < in_bytes + n + 64 ? by_hint : in_bytes + n + 64;
  }
  int32_t longest = 0;
  const int subc_dbg = env_knobs().subc_debug;   // (ablation runs: wrong results)
  const int64_t blocks = (n + (kBlock / 64) - 1) / (kBlock / 64);
  const unsigned grid = (unsigned)(blocks < grid_cap() ? blocks : grid_cap());
  int rc = MRX_OK;
  // (one host synchronisation: the match total, which sizes the spans; the output total and the longest output, which
  // picks the emit kernel's tile)
  for (int attempt = 0; attempt < 2; ++attempt) {
    HIP_TRY(scratch_alloc((void**)&d_spans, sizeof(int32_t) * 2 * (size_t)cap, s));
    HIP_TRY(scratch_alloc((void**)&d_dcum, sizeof(int32_t) * (size_t)(cap + 1), s));
    rc = run_findall(h, lay, n, d_prefix, d_spans, cap, nullptr, s, /*match_next_sequence=*/true, in_bytes, max_len);
    if (rc != MRX_OK) return rc;
    // (sizes behind the spans without waiting: when the spans did not fit it returns at once)
    if (const_delta)
      hipLaunchKernelGGL(k_subc_sizes_const, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, lay.offsets, d_prefix,
                         (long long)count, cdelta, d_sizes, d_longest, cap);
    else if (max_len <= 2048)
      hipLaunchKernelGGL(k_subc_sizes<2048>, dim3(grid), dim3(kBlock), 0, s, cd, n, lay.data, lay.offsets, d_prefix, d_spans,
                         (long long)count, d_mask, d_sizes, d_dcum, d_longest, cap, subc_dbg);
    else
      hipLaunchKernelGGL(k_subc_sizes<4096>, dim3(grid), dim3(kBlock), 0, s, cd, n, lay.data, lay.offsets, d_prefix, d_spans,
                         (long long)count, d_mask, d_sizes, d_dcum, d_longest, cap, subc_dbg);
    HIP_TRY(hipGetLastError());
    rc = device_scan<int64_t>(d_sizes, n, out_off, d_total, s);
    if (rc != MRX_OK) return rc;
    int64_t h3[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(d_total + 2, d_prefix + n, sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(h3, d_total, sizeof h3, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    tot = h3[0]; longest = (int32_t)(h3[1] & 0xFFFFFFFF); nm = h3[2];
    h->sub_matches_per_kib.store(in_bytes >= 1024 ? nm / (in_bytes >> 10) : 0, std::memory_order_relaxed);
    if (nm <= cap) break;
    if (attempt == 1) return fail(MRX_E_NO_DEVICE, "sub: match count changed between two passes");
    HIP_TRY(scratch_free(d_spans, s));
    HIP_TRY(scratch_free(d_dcum, s));
    cap = nm;
  }
  if (longest > 4096) {
    rc = kSubsRetryGeneric;
  } else {
    if (total_bytes) *total_bytes = tot;
    if (tot > out_cap) {
      rc = fail(MRX_E_CAPACITY, "output buffer too small: need " + std::to_string(tot));
    } else if (tot > 0) {
#define MRX_SUBC_EMIT(TT)                                                                                              \
  hipLaunchKernelGGL(k_subc_emit<TT>, dim3(grid), dim3(kBlock), 0, s, cd, n, lay.data, lay.offsets, d_prefix, d_spans, \
                     (long long)count, d_mask, d_repl, (int)r.size(), const_delta ? (const int32_t*)nullptr : d_dcum, out_off, out, cdelta, subc_dbg)
      if (longest <= 2048 && max_len <= 2048) MRX_SUBC_EMIT(2048);
      else MRX_SUBC_EMIT(4096);
#undef MRX_SUBC_EMIT
      HIP_TRY(hipGetLastError());
      g_last_kernel = "k_subc_emit";
    }
  }
  HIP_TRY(scratch_free(d_prefix, s));
  HIP_TRY(scratch_free(d_sizes, s));
  HIP_TRY(scratch_free(d_total, s));
  HIP_TRY(scratch_free(d_mask, s));
  HIP_TRY(scratch_free(d_repl, s));
  HIP_TRY(scratch_free(d_spans, s));
  HIP_TRY(scratch_free(d_dcum, s));
  return rc;
}
}  // namespace

namespace {
struct DevBatch {
  uint8_t* data = nullptr;
  int64_t* offsets = nullptr;
  int64_t nbytes = 0;
  ~DevBatch() { if (data) (void)hipFree(data); if (offsets) (void)hipFree(offsets); }
  int upload(const uint8_t* h_data, const int64_t* h_off, int64_t n) {
    if (n < 0 || !h_off) return fail(MRX_E_ARGUMENT, "bad batch");
    nbytes = h_off[n];
    HIP_TRY(hipMalloc((void**)&data, (size_t)nbytes + 64));
    HIP_TRY(hipMalloc((void**)&offsets, sizeof(int64_t) * (n + 1)));
    if (nbytes) HIP_TRY(hipMemcpy(data, h_data + h_off[0], (size_t)(nbytes - h_off[0]), hipMemcpyHostToDevice));
    std::vector<int64_t> rel(n + 1);
    for (int64_t i = 0; i <= n; ++i) rel[i] = h_off[i] - h_off[0];
    HIP_TRY(hipMemcpy(offsets, rel.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice));
    return MRX_OK;
  }
};
template <class T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t count) { HIP_TRY(hipMalloc((void**)&p, sizeof(T) * (count ? count : 1))); return MRX_OK; }
};
}  // namespace

extern "C" {

int mrx_compile(const char* pattern, size_t pattern_len, mrx_handle** out) {
  return mrx_compile_ex(pattern, pattern_len, 0u, out);
}

int mrx_compile_ex(const char* pattern, size_t pattern_len, uint32_t options, mrx_handle** out) {
  if (const char* e = getenv("MRX_NO_PAIR_TABLES")) g_pair_tables = !(e[0] == '1');
  {   // MRX_FUSED is read once per process, so that it does not override a later mrx_debug_fused_findall()
    static const bool once = [] {
      if (const char* e = getenv("MRX_FUSED")) g_fused = atoi(e) < 0 ? 0 : atoi(e) > 2 ? 2 : atoi(e);
      return true;
    }();
    (void)once;
  }
  if (const char* e = getenv("MRX_FUSED_BPC")) g_fused_bpc = atoi(e);
  if (!out || (!pattern && pattern_len)) return fail(MRX_E_ARGUMENT, "null argument");
  if (options & ~(uint32_t)(MRX_COMPILE_LAZYDFA_SEMANTICS | MRX_COMPILE_BITSET_NFA | MRX_COMPILE_NFA_ENGINE |
                             MRX_COMPILE_DFA_ENGINE))
    return fail(MRX_E_ARGUMENT, "unknown compile option");
  if ((options & MRX_COMPILE_NFA_ENGINE) && (options & (MRX_COMPILE_DFA_ENGINE | MRX_COMPILE_LAZYDFA_SEMANTICS)))
    return fail(MRX_E_ARGUMENT, "MRX_COMPILE_NFA_ENGINE excludes the other engine options");
  if ((options & MRX_COMPILE_DFA_ENGINE) && (options & MRX_COMPILE_LAZYDFA_SEMANTICS))
    return fail(MRX_E_ARGUMENT, "MRX_COMPILE_DFA_ENGINE excludes MRX_COMPILE_LAZYDFA_SEMANTICS");
  *out = nullptr;
  mrx_handle* h = new mrx_handle();
  try {
    build_plan(std::string(pattern, pattern_len), h->hp, (options & MRX_COMPILE_LAZYDFA_SEMANTICS) != 0,
               (options & MRX_COMPILE_BITSET_NFA) != 0, (options & MRX_COMPILE_NFA_ENGINE) != 0,
               (options & MRX_COMPILE_DFA_ENGINE) != 0);
  } catch (const SyntaxError& e) {
    delete h;
    return fail(MRX_E_SYNTAX, e.what());
  } catch (const std::exception& e) {
    delete h;
    return fail(MRX_E_UNSUPPORTED, e.what());
  }
  *out = h;
  return MRX_OK;
}

void mrx_free(mrx_handle* h) {
  if (!h) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (int d = 0; d < kMaxDevices; ++d)
    if (uint8_t* p = h->d_blobs[d].load()) { (void)hipSetDevice(d); (void)hipFree(p); }
  for (auto& kv : h->req_tune) {   // the route tuner's events
    (void)hipSetDevice(kv.second.dev);
    for (auto& pr : kv.second.ev)
      for (auto& e : pr)
        if (e) (void)hipEventDestroy(e);
  }
  if (h->dense_probe.ev || h->dense_probe.host_total) {
    (void)hipSetDevice(h->dense_probe.dev);
    if (h->dense_probe.ev) { (void)hipEventSynchronize(h->dense_probe.ev); (void)hipEventDestroy(h->dense_probe.ev); }
    if (h->dense_probe.host_total) (void)hipHostFree(h->dense_probe.host_total);
  }
  (void)hipSetDevice(cur);
  delete h;
}

const char* mrx_last_error(void) { return g_err.c_str(); }
const char* mrx_engine_type(const mrx_handle* h) { return h ? h->hp.engine_type.c_str() : ""; }
const char* mrx_stats(const mrx_handle* h) { return h ? h->hp.stats.c_str() : ""; }
int mrx_num_groups(const mrx_handle* h) {
  if (!h) return 0;
  if (h->hp.fixed_total >= 0) return h->hp.fixed_ngroups;
  return h->hp.bt.ok ? h->hp.bt.ngroups : 0;
}
const char* mrx_version(void) { return "mrx-hip 0.1 (gfx950)"; }

size_t mrx_describe(const mrx_handle* h, char* buf, size_t cap) {
  if (!h) return 0;
  const std::string s = describe_plan(h->hp);
  if (buf && cap) {
    const size_t k = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(buf, s.data(), k);
    buf[k] = 0;
  }
  return s.size();
}

// regex.search / regex.match_first for any layout: the streaming kernel when the plan and the
// layout allow it, the generic lane-per-text kernel otherwise
static int run_search_any(const mrx_handle* h, const Layout& lay, int64_t n, int32_t* ds, int32_t* de,
                          void* st) {
  ScratchScope scratch_scope_((hipStream_t)st);
  if (!h) return fail(MRX_E_ARGUMENT, "null handle");
  const DevPlan& p = h->hp.dev;
  // '^'-anchored DFA plans: match_next only ever tries position 0 (dfa.mojo:1875-1886), so search is
  // the anchored automaton's run (the pure-literal case differs: simd_search is not anchored)
  const bool anchored0 = anchored_at_zero(h) && stream_layout_ok(lay, n);
  if (anchored0) {
    if (int rc = ensure_device(h)) return rc;
    hipStream_t s = (hipStream_t)st;
    ScanTimer tm(s);
    launch_stream<ST_FIRST>(h, lay, n, nullptr, nullptr, nullptr, 0, ds, de, s, lay.vlen, lay.vskip);
    g_last_kernel = "k_stream_first";
    HIP_TRY(hipGetLastError());
    tm.stop();
    return MRX_OK;
  }
  // Memchr prefilter (matcher.mojo:784-796): match_next = "find the literal from start on, then let the engine
  // search from THERE" -- first occurrence per text by shift-and (k_litscan), then the engine's search on the
  // view of each text from that occurrence on (streaming kernel or stepper), offsets added back
  if (!g_force_generic && (p.flags & PF_PREFILTER) && !t_prefilter_done && p.pre_len >= 1 && p.pre_len <= 32 && n > 0 &&
      h->hp.why_no_search.empty() && (p.flags & (PF_STREAMABLE | PF_STEP_SEARCH))) {
    if (int rc = ensure_device(h)) return rc;
    hipStream_t s = (hipStream_t)st;
    int32_t* d_cand = nullptr;
    int64_t* vstart = nullptr;
    int32_t* vlen = nullptr;
    uint32_t* vskip = nullptr;
    HIP_TRY(scratch_alloc((void**)&d_cand, sizeof(int32_t) * n, s));
    HIP_TRY(scratch_alloc((void**)&vstart, sizeof(int64_t) * (n + 1), s));
    HIP_TRY(scratch_alloc((void**)&vlen, sizeof(int32_t) * n, s));
    HIP_TRY(scratch_alloc((void**)&vskip, sizeof(uint32_t) * n, s));
    {
      const int64_t nw = (n + 63) / 64;
      int64_t g = (nw + kWsWaves - 1) / kWsWaves;
      if (g > grid_cap()) g = grid_cap();
      hipLaunchKernelGGL(k_litscan<false>, dim3((unsigned)g), dim3(64 * kWsWaves), 0, s, H_BLOB(h) + p.off_pre, p.pre_len, H_BLOB(h), lay, n,
                         d_cand, (int2*)nullptr);
    }
    hipLaunchKernelGGL(k_view_build, dim3(grid_for(n + 1, kBlock)), dim3(kBlock), 0, s, lay, n, 0, d_cand, vstart, vlen, vskip, 0);
    HIP_TRY(hipGetLastError());
    Layout view{lay.data, vstart, 0, nullptr, 0};
    view.vlen = vlen;
    view.vskip = vskip;
    t_prefilter_done = true;
    const int rc = run_search_any(h, view, n, ds, de, st);
    t_prefilter_done = false;
    if (rc != MRX_OK) return rc;
    hipLaunchKernelGGL(k_view_fix, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, lay, n, 0, d_cand, 0, ds, de, (uint8_t*)nullptr);
    HIP_TRY(hipGetLastError());
    return MRX_OK;
  }
  if (g_force_generic || !((p.flags & PF_STREAM_SEARCH) || ((p.flags & PF_STREAMABLE) && t_prefilter_done)) ||
      !stream_layout_ok(lay, n))
    return run_match<OP_SEARCH>(h, lay, n, ds, de, nullptr, st);
  if (int rc = check_search_supported(h)) return rc;
  if (int rc = ensure_device(h)) return rc;
  hipStream_t s = (hipStream_t)st;
  Pieces pc;
  if (!lay.vlen)   // (a view is searched as it stands)
    if (int rc = pieces_prepare(h, lay, n, s, &pc)) return rc;
  if (pc.on) {   // long texts: search every piece, keep each text's first
    int32_t* d_vs = nullptr;
    HIP_TRY(scratch_alloc((void**)&d_vs, sizeof(int32_t) * 2 * pc.nv, s));
    ScanTimer tm(s);
    launch_stream<ST_SEARCH>(h, pc.lay, pc.nv, nullptr, nullptr, nullptr, 0, d_vs, d_vs + pc.nv, s, pc.vlen, pc.vskip);
    tm.stop();
    hipLaunchKernelGGL(k_virt_first, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, pc.vfirst, d_vs, d_vs + pc.nv,
                       pc.vbase, ds, de);
    HIP_TRY(hipGetLastError());
    HIP_TRY(scratch_free(d_vs, s));
    g_last_kernel = "k_stream_search_pieces";
    return pieces_release(&pc, s);
  }
  ScanTimer tm(s);
  if (dyn_ok(h, lay, n)) {
    launch_stream_dyn<ST_SEARCH>(h, lay, n, nullptr, nullptr, nullptr, ds, de, s, false);
    g_last_kernel = "k_stream_search_dyn";
    HIP_TRY(hipGetLastError());
    tm.stop();
    return MRX_OK;
  }
  launch_stream<ST_SEARCH>(h, lay, n, nullptr, nullptr, nullptr, 0, ds, de, s, lay.vlen, lay.vskip);
  g_last_kernel = "k_stream_search";
  HIP_TRY(hipGetLastError());
  tm.stop();
  return MRX_OK;
}
static int run_first_any(const mrx_handle* h, const Layout& lay, int64_t n, int32_t* ds, int32_t* de,
                         void* st) {
  ScratchScope scratch_scope_((hipStream_t)st);
  if (!h) return fail(MRX_E_ARGUMENT, "null handle");
  const DevPlan& p = h->hp.dev;
  if (n == 0 && h->hp.why_no_match_first.empty()) return MRX_OK;
  if ((g_force_generic && !h->hp.first_onepass) || p.fa_bytes <= 0 || !stream_layout_ok(lay, n))
    return run_match<OP_MATCH_FIRST>(h, lay, n, ds, de, nullptr, st);
  if (!h->hp.why_no_match_first.empty()) return fail(MRX_E_UNSUPPORTED, h->hp.why_no_match_first);
  if (int rc = ensure_device(h)) return rc;
  hipStream_t s = (hipStream_t)st;
  // a single class run on long texts of a fixed-pitch batch: a wavefront per text (k_first_run)
  if (p.off_fa_run >= 0 && !lay.offsets && !lay.vlen && g_long_text_mode != 2 &&
      (g_long_text_mode == 1 || g_long_text_mode == 3 || ((lay.lens ? lay.stride : (int64_t)lay.len) >= 2048 && n <= 131072))) {
    ScanTimer tm(s);
    hipLaunchKernelGGL(k_first_run, dim3(grid_for(n * 64, kBlock)), dim3(kBlock), 0, s, p, H_BLOB(h), lay, n, ds, de);
    g_last_kernel = "k_first_run";
    HIP_TRY(hipGetLastError());
    tm.stop();
    return MRX_OK;
  }
  ScanTimer tm(s);
  launch_stream<ST_FIRST>(h, lay, n, nullptr, nullptr, nullptr, 0, ds, de, s, lay.vlen, lay.vskip);
  g_last_kernel = "k_stream_first";
  HIP_TRY(hipGetLastError());
  tm.stop();
  return MRX_OK;
}

int mrx_match_first_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n,
                        int32_t* s, int32_t* e, void* st) {
  if (!off) return fail(MRX_E_ARGUMENT, "null offsets");
  return run_first_any(h, Layout{d, off, 0, nullptr, 0}, n, s, e, st);
}
int mrx_search_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n, int32_t* s,
                   int32_t* e, void* st) {
  if (!off) return fail(MRX_E_ARGUMENT, "null offsets");
  return run_search_any(h, Layout{d, off, 0, nullptr, 0}, n, s, e, st);
}
int mrx_search_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride, const int32_t* lens,
                           int32_t len, int64_t n, int32_t* ds, int32_t* de, void* st) {
  if (stride <= 0) return fail(MRX_E_ARGUMENT, "stride must be positive");
  if (!lens && (len < 0 || len > stride)) return fail(MRX_E_ARGUMENT, "len must be in [0, stride]");
  return run_search_any(h, Layout{d, nullptr, stride, lens, len}, n, ds, de, st);
}
int mrx_match_first_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride,
                                const int32_t* lens, int32_t len, int64_t n, int32_t* ds,
                                int32_t* de, void* st) {
  if (stride <= 0) return fail(MRX_E_ARGUMENT, "stride must be positive");
  if (!lens && (len < 0 || len > stride)) return fail(MRX_E_ARGUMENT, "len must be in [0, stride]");
  return run_first_any(h, Layout{d, nullptr, stride, lens, len}, n, ds, de, st);
}
static int run_is_match_any(const mrx_handle* h, const Layout& lay, int64_t n, uint8_t* f, void* st) {
  ScratchScope scratch_scope_((hipStream_t)st);
  if (h && n > 0 && !g_force_generic && h->hp.why_no_match_first.empty() && h->hp.dev.kind == PLAN_DFA &&
      (h->hp.dev.flags & PF_HAS_MATCHER) && h->hp.dev.nstates > 0) {
    // the first-byte quirk (A.6 #8): no walk at all
    if (int rc = ensure_device(h)) return rc;
    hipStream_t s = (hipStream_t)st;
    hipLaunchKernelGGL(k_is_match_byte, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, lay, n, H_BLOB(h) + h->hp.dev.off_first,
                       (h->hp.dev.flags & PF_START_ACCEPTING) ? 1 : 0, f);
    g_last_kernel = "k_is_match_byte";
    HIP_TRY(hipGetLastError());
    return MRX_OK;
  }
  // everything else is "match_first(text, 0) is not None" (dfa.mojo:1845-1849, matcher.mojo:721-731): the
  // anchored automaton on the streaming kernel where the plan has one
  const bool via_first = h && n > 0 && (h->hp.first_onepass ||
                                        (!g_force_generic && h->hp.why_no_match_first.empty() && h->hp.dev.kind != PLAN_ANY &&
                                         h->hp.dev.fa_bytes > 0));
  if (via_first) {
    // NFA-routed: is_match = match_first(text, 0) is not None (matcher.mojo:721-731)
    hipStream_t s = (hipStream_t)st;
    int32_t* tmp = nullptr;
    HIP_TRY(scratch_alloc((void**)&tmp, sizeof(int32_t) * 2 * n, s));
    int rc = run_first_any(h, lay, n, tmp, tmp + n, st);
    if (rc == MRX_OK) {
      hipLaunchKernelGGL(k_span_to_flag, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, tmp, f);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(scratch_free(tmp, s));
    return rc;
  }
  return run_match<OP_IS_MATCH>(h, lay, n, nullptr, nullptr, f, st);
}
int mrx_is_match_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n,
                     uint8_t* f, void* st) {
  if (!off) return fail(MRX_E_ARGUMENT, "null offsets");
  return run_is_match_any(h, Layout{d, off, 0, nullptr, 0}, n, f, st);
}
int mrx_is_match_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride, const int32_t* lens,
                             int32_t len, int64_t n, uint8_t* f, void* st) {
  if (stride <= 0) return fail(MRX_E_ARGUMENT, "stride must be positive");
  if (!lens && (len < 0 || len > stride)) return fail(MRX_E_ARGUMENT, "len must be in [0, stride]");
  return run_is_match_any(h, Layout{d, nullptr, stride, lens, len}, n, f, st);
}
// match_first / match_next / is_match at a start position (see k_view_build): the operation runs on a
// view of the batch, k_view_fix turns the view's answers into the text's.
enum { AT_FIRST = 0, AT_SEARCH = 1, AT_IS_MATCH = 2 };
static int run_at(int op, const mrx_handle* h, const Layout& lay, int64_t n, int32_t start, const int32_t* d_starts,
                  int32_t* ds, int32_t* de, uint8_t* flag, void* st) {
  if (!h) return fail(MRX_E_ARGUMENT, "null handle");
  if (n < 0) return fail(MRX_E_ARGUMENT, "negative batch size");
  hipStream_t s = (hipStream_t)st;
  ScratchScope scratch_scope_(s);
  if (start == 0 && !d_starts) {
    return op == AT_FIRST ? run_first_any(h, lay, n, ds, de, st)
         : op == AT_SEARCH ? run_search_any(h, lay, n, ds, de, st) : run_is_match_any(h, lay, n, flag, st);
  }
  // refusals first (nothing is launched for an operation the plan does not support)
  if (op == AT_SEARCH) { if (int rc = check_search_supported(h)) return rc; }
  else if (!h->hp.why_no_match_first.empty()) return fail(MRX_E_UNSUPPORTED, h->hp.why_no_match_first);
  // (a program without '^', '.*' fast paths and look-back -- a prefix literal or none -- only ever looks at
  // text[start:], like the other engines)
  bool bt_abs = (h->hp.dev.bt_flags & (4 | 8)) != 0 || ((h->hp.dev.bt_flags & 1) && !(h->hp.dev.bt_flags & 2));
  for (const BtItem& it : h->hp.bt.items) bt_abs = bt_abs || it.kind == BT_START;
  if (bt_abs && (h->hp.dev.flags & (op == AT_SEARCH ? PF_BT_SEARCH : PF_BT_FIRST)))
    return fail(MRX_E_UNSUPPORTED,
                "start != 0 on an operation the reference runs on its backtracking matcher: '^' and the literal "
                "prefilter's look-back use absolute text positions (nfa.mojo:998-1006, 466-467), which the view "
                "of the text from `start` on does not have");
  if (n == 0) return MRX_OK;
  if (int rc = ensure_device(h)) return rc;
  int64_t* vstart = nullptr;
  int32_t* vlen = nullptr;
  uint32_t* vskip = nullptr;
  HIP_TRY(scratch_alloc((void**)&vstart, sizeof(int64_t) * (n + 1), s));
  HIP_TRY(scratch_alloc((void**)&vlen, sizeof(int32_t) * n, s));
  HIP_TRY(scratch_alloc((void**)&vskip, sizeof(uint32_t) * n, s));
  // NFAEngine.match_first (is_match = bool of it) at start > len: the program decides -- zero repetitions and
  // groups pass, bytes and '$' do not ('[^0-9]{0,2}$' at len + 1: None; 'x?' there: the empty match)
  const bool bt_first = op != AT_SEARCH && (h->hp.dev.flags & PF_BT_FIRST) && plan_uses_backtracker(h);
  hipLaunchKernelGGL(k_view_build, dim3(grid_for(n + 1, kBlock)), dim3(kBlock), 0, s, lay, n, start, d_starts, vstart, vlen, vskip,
                     bt_first ? 1 : 0);
  Layout view{lay.data, vstart, 0, nullptr, 0};
  view.vlen = vlen;
  view.vskip = vskip;
  const HostPlan& hp = h->hp;
  const DevPlan& p = hp.dev;
  // '^': DFAEngine.match_first / match_next / is_match (dfa.mojo:1866-1867, 1887-1891, 1828-1829) and
  // OnePassNFA.match_first (onepass.mojo:445) answer None at start > 0; the LazyDFA does not (quirk A.6 #9)
  const bool caret = p.kind == PLAN_DFA ? (p.flags & PF_START_ANCHOR) != 0
                                         : (hp.first_onepass && hp.onepass.has_start_anchor && op != AT_SEARCH);
  int rules = (caret ? 1 : 0) | (bt_first ? 16 : 0);
  // start > len.  match_first: DFAEngine None (dfa.mojo:1922-1923), ".*" None (matcher.mojo:741-745); LazyDFA
  // ._run_lazy and OnePassNFA.match_first run no step and report the empty match (start, start) when the
  // start state accepts (pikevm.mojo:820-867, onepass.mojo:447-488).  match_next: None on every route.
  // is_match: DFAEngine with a first-byte matcher = "start state accepts" (dfa.mojo:1832-1836), without one
  // None; NFA route = bool(match_first).
  const bool start_acc = hp.first_onepass ? p.fa_start_acc != 0 : (p.flags & PF_START_ACCEPTING) != 0;
  if (hp.first_onepass ? start_acc : (p.kind == PLAN_LAZY && start_acc && !(p.flags & PF_START_DEAD))) rules |= 2 | 4;
  if (p.kind == PLAN_DFA && (p.flags & PF_HAS_MATCHER) && start_acc) rules |= 4;
  int rc = MRX_OK;
  if (op == AT_FIRST) rc = run_first_any(h, view, n, ds, de, st);
  else if (op == AT_SEARCH) { rules &= ~2; rc = run_search_any(h, view, n, ds, de, st); }
  else { rules |= 8; rc = run_is_match_any(h, view, n, flag, st); }
  if (rc == MRX_OK) {
    hipLaunchKernelGGL(k_view_fix, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, lay, n, start, d_starts, rules, ds, de, flag);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(scratch_free(vstart, s));
  HIP_TRY(scratch_free(vlen, s));
  HIP_TRY(scratch_free(vskip, s));
  return rc;
}
#define MRX_CHECK_STRIDED()                                                                        \
  if (stride <= 0) return fail(MRX_E_ARGUMENT, "stride must be positive");                         \
  if (!lens && (len < 0 || len > stride)) return fail(MRX_E_ARGUMENT, "len must be in [0, stride]")
int mrx_match_first_at_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n, int32_t start,
                           const int32_t* d_starts, int32_t* s, int32_t* e, void* st) {
  if (!off) return fail(MRX_E_ARGUMENT, "null offsets");
  return run_at(AT_FIRST, h, Layout{d, off, 0, nullptr, 0}, n, start, d_starts, s, e, nullptr, st);
}
int mrx_search_at_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n, int32_t start,
                      const int32_t* d_starts, int32_t* s, int32_t* e, void* st) {
  if (!off) return fail(MRX_E_ARGUMENT, "null offsets");
  return run_at(AT_SEARCH, h, Layout{d, off, 0, nullptr, 0}, n, start, d_starts, s, e, nullptr, st);
}
int mrx_is_match_at_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n, int32_t start,
                        const int32_t* d_starts, uint8_t* f, void* st) {
  if (!off) return fail(MRX_E_ARGUMENT, "null offsets");
  return run_at(AT_IS_MATCH, h, Layout{d, off, 0, nullptr, 0}, n, start, d_starts, nullptr, nullptr, f, st);
}
int mrx_match_first_at_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride, const int32_t* lens, int32_t len,
                                   int64_t n, int32_t start, const int32_t* d_starts, int32_t* s, int32_t* e, void* st) {
  MRX_CHECK_STRIDED();
  return run_at(AT_FIRST, h, Layout{d, nullptr, stride, lens, len}, n, start, d_starts, s, e, nullptr, st);
}
int mrx_search_at_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride, const int32_t* lens, int32_t len,
                              int64_t n, int32_t start, const int32_t* d_starts, int32_t* s, int32_t* e, void* st) {
  MRX_CHECK_STRIDED();
  return run_at(AT_SEARCH, h, Layout{d, nullptr, stride, lens, len}, n, start, d_starts, s, e, nullptr, st);
}
int mrx_is_match_at_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride, const int32_t* lens, int32_t len,
                                int64_t n, int32_t start, const int32_t* d_starts, uint8_t* f, void* st) {
  MRX_CHECK_STRIDED();
  return run_at(AT_IS_MATCH, h, Layout{d, nullptr, stride, lens, len}, n, start, d_starts, nullptr, nullptr, f, st);
}
#undef MRX_CHECK_STRIDED

static int run_captures_any(const mrx_handle* h, const Layout& lay, int64_t n, int32_t* spans, void* st) {
  ScratchScope scratch_scope_((hipStream_t)st);
  if (!h) return fail(MRX_E_ARGUMENT, "null handle");
  const uint32_t fl = h->hp.dev.flags;
  const bool fast_search = (!g_force_generic && (fl & PF_STREAM_SEARCH)) ||
                           (g_force_generic < 2 && (fl & PF_STEP_SEARCH) && !(fl & PF_PREFILTER)) ||
                           (!g_force_generic && (fl & PF_PREFILTER) && (fl & (PF_STREAMABLE | PF_STEP_SEARCH)) &&
                            h->hp.dev.pre_len >= 1 && h->hp.dev.pre_len <= 32);   // literal scan + view (run_search_any)
  if (!fast_search || h->hp.fixed_total < 0 || n <= 0)
    return run_match<OP_CAPTURES>(h, lay, n, spans, nullptr, nullptr, st);
  // search on the streaming kernel (or the windowed stepper), then the groups at their fixed offsets
  hipStream_t s = (hipStream_t)st;
  int32_t* tmp = nullptr;
  HIP_TRY(scratch_alloc((void**)&tmp, sizeof(int32_t) * 2 * n, s));
  int rc = run_search_any(h, lay, n, tmp, tmp + n, st);
  if (rc == MRX_OK) {
    hipLaunchKernelGGL(k_expand_captures, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, h->hp.dev, n, tmp,
                       tmp + n, spans);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(scratch_free(tmp, s));
  return rc;
}
int mrx_captures_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n,
                     int32_t* spans, void* st) {
  if (!off) return fail(MRX_E_ARGUMENT, "null offsets");
  return run_captures_any(h, Layout{d, off, 0, nullptr, 0}, n, spans, st);
}
int mrx_captures_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride, const int32_t* lens,
                             int32_t len, int64_t n, int32_t* spans, void* st) {
  if (stride <= 0) return fail(MRX_E_ARGUMENT, "stride must be positive");
  if (!lens && (len < 0 || len > stride)) return fail(MRX_E_ARGUMENT, "len must be in [0, stride]");
  return run_captures_any(h, Layout{d, nullptr, stride, lens, len}, n, spans, st);
}
int mrx_findall_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n,
                    int64_t* prefix, int32_t* spans, int64_t cap, int64_t* total, void* st) {
  return run_findall(h, Layout{d, off, 0, nullptr, 0}, n, prefix, spans, cap, total, st);
}
int mrx_findall_known_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n, int64_t end_offset,
                          int64_t max_text_len, int64_t* prefix, int32_t* spans, int64_t cap, int64_t* total, void* st) {
  if (end_offset < 0 || max_text_len < 0) return fail(MRX_E_ARGUMENT, "end_offset and max_text_len must not be negative");
  return run_findall(h, Layout{d, off, 0, nullptr, 0}, n, prefix, spans, cap, total, st, false, end_offset, max_text_len);
}
int mrx_findall_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride,
                            const int32_t* lens, int32_t len, int64_t n, int64_t* prefix,
                            int32_t* spans, int64_t cap, int64_t* total, void* st) {
  if (stride <= 0) return fail(MRX_E_ARGUMENT, "stride must be positive");
  if (!lens && (len < 0 || len > stride)) return fail(MRX_E_ARGUMENT, "len must be in [0, stride]");
  return run_findall(h, Layout{d, nullptr, stride, lens, len}, n, prefix, spans, cap, total, st);
}

}  // extern "C"
namespace {
// ---- regex.split (matcher.mojo:1357-1393): the text between successive findall matches ------------------------
// kept[i] = min(matches of text i, maxsplit) (all of them for maxsplit == 0, none for a negative maxsplit)
__global__ __launch_bounds__(kBlock) void k_split_kept(int64_t n, const int64_t* __restrict__ prefix, int64_t maxsplit,
                                                       int32_t* __restrict__ kept) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = prefix[i + 1] - prefix[i];
    kept[i] = (int32_t)(maxsplit == 0 ? c : maxsplit < 0 ? 0 : (c < maxsplit ? c : maxsplit));
  }
}
// piece_prefix[i] = (kept matches of the texts before i) + i: text i yields kept[i] + 1 pieces
__global__ __launch_bounds__(kBlock) void k_split_prefix(int64_t n, const int64_t* __restrict__ kept_prefix,
                                                         int64_t* __restrict__ piece_prefix, int64_t* __restrict__ total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) {
    piece_prefix[i] = kept_prefix[i] + i;
    if (i == n && total) *total = kept_prefix[n] + n;
  }
}
// one lane per PIECE: piece q of text i is [end of match q - 1 (0 for q = 0), start of match q (the text's end behind the
// last kept match)); its text is found by bisection of the piece offsets (coalesced stores whatever the texts hold)
__global__ __launch_bounds__(kBlock) void k_split_pieces(Layout lay, int64_t n, const int64_t* __restrict__ prefix,
                                                         const int32_t* __restrict__ spans, const int64_t* __restrict__ piece_prefix,
                                                         int32_t* __restrict__ pieces, int64_t piece_cap) {
  const int64_t total = piece_prefix[n] < piece_cap ? piece_prefix[n] : piece_cap;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n;   // the last i with piece_prefix[i] <= j
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (piece_prefix[mid] <= j) lo = mid; else hi = mid;
    }
    const int64_t i = lo, q = j - piece_prefix[i];
    const int64_t kept = piece_prefix[i + 1] - piece_prefix[i] - 1;
    const int32_t* sp = spans + 2 * prefix[i];
    const int a = q == 0 ? 0 : sp[2 * (q - 1) + 1];
    const int b = q < kept ? sp[2 * q] : lay.text(i).len;
    *(int2*)(pieces + 2 * j) = make_int2(a, b);
  }
}

int run_split(const mrx_handle* h, const Layout& lay, int64_t n, int64_t maxsplit, int64_t* d_piece_prefix, int32_t* d_pieces,
              int64_t piece_cap, int64_t* total, void* st) {
  hipStream_t s = (hipStream_t)st;
  ScratchScope scratch_scope_(s);
  if (!h) return fail(MRX_E_ARGUMENT, "null handle");
  if (n < 0 || piece_cap < 0 || !d_piece_prefix || (!d_pieces && piece_cap > 0)) return fail(MRX_E_ARGUMENT, "bad arguments");
  if ((uintptr_t)d_pieces & 7) return fail(MRX_E_ARGUMENT, "d_pieces must be 8-byte aligned");
  // findall into scratch: every piece but a text's last ends at a match, so piece_cap holds the spans of any result
  // that fits (maxsplit == 0: pieces = spans + n)
  int64_t* d_prefix = nullptr;
  int32_t* d_spans = nullptr;
  int32_t* d_kept = nullptr;
  int64_t* d_kprefix = nullptr;
  int64_t* d_tot = nullptr;
  const int64_t span_cap = piece_cap > n ? piece_cap : n + 1;
  HIP_TRY(scratch_alloc((void**)&d_prefix, sizeof(int64_t) * (n + 1), s));
  HIP_TRY(scratch_alloc((void**)&d_spans, sizeof(int32_t) * 2 * (size_t)span_cap, s));
  HIP_TRY(scratch_alloc((void**)&d_kept, sizeof(int32_t) * (n > 0 ? n : 1), s));
  HIP_TRY(scratch_alloc((void**)&d_kprefix, sizeof(int64_t) * (n + 1), s));
  HIP_TRY(scratch_alloc((void**)&d_tot, sizeof(int64_t) * 2, s));
  int64_t nspans = 0;
  int rc_f = run_findall(h, lay, n, d_prefix, d_spans, span_cap, &nspans, s);
  if (rc_f == MRX_E_CAPACITY && maxsplit == 0) {   // pieces = matches + n: the need is known
    if (total) *total = nspans + n;
    return fail(MRX_E_CAPACITY, "piece buffer too small: need " + std::to_string(nspans + n));
  }
  if (rc_f == MRX_E_CAPACITY) {   // a limit is on: the pieces may well fit although the matches did not -- once more, all of them
    HIP_TRY(scratch_alloc((void**)&d_spans, sizeof(int32_t) * 2 * (size_t)nspans, s));
    rc_f = run_findall(h, lay, n, d_prefix, d_spans, nspans, &nspans, s);
  }
  if (rc_f != MRX_OK) return rc_f;
  const char* scanned = g_last_kernel;
  if (n > 0) {
    hipLaunchKernelGGL(k_split_kept, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, d_prefix, maxsplit, d_kept);
    if (int rc = device_scan<int32_t>(d_kept, n, d_kprefix, d_tot, s)) return rc;
  } else {
    HIP_TRY(hipMemsetAsync(d_kprefix, 0, sizeof(int64_t), s));
  }
  hipLaunchKernelGGL(k_split_prefix, dim3(grid_for(n + 1, kBlock)), dim3(kBlock), 0, s, n, d_kprefix, d_piece_prefix, d_tot + 1);
  int64_t pieces_total = 0;
  HIP_TRY(hipMemcpyAsync(&pieces_total, d_tot + 1, sizeof pieces_total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (total) *total = pieces_total;
  if (piece_cap > 0 && n > 0)
    hipLaunchKernelGGL(k_split_pieces, dim3(grid_for(pieces_total < piece_cap ? pieces_total : piece_cap, kBlock)), dim3(kBlock), 0, s,
                       lay, n, d_prefix, d_spans, d_piece_prefix, d_pieces, piece_cap);
  HIP_TRY(hipGetLastError());
  g_last_kernel = scanned;   // (the scan that found the separators)
  if (pieces_total > piece_cap) return fail(MRX_E_CAPACITY, "piece buffer too small: need " + std::to_string(pieces_total));
  return MRX_OK;
}
}  // namespace
extern "C" {
int mrx_split_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n, int64_t maxsplit,
                  int64_t* d_piece_prefix, int32_t* d_pieces, int64_t piece_cap, int64_t* total, void* st) {
  return run_split(h, Layout{d, off, 0, nullptr, 0}, n, maxsplit, d_piece_prefix, d_pieces, piece_cap, total, st);
}
int mrx_split_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride, const int32_t* lens, int32_t len, int64_t n,
                          int64_t maxsplit, int64_t* d_piece_prefix, int32_t* d_pieces, int64_t piece_cap, int64_t* total, void* st) {
  if (stride <= 0) return fail(MRX_E_ARGUMENT, "stride must be positive");
  if (!lens && (len < 0 || len > stride)) return fail(MRX_E_ARGUMENT, "len must be in [0, stride]");
  return run_split(h, Layout{d, nullptr, stride, lens, len}, n, maxsplit, d_piece_prefix, d_pieces, piece_cap, total, st);
}

static int run_count_any(const mrx_handle* h, const Layout& lay, int64_t n, int32_t* counts, void* st) {
  ScratchScope scratch_scope_((hipStream_t)st);
  if (!h) return fail(MRX_E_ARGUMENT, "null handle");
  if (int rc = check_search_supported(h)) return rc;
  if (int rc = check_lds(h)) return rc;
  if (int rc = ensure_device(h)) return rc;
  if (n <= 0) return MRX_OK;
  hipStream_t s = (hipStream_t)st;
  if (anchored_at_zero(h) && stream_layout_ok(lay, n)) {
    int32_t* d_se = nullptr;
    HIP_TRY(scratch_alloc((void**)&d_se, sizeof(int32_t) * 2 * n, s));
    ScanTimer tm(s);
    launch_stream<ST_FIRST>(h, lay, n, nullptr, nullptr, nullptr, 0, d_se, d_se + n, s, lay.vlen, lay.vskip);
    HIP_TRY(hipGetLastError());
    tm.stop();
    hipLaunchKernelGGL(k_first_to_counts, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, d_se, counts);
    HIP_TRY(hipGetLastError());
    g_last_kernel = "k_stream_first_count";
    return MRX_OK;
  }
  ScanTimer tm(s);
  if (!g_force_generic && (h->hp.dev.flags & PF_STREAMABLE) && stream_layout_ok(lay, n)) {
    Pieces pc;
    if (int rc = pieces_prepare(h, lay, n, s, &pc)) return rc;
    if (pc.on) {   // long texts: count per piece, then add up each text's pieces
      int32_t* d_vcounts = nullptr;
      HIP_TRY(scratch_alloc((void**)&d_vcounts, sizeof(int32_t) * pc.nv, s));
      launch_stream<ST_COUNT>(h, pc.lay, pc.nv, d_vcounts, nullptr, nullptr, 0, nullptr, nullptr, s, pc.vlen, pc.vskip);
      hipLaunchKernelGGL(k_virt_sum, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, pc.vfirst, d_vcounts, counts);
      HIP_TRY(hipGetLastError());
      HIP_TRY(scratch_free(d_vcounts, s));
      if (int rc = pieces_release(&pc, s)) return rc;
      g_last_kernel = "k_stream_count_pieces";
    } else if (dyn_ok(h, lay, n)) {
      launch_stream_dyn<ST_COUNT>(h, lay, n, counts, nullptr, nullptr, nullptr, nullptr, s, false);
      g_last_kernel = "k_stream_count_dyn";
    } else {
    launch_stream<ST_COUNT>(h, lay, n, counts, nullptr, nullptr, 0, nullptr, nullptr, s);
    g_last_kernel = "k_stream_count";
    }
  } else {
    const bool use_req_route = (h->hp.dev.flags & PF_STEP_REQ) != 0;
    const bool wstep_bits = (h->hp.dev.flags & PF_BSTEP) != 0;
    const bool wstep_lz = (h->hp.dev.flags & PF_LAZY_END) != 0;
    const bool mw_empty = (h->hp.dev.flags & PF_MW_EMPTY) != 0 && mwalk_enabled() && g_force_generic < 2;
    const bool wstep_empty = (h->hp.dev.flags & PF_STEP_EMPTY) != 0 && g_force_generic < 2 && !mw_empty;
    const bool mwalk_req = use_req_route && (h->hp.dev.flags & PF_MWALK_REQ) && mwalk_enabled();
    bool mw_tries = mw_tries_on(h->hp.dev) && !use_req_route && g_force_generic < 2;
    if (t_in_pieces && t_piece_tries >= 0) mw_tries = mw_tries && t_piece_tries == 1;
    int cnt_slot = -1;
    const uint32_t cnt_key = FindallJob::tries_tune_key(lay, n) | (1u << 29);   // (count's own measurement)
    if (mw_tries && !t_in_pieces && n >= 256 && backset_on(h->hp.dev) && !g_tries_always)
      mw_tries = ab_tuner_begin(h, cnt_key, s, "count", &cnt_slot) == 1;
    struct CountTuneEnd {   // (closes the measurement on every way out of this call)
      const mrx_handle* h; uint32_t key; int slot; hipStream_t s;
      ~CountTuneEnd() { ab_tuner_end(h, key, slot, s); }
    } cnt_tune_end{h, cnt_key, cnt_slot, s};
    const bool wstep_mwalk = (mwalk_req || (mwalk_on(h->hp.dev) && !use_req_route) || mw_empty || mw_tries) && !wstep_bits && !wstep_empty;
    const DevPlan pk = mwalk_req ? mwalk_req_plan(h->hp.dev) : h->hp.dev;
    const int wstep_mwalk_k = pk.mw_k;
    const bool wstep_mwalk_pk = false;   // (count keeps no start registers)
    bool req_wave = false;
    int split = 0;
    if (g_force_generic < 2 && !wstep_bits && !t_in_pieces && (h->hp.dev.flags & (PF_STEPPABLE | PF_STEP_REQ)) &&
        !(h->hp.dev.flags & (PF_STEP_BIG | PF_STREAMABLE)) && h->hp.dev.st_nsync > 0 &&
        (!use_req_route || g_long_text_mode == 1)) {
      // long texts: disjoint pieces between synchronising bytes, one lane each (see run_findall)
      Pieces spc;
      if (int rc = pieces_prepare(h, lay, n, s, &spc, -1, -1, /*disjoint=*/true, wstep_mwalk)) return rc;
      if (spc.on && !wstep_mwalk) {
        bool dense = true;
        if (int rc = dense_candidates(h, lay, n, s, &dense)) return rc;
        if (!dense)
          if (int rc = pieces_release(&spc, s)) return rc;
      }
      if (spc.on) {
        int32_t* d_vcounts = nullptr;
        HIP_TRY(scratch_alloc((void**)&d_vcounts, sizeof(int32_t) * spc.nv, s));
        t_in_pieces = true;
        t_piece_tries = (h->hp.dev.flags & PF_MW_TRIES) ? (mw_tries ? 1 : 0) : -1;
        const int rc = run_count_any(h, spc.lay, spc.nv, d_vcounts, st);
        t_in_pieces = false;
        t_piece_tries = -1;
        if (rc != MRX_OK) return rc;
        hipLaunchKernelGGL(k_virt_sum, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, n, spc.vfirst, d_vcounts, counts);
        HIP_TRY(hipGetLastError());
        HIP_TRY(scratch_free(d_vcounts, s));
        return pieces_release(&spc, s);
      }
    }
    if (g_force_generic < 2 && !wstep_bits && !wstep_lz && !mw_tries && !t_in_pieces && (h->hp.dev.flags & (PF_STEPPABLE | PF_STEP_REQ)))
      if (int rc = req_wave_pays(lay, n, use_req_route, s, &req_wave, (h->hp.dev.flags & PF_STEP_BIG) ? nullptr : &split,
                                 (h->hp.dev.flags & PF_STEP_BIG) != 0, wstep_mwalk, backset_on(h->hp.dev) && !wstep_mwalk))
        return rc;
    Layout lay2 = lay;
    lay2.split = split;
    int32_t* d_blimit = nullptr;
    const bool wstep_bm = backset_on(h->hp.dev) && !wstep_mwalk && !wstep_bits && !wstep_empty && !use_req_route && !req_wave &&
                          split == 0 && (h->hp.dev.flags & PF_STEPPABLE);
    const bool wstep_bm_big = wstep_bm && (h->hp.dev.flags & PF_STEP_BIG) != 0;
    if (wstep_bm)
      if (int rc = backscan_marks(h, lay, n, s, &lay2)) return rc;
    const bool bits_fixed = wstep_bits && bits_fixed_on(h->hp.dev);
    if (g_force_generic < 2 && !bits_fixed && !wstep_mwalk && !wstep_bm && (wstep_bits || (!req_wave && split == 0 && !use_req_route && union_pass_for_table_plan(h->hp.dev, false))))
      if (int rc = bscan_limits(h, lay, n, 0, s, &lay2, &d_blimit)) return rc;   // union automaton first
    const bool big_lane = (h->hp.dev.flags & PF_STEP_BIG) && !req_wave && !wstep_mwalk && !wstep_bm;   // -> literal restatement
    if (bits_fixed) {
      if (int rc = bscan_fixed(h, lay, n, 2, s, counts, nullptr, nullptr, 0)) return rc;
      g_last_kernel = "k_bscan_fixed";
    } else if (req_wave) {
      MRX_REQWAVE_LAUNCH(STEP_COUNT, h, lay, n, counts, (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0, s);
      g_last_kernel = "k_req_wave";
    } else if (g_force_generic < 2 && !big_lane && ((h->hp.dev.flags & (PF_STEPPABLE | PF_STEP_REQ | PF_STEP_EMPTY)) || wstep_mwalk)) {
      MRX_WSTEP_LAUNCH(STEP_COUNT, dim3(wstep_grid(n)), dim3(64 * kWsWaves), wstep_lds(pk, wstep_mwalk, wstep_bm_big), s, pk,
                         H_BLOB(h), lay2, n, counts, (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0,
                         (int32_t*)nullptr, (int32_t*)nullptr);
      g_last_kernel = wstep_mwalk ? "k_mwalk" : wstep_bm ? "k_backscan+k_step_count" : wstep_bits ? "k_bstep_count" : wstep_empty ? "k_estep_count" : "k_step_count";
      if (split > 0) {
        MRX_REQWAVE_LAUNCH(STEP_COUNT, h, lay2, n, counts, (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0, s);
        g_last_kernel = "k_step_count+k_req_wave";
      }
    } else {
      Layout layp = lay;
      if (h->hp.dev.flags & PF_BT_SEARCH)
        if (int rc = bt_prepass(h, lay, n, s, &layp)) return rc;
#define MRX_L(B) hipLaunchKernelGGL((k_findall<FA_COUNT, B>), dim3(grid_for(n, kBlock)), dim3(kBlock), lds_for(h), s, \
                                   h->hp.dev, H_BLOB(h), layp, n, counts, (const int64_t*)nullptr, (int32_t*)nullptr, (int64_t)0)
      MRX_BT_DISPATCH(bt_kernel_kind(h, plan_uses_backtracker(h)), MRX_L);
#undef MRX_L
      g_last_kernel = "k_findall_count";
    }
  }
  HIP_TRY(hipGetLastError());
  tm.stop();
  return MRX_OK;
}
int mrx_count_dev(const mrx_handle* h, const uint8_t* d, const int64_t* off, int64_t n,
                  int32_t* counts, void* st) {
  if (!off) return fail(MRX_E_ARGUMENT, "null offsets");
  return run_count_any(h, Layout{d, off, 0, nullptr, 0}, n, counts, st);
}
int mrx_count_strided_dev(const mrx_handle* h, const uint8_t* d, int64_t stride, const int32_t* lens,
                          int32_t len, int64_t n, int32_t* counts, void* st) {
  if (stride <= 0) return fail(MRX_E_ARGUMENT, "stride must be positive");
  if (!lens && (len < 0 || len > stride)) return fail(MRX_E_ARGUMENT, "len must be in [0, stride]");
  return run_count_any(h, Layout{d, nullptr, stride, lens, len}, n, counts, st);
}

}  // extern "C"
namespace {
__global__ __launch_bounds__(kBlock) void k_pitch_offsets(int64_t n, int64_t stride, int64_t* __restrict__ off) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) off[i] = i * stride;
}
}  // namespace
static int sub_any(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count, const Layout& lay_in,
                   int64_t n, int64_t* out_off, uint8_t* out, int64_t out_cap, int64_t* total_bytes, void* st,
                   int64_t known_bytes = -1, int64_t known_max = -1);   // (byte count and longest text, when the caller has them)
extern "C" {
int mrx_sub_dev(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count,
                const uint8_t* d, const int64_t* off, int64_t n, int64_t* out_off, uint8_t* out,
                int64_t out_cap, int64_t* total_bytes, void* st) {
  return sub_any(h, repl, repl_len, count, Layout{d, off, 0, nullptr, 0}, n, out_off, out, out_cap, total_bytes, st);
}
int mrx_sub_known_dev(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count,
                      const uint8_t* d, const int64_t* off, int64_t n, int64_t end_offset, int64_t max_text_len,
                      int64_t* out_off, uint8_t* out, int64_t out_cap, int64_t* total_bytes, void* st) {
  if (end_offset < 0 || max_text_len < 0) return fail(MRX_E_ARGUMENT, "negative end offset / text length");
  return sub_any(h, repl, repl_len, count, Layout{d, off, 0, nullptr, 0}, n, out_off, out, out_cap, total_bytes, st,
                 end_offset, max_text_len);
}
// Texts at a fixed pitch.  Rows without padding (len == stride, no per-text lengths) are a CSR batch whose offsets
// are i * stride: they are written once on the device and the call takes every fast path of mrx_sub_dev; padded
// rows run on the lane-per-text kernels (the spans route assembles its output from CSR offsets).
int mrx_sub_strided_dev(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count,
                        const uint8_t* d, int64_t stride, const int32_t* d_lens, int32_t len, int64_t n,
                        int64_t* out_off, uint8_t* out, int64_t out_cap, int64_t* total_bytes, void* st) {
  // (`len` is ignored when d_lens is given, as in every other strided entry point)
  if (stride <= 0 || (!d_lens && (len < 0 || len > stride))) return fail(MRX_E_ARGUMENT, "bad pitch / length");
  if (!d_lens && (int64_t)len == stride && n > 0) {
    ScratchScope scope_((hipStream_t)st);
    int64_t* d_off = nullptr;
    HIP_TRY(scratch_alloc((void**)&d_off, sizeof(int64_t) * (n + 1), (hipStream_t)st));
    hipLaunchKernelGGL(k_pitch_offsets, dim3(grid_for(n + 1, kBlock)), dim3(kBlock), 0, (hipStream_t)st, n, stride, d_off);
    HIP_TRY(hipGetLastError());
    return sub_any(h, repl, repl_len, count, Layout{d, d_off, 0, nullptr, 0}, n, out_off, out, out_cap, total_bytes, st,
                   n * stride, stride);
  }
  return sub_any(h, repl, repl_len, count, Layout{d, nullptr, stride, d_lens, len}, n, out_off, out, out_cap, total_bytes, st);
}
}  // extern "C"
static int sub_any(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count, const Layout& lay_in,
                   int64_t n, int64_t* out_off, uint8_t* out, int64_t out_cap, int64_t* total_bytes, void* st,
                   int64_t known_bytes, int64_t known_max) {
  const uint8_t* d = lay_in.data;
  const int64_t* off = lay_in.offsets;
  ScratchScope scratch_scope_((hipStream_t)st);
  if (!h) return fail(MRX_E_ARGUMENT, "null handle");
  if (n < 0) return fail(MRX_E_ARGUMENT, "negative batch size");
  const std::string r(repl ? repl : "", repl_len);
  const bool groups = repl_has_group_refs(r);
  // group references on a pattern outside the fixed-width form: every match comes from
  // NFAEngine.match_next_with_groups (matcher.mojo:1781-1822), not from the search engines
  const bool general_groups = groups && h->hp.fixed_total < 0;
  std::vector<ReplSeg> tpl;
  if (general_groups) {
    if (!h->hp.bt.ok)
      return fail(MRX_E_UNSUPPORTED,
                  "sub() with \\1..\\9 on this pattern uses NFAEngine.match_next_with_groups (recursive "
                  "backtracking matcher, nfa.mojo:500-574); its flat-program form does not cover: " +
                      (h->hp.bt.why_not.empty() ? std::string("'.*'") : h->hp.bt.why_not));
  } else if (int rc = check_search_supported(h)) {
    return rc;
  }
  if (groups) tpl = parse_repl_template(r);
  if (int rc = check_lds(h)) return rc;
  if (int rc = ensure_device(h)) return rc;
  hipStream_t s = (hipStream_t)st;
  // \1..\9 on a deterministic chain whose matches are the table walk's: spans of the plain search, groups from the
  // leaves' runs (k_subc_sizes / k_subc_emit); batches with a text or an output beyond the tiles stay the interpreter's
  if (general_groups && h->hp.chain.ok && !g_force_generic && n > 0 && off && count >= 0 &&
      (h->hp.dev.flags & (PF_STREAM_SEARCH | PF_STEP_SEARCH)) && h->hp.why_no_search.empty()) {
    const int rc = sub_chain_from_spans(h, Layout{d, off, 0, nullptr, 0}, n, r, tpl, count, out_off, out, out_cap,
                                        total_bytes, s, known_bytes, known_max);
    if (rc != kSubsRetryGeneric) return rc;
  }
  // sub() iterates match_next from the previous match end; on the plain route that is exactly the
  // findall sequence, so the spans of the streaming kernel or of the windowed stepper serve it.
  // (Not with a memchr prefilter, which only match_next consults; not for exact literals, whose
  // findall spans overlap while sub's own search loop does not.)
  const uint32_t sfl = h->hp.dev.flags;
  const bool spans_ok = !(sfl & PF_EXACT_LITERAL) &&
                        ((!g_force_generic && (sfl & PF_STREAM_SEARCH)) ||
                         (g_force_generic < 2 && (sfl & PF_STEP_SEARCH) && !(sfl & PF_PREFILTER)));
  // (group templates of "concat" patterns that are not purely groups: texts of exactly fixed_total bytes take the
  // whole-text shortcut, which only sub_text() has)
  const bool shortcut_differs = groups && h->hp.fixed_concat && !h->hp.fixed_pure;
  if (spans_ok && !general_groups && !shortcut_differs && n > 0 && off) {
    // replacement as a fixed-length byte map
    std::vector<uint16_t> rmap;
    int group_reach = 0;   // how far behind a match's start the template reads
    if (groups) {
      for (const ReplSeg& sg : tpl) {
        if (sg.group_ref > 0 && sg.group_ref <= h->hp.fixed_ngroups) {
          for (int j = 0; j < h->hp.fixed_w[sg.group_ref]; ++j)
            rmap.push_back((uint16_t)(0x8000 | (h->hp.fixed_off[sg.group_ref] + j)));
          group_reach = std::max(group_reach, h->hp.fixed_off[sg.group_ref] + h->hp.fixed_w[sg.group_ref]);
        }
        else
          for (int j = 0; j < sg.length; ++j) rmap.push_back((uint8_t)r[sg.start + j]);
      }
    } else {
      for (unsigned char ch : r) rmap.push_back(ch);
    }
    // (nothing but (\d{N}) groups: every match is fixed_total bytes long and holds all its windows)
    if (h->hp.fixed_pure) group_reach = 0;
    if (rmap.size() <= 4096 && h->hp.fixed_total < 0x7FFF) {
      const int rc = sub_from_spans(h, Layout{d, off, 0, nullptr, 0}, n, rmap, count, out_off, out, out_cap,
                                    total_bytes, s, group_reach, known_bytes, known_max);
      if (rc != kSubsRetryGeneric) return rc;   // else: a group reaches behind its text, the lane-per-text form cuts it
    }
  }
  uint8_t* d_repl = nullptr;
  ReplSeg* d_tpl = nullptr;
  int64_t *d_sizes = nullptr, *d_total = nullptr;
  HIP_TRY(scratch_alloc((void**)&d_repl, r.size() + 16, s));
  HIP_TRY(scratch_alloc((void**)&d_tpl, sizeof(ReplSeg) * (tpl.size() + 1), s));
  HIP_TRY(scratch_alloc((void**)&d_sizes, sizeof(int64_t) * (n > 0 ? n : 1), s));
  HIP_TRY(scratch_alloc((void**)&d_total, sizeof(int64_t), s));
  if (!r.empty()) HIP_TRY(hipMemcpyAsync(d_repl, r.data(), r.size(), hipMemcpyHostToDevice, s));
  if (!tpl.empty())
    HIP_TRY(hipMemcpyAsync(d_tpl, tpl.data(), sizeof(ReplSeg) * tpl.size(), hipMemcpyHostToDevice, s));
  Layout lay = lay_in;
  const bool sub_bt = plan_uses_backtracker(h) || general_groups;
  if ((h->hp.dev.flags & PF_BT_SEARCH) || general_groups) {
    const Layout plain = lay;
    if (int rc = bt_prepass(h, plain, n, s, &lay)) return rc;
  }
  if (n > 0) {
    ScanTimer tm(s);
#define MRX_L(B) hipLaunchKernelGGL((k_sub<SUB_SIZE, B>), dim3(grid_for(n, kBlock)), dim3(kBlock), lds_for(h), s,                     \
                                   h->hp.dev, H_BLOB(h), lay, n, d_repl, (int)r.size(), general_groups ? 2 : groups ? 1 : 0, d_tpl, \
                                   (int)tpl.size(), (long long)count, d_sizes, (const int64_t*)nullptr, (uint8_t*)nullptr, (int64_t)0)
    MRX_BT_DISPATCH(bt_kernel_kind(h, sub_bt), MRX_L);
#undef MRX_L
    g_last_kernel = "k_sub_size";
    HIP_TRY(hipGetLastError());
    tm.stop();
  }
  if (int rc = device_scan<int64_t>(d_sizes, n, out_off, d_total, s)) return rc;
  int64_t tot = 0;
  HIP_TRY(hipMemcpyAsync(&tot, d_total, sizeof tot, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (total_bytes) *total_bytes = tot;
  int rc = MRX_OK;
  if (tot > out_cap) {
    rc = fail(MRX_E_CAPACITY, "output buffer too small: need " + std::to_string(tot));
  } else if (n > 0 && tot > 0) {
#define MRX_L(B) hipLaunchKernelGGL((k_sub<SUB_EMIT, B>), dim3(grid_for(n, kBlock)), dim3(kBlock), lds_for(h), s,                     \
                                   h->hp.dev, H_BLOB(h), lay, n, d_repl, (int)r.size(), general_groups ? 2 : groups ? 1 : 0, d_tpl, \
                                   (int)tpl.size(), (long long)count, (int64_t*)nullptr, out_off, out, out_cap)
    MRX_BT_DISPATCH(bt_kernel_kind(h, sub_bt), MRX_L);
#undef MRX_L
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(scratch_free(d_repl, s));
  HIP_TRY(scratch_free(d_tpl, s));
  HIP_TRY(scratch_free(d_sizes, s));
  HIP_TRY(scratch_free(d_total, s));
  return rc;
}

extern "C" {
// ---- host-buffer wrappers -----------------------------------------------------------

int mrx_match_first_batch(const mrx_handle* h, const uint8_t* data, const int64_t* off, int64_t n,
                          int32_t* start, int32_t* end) {
  DevBatch b; DevBuf<int32_t> s, e;
  if (int rc = b.upload(data, off, n)) return rc;
  if (int rc = s.alloc(n)) return rc;
  if (int rc = e.alloc(n)) return rc;
  if (int rc = mrx_match_first_dev(h, b.data, b.offsets, n, s.p, e.p, nullptr)) return rc;
  HIP_TRY(hipMemcpy(start, s.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(end, e.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  return MRX_OK;
}
int mrx_search_batch(const mrx_handle* h, const uint8_t* data, const int64_t* off, int64_t n,
                     int32_t* start, int32_t* end) {
  DevBatch b; DevBuf<int32_t> s, e;
  if (int rc = b.upload(data, off, n)) return rc;
  if (int rc = s.alloc(n)) return rc;
  if (int rc = e.alloc(n)) return rc;
  if (int rc = mrx_search_dev(h, b.data, b.offsets, n, s.p, e.p, nullptr)) return rc;
  HIP_TRY(hipMemcpy(start, s.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(end, e.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  return MRX_OK;
}
int mrx_is_match_batch(const mrx_handle* h, const uint8_t* data, const int64_t* off, int64_t n,
                       uint8_t* flag) {
  DevBatch b; DevBuf<uint8_t> f;
  if (int rc = b.upload(data, off, n)) return rc;
  if (int rc = f.alloc(n)) return rc;
  if (int rc = mrx_is_match_dev(h, b.data, b.offsets, n, f.p, nullptr)) return rc;
  HIP_TRY(hipMemcpy(flag, f.p, n, hipMemcpyDeviceToHost));
  return MRX_OK;
}
int mrx_findall_batch(const mrx_handle* h, const uint8_t* data, const int64_t* off, int64_t n,
                      int64_t* counts_prefix, int32_t* spans, int64_t span_cap, int64_t* total) {
  DevBatch b; DevBuf<int64_t> pre; DevBuf<int32_t> sp;
  if (int rc = b.upload(data, off, n)) return rc;
  if (int rc = pre.alloc(n + 1)) return rc;
  if (int rc = sp.alloc(2 * (size_t)span_cap)) return rc;
  int64_t tot = 0;
  const int rc = mrx_findall_dev(h, b.data, b.offsets, n, pre.p, sp.p, span_cap, &tot, nullptr);
  if (total) *total = tot;
  if (rc != MRX_OK && rc != MRX_E_CAPACITY) return rc;
  HIP_TRY(hipMemcpy(counts_prefix, pre.p, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost));
  if (rc == MRX_OK && tot > 0)
    HIP_TRY(hipMemcpy(spans, sp.p, sizeof(int32_t) * 2 * tot, hipMemcpyDeviceToHost));
  return rc;
}
int mrx_split_batch(const mrx_handle* h, const uint8_t* data, const int64_t* off, int64_t n, int64_t maxsplit,
                    int64_t* piece_prefix, int32_t* pieces, int64_t piece_cap, int64_t* total) {
  DevBatch b; DevBuf<int64_t> pre; DevBuf<int32_t> pc;
  if (int rc = b.upload(data, off, n)) return rc;
  if (int rc = pre.alloc(n + 1)) return rc;
  if (int rc = pc.alloc(2 * (size_t)(piece_cap > 0 ? piece_cap : 1))) return rc;
  int64_t tot = 0;
  const int rc = mrx_split_dev(h, b.data, b.offsets, n, maxsplit, pre.p, pc.p, piece_cap, &tot, nullptr);
  if (total) *total = tot;
  if (rc != MRX_OK) return rc;
  HIP_TRY(hipMemcpy(piece_prefix, pre.p, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost));
  if (tot > 0) HIP_TRY(hipMemcpy(pieces, pc.p, sizeof(int32_t) * 2 * tot, hipMemcpyDeviceToHost));
  return MRX_OK;
}
int mrx_captures_batch(const mrx_handle* h, const uint8_t* data, const int64_t* off, int64_t n,
                       int32_t* spans) {
  DevBatch b; DevBuf<int32_t> sp;
  if (int rc = b.upload(data, off, n)) return rc;
  const size_t per = (size_t)(mrx_num_groups(h) + 1) * 2;
  if (int rc = sp.alloc(per * n)) return rc;
  if (int rc = mrx_captures_dev(h, b.data, b.offsets, n, sp.p, nullptr)) return rc;
  HIP_TRY(hipMemcpy(spans, sp.p, sizeof(int32_t) * per * n, hipMemcpyDeviceToHost));
  return MRX_OK;
}
int mrx_sub_batch(const mrx_handle* h, const char* repl, size_t repl_len, int64_t count,
                  const uint8_t* data, const int64_t* off, int64_t n, int64_t* out_offsets,
                  uint8_t* out_data, int64_t out_cap, int64_t* total_bytes) {
  DevBatch b; DevBuf<int64_t> oo; DevBuf<uint8_t> od;
  if (int rc = b.upload(data, off, n)) return rc;
  if (int rc = oo.alloc(n + 1)) return rc;
  if (int rc = od.alloc((size_t)out_cap)) return rc;
  int64_t tot = 0;
  const int rc = mrx_sub_dev(h, repl, repl_len, count, b.data, b.offsets, n, oo.p, od.p, out_cap,
                             &tot, nullptr);
  if (total_bytes) *total_bytes = tot;
  if (rc != MRX_OK && rc != MRX_E_CAPACITY) return rc;
  HIP_TRY(hipMemcpy(out_offsets, oo.p, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost));
  if (rc == MRX_OK && tot > 0) HIP_TRY(hipMemcpy(out_data, od.p, (size_t)tot, hipMemcpyDeviceToHost));
  return rc;
}

void mrx_timing_reset(void) { g_scan_ms = 0; g_scan_launches = 0; }
void mrx_timing_enable(int on) { g_timing = on != 0; }
double mrx_timing_scan_ms(int64_t* launches) {
  if (launches) *launches = g_scan_launches;
  return g_scan_launches ? g_scan_ms / (double)g_scan_launches : 0.0;
}
const char* mrx_last_kernel_name(void) { return g_last_kernel; }
void mrx_debug_force_generic(int on) { g_force_generic = on < 0 ? 0 : on > 2 ? 2 : on; }
void mrx_debug_long_text_kernels(int mode) { g_long_text_mode = mode < 0 ? 0 : mode > 3 ? 0 : mode; }
void mrx_debug_rec_skew(int64_t bytes) { g_rec_skew = bytes < 0 ? 0 : (bytes & ~int64_t(15)); }
void mrx_debug_fused_findall(int mode) { g_fused = mode < 0 ? 0 : mode > 2 ? 0 : mode; }
void mrx_debug_stream_bits(int on) { mrx::stream_bits_set_mode(on); }
void mrx_debug_stream_bits_trace(int64_t* d_trace) { mrx::stream_bits_set_trace(d_trace); }
void mrx_debug_dynamic_texts(int mode) { g_dyn_mode = mode < 0 ? 0 : mode > 2 ? 0 : mode; }
void mrx_debug_split_findall(int on) { g_split_findall = on ? 1 : 0; }
void mrx_debug_dense_rows(int mode) { g_dense_rows = mode; }
void mrx_debug_tries_always(int on) { g_tries_always = on ? 1 : 0; }
void mrx_debug_chain_sub_general(int on) { g_chain_sub_general = on ? 1 : 0; }
int mrx_testing_emptywalk_findall(const mrx_handle* h, const uint8_t* text, int len, int32_t* spans, int cap) {
  if (!h || !h->hp.ew2_ok || len < 0) return -1;
  const std::vector<std::pair<int, int>> v = emptywalk2_run(h->hp.ew2, text, len);
  for (size_t k = 0; k < v.size() && (int)k < cap; ++k) { spans[2 * k] = v[k].first; spans[2 * k + 1] = v[k].second; }
  return (int)v.size();
}
void mrx_debug_subs_group(int g) { g_subs_group = (g == 0 || g == 16 || g == 32 || g == 64 || g == 256) ? g : -1; }
void mrx_debug_litscan_pieces(int mode) { g_litscan_pieces = (mode == 0 || mode == 1) ? mode : 2; }
void mrx_debug_multiwalk(int mode) { g_mwalk_mode = mode == 2 ? 2 : 0; g_mwalk_pk = mode == 3 ? 0 : 1; }
void mrx_release_scratch(void) { scratch_release_all(); }
size_t mrx_debug_scratch_bytes(void) { return scratch_bytes_reserved(); }

}  // extern "C"
