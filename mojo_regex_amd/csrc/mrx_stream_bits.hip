// findall of a streamable plan over short texts (fixed pitch, at most 1 KiB each) in ONE launch, with no
// intermediate in memory: k_stream_bits.  OPT-IN (MRX_STREAM_BITS=1 / mrx_debug_stream_bits(1)): built in round 4
// to remove the record round trip of the three-launch form, parity green, and measured SLOWER than it -- the numbers
// and the reason are at the end of this comment and in profiles/r04_stream_bits.md.
//
// The three-launch form (k_stream_findall -> prefix sums -> k_decode, mrx_kernels.hip) writes one 16-byte event
// record per 32 text bytes that hold a match end and reads it back: 0.46 GB of the 1.74 GB the headline step
// moves.  Round 2's single launch (ST_FUSED) kept those records in memory too -- 4096 resident wavefronts hold
// three times what L2 holds, so they streamed out and back all the same.  Here the events of a text never leave
// the lane that found them: the two event bits per byte (NEWSTART, EMIT -- the same event words F the records
// carry) of a text of up to 1024 bytes are 64 registers.  When a wavefront has walked its 64 texts it
//   1. counts its matches and publishes the sum (one 8-byte descriptor + one add to its group's word),
//   2. compacts the event words that hold a match end into LDS as ITEMS {event word, start carried in | word |
//      index of its first match within the wavefront}, text by text (a branch-free pass over the registers),
//   3. obtains the number of matches of all texts before its own by decoupled look-back (lookback_quiet,
//      mrx_lookback.hpp: one lane polls one word),
//   4. expands the items with lane = item (every lane has work; neighbouring lanes hold neighbouring words of one
//      text, so their 8-byte stores -- straight to the spans' final place -- fall into the same few cache lines),
//      and writes the 64 CSR offsets.
// Traffic: the texts once, the spans once, 8 bytes per text of offsets (1.27 GB on the headline batch).
//
// Tasks (4 consecutive groups of 64 texts per workgroup) are handed out in text order by a ticket counter, so a
// wavefront only ever waits for tasks that running wavefronts hold -- whatever else occupies the device.  One
// atomic per workgroup and round, asked for by wavefront 0 when its texts are walked (the answer is back before its
// expansion is through) and handed to the other three wavefronts through a ring in LDS.  Asked for a round AHEAD
// (as round 2's fused form did) a ticket is a task nobody works on for a whole round while every later task waits
// for its count: the median wavefront then waited 15-33 us of a 50 us round.
//
// Measured (MI355X, headline batch 2^20 x 1 KiB, `[a-z]+\d+`; tools/r04_ablate.py, r04_trace.py, r04_ab_lengths.py):
//   whole step 0.355 ms against 0.342 ms for the three launches on the same box; 512 / 256 / 128-byte texts 1.5 / 2.2 /
//   3.0 x slower (the per-task hand-offs do not shrink with the task).  Scan + publish alone 0.213 ms (the old scan
//   kernel with its record stores: 0.217); + expansion 0.29; + look-back 0.33-0.34.  Per 64-text task the scan is
//   ~3600 VALU instructions and the expansion ~4000 (per-word passes over 64 registers whether they hold an event or
//   not, and a `while (match ends left)` whose trip count is the maximum over 64 items): the launch is bound by
//   instruction issue, not by the 1.27 GB it moves, and 3 wavefronts per SIMD (168 registers: 64 of them the
//   bitmap) leave 16384 tasks on 3072 wavefronts = 5.33 rounds, the last of them a third full.
//
// Reference semantics: DFAEngine.match_all, /root/reference src/regex/dfa.mojo:2028-2130 -- the restart-per-
// position loop, as proved equal to the single left-to-right walk by check_streamable() (mrx_plan.cpp).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/mrx.h"
#include "mrx_internal.hpp"
#include "mrx_lookback.hpp"
#include "mrx_plan.hpp"

namespace mrx {
namespace {

constexpr int kSbWaves = 4;
constexpr int kSbChunk = 128;                    // bytes of every text staged per step: one cache line
constexpr int kSbRowPitch = kSbChunk + 16;       // +16: the per-lane 16-byte read-back is bank-conflict free
constexpr int kSbTileBytes = 64 * kSbRowPitch;   // 9216
#ifndef MRX_SB_WAVE_LDS
#define MRX_SB_WAVE_LDS 12800
#endif
constexpr int kSbWaveLds = MRX_SB_WAVE_LDS;                // per wavefront: the text tile, then the items of the expansion
static_assert(kSbTileBytes <= kSbWaveLds, "the text tile lives at the start of the wavefront's LDS");

struct SbPlan {   // the few DevPlan fields the kernel reads (the whole struct by value costs ~60 SGPRs)
  int32_t off_col;          // u16 column table [256]: DevPlan::off_stcol32 (code columns) or off_stcol (4-bit columns)
  int32_t reset_byte;       // DevPlan::st_reset_byte
  uint32_t acc;             // st_acc32 (code columns) / st_accept_mask
  int32_t fixed_len;        // DevPlan::st_fixed_len
};

struct SbArgs {   // in device memory, read where used: as kernel arguments they would sit in SGPRs across the scan
  unsigned long long* ctrl;   // [0] ticket counter, [1] error word, [2..] look-back words (mrx_lookback.hpp)
  int64_t* prefix;            // [n + 1] CSR offsets (output)
  int32_t* spans;             // [span_cap][2] (output)
  int64_t span_cap;
  int64_t* total_out;
  int32_t phase_mode;    // how a workgroup picks its start delay (see k_stream_bits); 0 = none
  int32_t phase_sleeps;  // s_sleep(127) periods (3.9 us each at 2.1 GHz) per phase step
  int64_t* trace;        // measurement only (mrx_debug_stream_bits_trace): per task {start, scan end, base known, done}, 10 ns ticks
  int32_t debug;   // measurement only (MRX_SB_DEBUG): 1 no expansion, 2 no look-back, 4 no span stores, 8 no offsets
};

__global__ __launch_bounds__(256) void k_sb_init(unsigned long long* __restrict__ ctrl, int64_t words,
                                                 SbArgs* __restrict__ dst, SbArgs args) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x)
    ctrl[i] = 0ull;
  if (blockIdx.x == 0 && threadIdx.x == 0) *dst = args;
}

__device__ __forceinline__ void sb_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// AUTO = 1: 4-bit byte columns (<= 4 states), AUTO = 5: code columns (DevPlan::off_stcol32) -- the two automaton
// forms of k_stream_findall that keep the per-byte step in registers.  NW: event words per text (16 text bytes
// each): texts of at most 16 * NW bytes.
template <int AUTO, int NW>
__global__ __launch_bounds__(64 * kSbWaves, 3) void k_stream_bits(
    SbPlan p, const uint8_t* __restrict__ blob, const uint8_t* __restrict__ data, uint32_t stride,
    const int32_t* __restrict__ lens, int32_t common_len, int64_t n, const SbArgs* __restrict__ ap) {
  static_assert(NW % 8 == 0 && NW <= 64, "whole 128-byte chunks, at most 1 KiB");
  constexpr int LPR = kSbChunk / 16;   // lanes that cover one text row in a load instruction
  constexpr int RPI = 64 / LPR;        // text rows per load instruction
  constexpr int NL = 64 / RPI;         // load instructions per chunk
  __shared__ __align__(16) uint8_t wave_lds[kSbWaves][kSbWaveLds];
  __shared__ __align__(16) uint16_t col_lds[256];
  __shared__ __align__(16) uint4 pmask[17];   // pmask[x]: the first x bytes of a 16-byte group set
  __shared__ unsigned long long blk_ticket[4];   // ring: {round + 1, first 64-text group of the round}
  __shared__ int wave_round[kSbWaves];           // the round each wavefront has taken its task of
  if (threadIdx.x < 17) {
    const int x = threadIdx.x;
    uint32_t w[4];
    for (int j = 0; j < 4; ++j) {
      const int nb = x - 4 * j;
      w[j] = nb <= 0 ? 0u : nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
    }
    pmask[x] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  {
    const uint16_t* src = (const uint16_t*)(blob + p.off_col);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) col_lds[i] = src[i];
  }
  if (threadIdx.x < kSbWaves) wave_round[threadIdx.x] = -1;
  if (threadIdx.x == 0) {
    // Phases.  Every task is the same amount of work, so left alone all resident wavefronts scan (memory) at the same
    // time and expand their bitmaps (VALU, LDS) at the same time, and neither resource is busy more than half of the
    // time.  The three workgroups of a CU therefore start a third of a task apart -- BEFORE they take their first
    // ticket, so ticket order stays start order and a later phase only ever waits for earlier ones.
    const int pm = ap->phase_mode;
    if (pm) {
      const uint32_t slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 15u;   // HW_ID.WAVE_ID: my slot on the SIMD
      const uint32_t ph = pm == 1 ? slot % 3u : pm == 2 ? blockIdx.x % 3u : (blockIdx.x / (gridDim.x / 3u ? gridDim.x / 3u : 1u)) % 3u;
      for (uint32_t k = 0; k < ph * (uint32_t)ap->phase_sleeps; ++k) __builtin_amdgcn_s_sleep(127);
    }
    blk_ticket[1] = blk_ticket[2] = blk_ticket[3] = 0ull;
    blk_ticket[0] = (1ull << 32) | (uint32_t)__hip_atomic_fetch_add(ap->ctrl, (unsigned long long)kSbWaves, __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const bool use_fill = AUTO == 5 || p.reset_byte >= 0;
  const uint32_t fillw = (uint32_t)(p.reset_byte & 0xFF) * 0x01010101u;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint8_t* tile = wave_lds[wave];
  const int64_t nw = (n + 63) >> 6;
  const int seg = lane % LPR;
  const int rsub = lane / LPR;

  for (uint32_t round = 0;; ++round) {
    unsigned long long tv;
    while (true) {
      tv = *(volatile unsigned long long*)&blk_ticket[round & 3u];
      if ((uint32_t)(tv >> 32) == round + 1u) break;
      __builtin_amdgcn_s_sleep(1);
    }
    const int64_t first = (int64_t)__builtin_amdgcn_readfirstlane((uint32_t)tv);
    if (lane == 0) *(volatile int*)&wave_round[wave] = (int)round;
    if (first >= nw) break;
    const int64_t w = first + wave;
    if (w >= nw) continue;   // (never wavefront 0: the ring is kept going by it)
    int64_t* const trace = ap->trace;
    if (trace && lane == 0) trace[4 * w] = (int64_t)wall_clock64();
    const int64_t base_text = w << 6;
    const int64_t my_text = base_text + lane;
    const bool live = my_text < n;
    const int my_len = live ? (lens ? lens[my_text] : common_len) : 0;
    int max_len = my_len;
    for (int off = 32; off > 0; off >>= 1) max_len = max(max_len, __shfl_xor(max_len, off));
    max_len = __builtin_amdgcn_readfirstlane(max_len);
    if (max_len > 16 * NW) max_len = 16 * NW;   // (the host never launches this form on longer texts)

    // rows this lane stages: texts RPI * j + lane / LPR of the wavefront.  Address = wave-uniform base + one 32-bit
    // lane offset + a wave-uniform step per load instruction; rows past the end of the batch (the last task only)
    // read row 0 instead.  The pitch is a multiple of 16, so a 16-byte load that starts inside a row stays inside
    // it; past the row end the first bytes are read instead (and ignored).
    const uint8_t* wbase = data + base_text * (int64_t)stride;
    const int rows_here = n - base_text < 64 ? (int)(n - base_text) : 64;
    const uint32_t roff0 = (uint32_t)rsub * stride + (uint32_t)seg * 16u;
#define SB_LOAD_CHUNK(CB)                                                                    \
    do {                                                                                     \
      uint32_t cb_ = (uint32_t)(CB);                                                         \
      if (cb_ + (uint32_t)seg * 16u >= stride) cb_ = (uint32_t)0 - (uint32_t)(seg * 16);      \
      if (rows_here == 64) {                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < NL; ++j_)                                    \
          v[j_] = MRX_LDG((const uint4*)(wbase + (uint32_t)(j_ * RPI) * stride + (uint32_t)(roff0 + cb_))); \
      } else {                                                                               \
        int rs_ = rsub;   /* (opaque: nothing of the last task's path is worth a register across the rounds) */ \
        asm volatile("" : "+v"(rs_));                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NL; ++j_) {                                  \
          const bool in_ = RPI * j_ + rs_ < rows_here;                                       \
          v[j_] = MRX_LDG((const uint4*)(wbase + (uint32_t)((in_ ? (uint32_t)(j_ * RPI) * stride + roff0 : (uint32_t)seg * 16u) + cb_))); \
        }                                                                                    \
      }                                                                                      \
    } while (0)

    uint32_t bm[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) bm[i] = 0u;
    uint32_t q4 = 0;        // AUTO 1: 4 * state; AUTO 5: bit offset of the state's field (low two bits = its code)
    uint32_t q_codes = 0;   // AUTO 5: the code word of the previous group
    unsigned long long tk_next = 0ull;
    uint4 v[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) v[j] = make_uint4(0, 0, 0, 0);
    if (max_len > 0) SB_LOAD_CHUNK(0);
    uint8_t* wr = tile + rsub * kSbRowPitch + seg * 16;
    int ci = 0;   // chunk index (wave uniform)
    for (int cbase = 0; cbase < max_len; cbase += kSbChunk, ++ci) {
#pragma unroll
      for (int j = 0; j < NL; ++j) *(uint4*)(wr + j * RPI * kSbRowPitch) = v[j];
      sb_wave_sync();   // the tile is private to this wavefront
      if (cbase + kSbChunk < max_len) SB_LOAD_CHUNK(cbase + kSbChunk);   // next chunk in flight
      const int lim = my_len - cbase;   // bytes [0, lim) of this chunk are text (lim may be <= 0 or > chunk)
      const bool all_inside = __all(lim >= kSbChunk);
      const bool full = all_inside || use_fill;   // branch-free steps: no byte needs a predicate
      uint32_t f[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const uint4 wv = *(const uint4*)(tile + lane * kSbRowPitch + g * 16);
        uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
        if (use_fill && !all_inside) {   // wave uniform: bytes behind the text become the reset byte
          const int b = min(max(lim - g * 16, 0), 16);
          const uint4 pb = pmask[b];
          const uint32_t m[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) words[j] = (words[j] & m[j]) | (fillw & ~m[j]);
        }
        uint32_t F = 0;
        if (AUTO == 5) {
          // code columns: one shift per byte, the state's 2-bit code recorded per byte, events from two code words
          uint32_t cv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) cv[k] = col_lds[(words[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
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
          for (int k = 0; k < 16; ++k) cv[k] = col_lds[(words[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const uint32_t e = cv[k] >> q4;
            q4 = e & 0xCu;
            F = __builtin_amdgcn_alignbit(e, F, 2);   // F = (F >> 2) | (e << 30)
          }
        } else {
          // a text of this wavefront ends inside the chunk and the plan has no reset byte: bytes past the end are no-ops
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const uint32_t b = (words[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            uint32_t e = col_lds[b] >> q4;
            if (g * 16 + k >= lim) e = q4;   // keep the state, no event bits
            q4 = e & 0xCu;
            F = __builtin_amdgcn_alignbit(e, F, 2);
          }
        }
        // one group at a time: the step chain is serial through q4, and left alone the compiler issues the table
        // lookups of all eight groups first (128 live registers) and runs the chain behind them
        asm volatile("" : "+v"(F), "+v"(q4));
        f[g] = F;
      }
      // the chunk's event words into the text's bitmap.  Register indices must be constants; a switch over the
      // chunk number was tried (eight moves per chunk instead of NW) and left seven spills inside this loop, so the
      // bitmap moves down by one chunk per step and the new words enter at the top (the words of a text of fewer
      // than NW / 8 chunks end up at the top: `wshift` below).
#pragma unroll
      for (int i = 0; i + 8 < NW; ++i) bm[i] = bm[i + 8];
#pragma unroll
      for (int g = 0; g < 8; ++g) bm[NW - 8 + g] = f[g];
      __builtin_amdgcn_wave_barrier();
    }
#undef SB_LOAD_CHUNK
    // The next round's tasks are asked for NOW, when this round's texts are walked -- not a round ahead: a ticket
    // taken early is a task nobody works on for a whole round while every task behind it waits for its count
    // (measured: with the ticket taken at the start of the round before, the median wavefront waited 15-33 us of a
    // 50 us round for the counts in front of it, profiles/r04_stream_bits.md).  The answer is back before the
    // expansion below is through.
    if (wave == 0 && lane == 0)
      tk_next = __hip_atomic_fetch_add(ap->ctrl, (unsigned long long)kSbWaves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- the task's texts are walked -------------------------------------------------------------------
    // end of text: a walk that is in an accepting state ends at len (bytes behind a shorter text were the reset
    // byte, which has emitted already and left the idle state)
    const bool tail = live && ((p.acc >> (AUTO == 5 ? (q4 & 31u) : (q4 >> 2))) & 1u) != 0;
    const int wshift = NW - 8 * ci;   // (wave uniform) register word i holds the text's word i - wshift
    // per text: matches (cnt) and event words that hold a match end (nz: the ITEMS of the expansion below)
    int cnt = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const uint32_t em = bm[i] & 0xAAAAAAAAu;
      cnt += __builtin_popcount(em);
      nz += em != 0u;
    }
    // one scan for both: matches in bits 0..15 (at most 64 x 513 per wavefront), items in bits 16.. (at most 4096)
    const uint32_t mine = (uint32_t)(cnt + (tail ? 1 : 0)) | ((uint32_t)nz << 16);
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v_ = __shfl_up(incl, d);
      if (lane >= d) incl += v_;
    }
    const uint32_t excl = incl - mine;
    const int rel = (int)(excl & 0xFFFFu);          // matches of the wavefront's texts before mine
    const int ibase = (int)(excl >> 16);            // items of the wavefront's texts before mine
    const uint32_t last = __shfl(incl, 63);
    const int total = (int)(last & 0xFFFFu), nitems = (int)(last >> 16);
    unsigned long long* ctrl = ap->ctrl;
    fused_publish(ctrl, w, nw, (uint32_t)total, lane);
    if (trace && lane == 0) trace[4 * w + 1] = (int64_t)wall_clock64();

    // ---- expansion.  lane = text, one `while (match ends left)` per event word, runs as long as the busiest of the
    // 64 texts at every word: 13 % of the lanes' steps do work on the headline mix.  So the words that hold a
    // match end are first compacted into LDS as ITEMS {event word, start carried in | word index | index of its first
    // match within the wavefront}, text by text (a branch-free pass), and then expanded with lane = item: every lane
    // has work, and neighbouring lanes hold neighbouring words of one text, so their 8-byte stores -- straight to the
    // spans' final place -- fall into the same few cache lines.
    int64_t base = 0;
    const int dbg = ap->debug;
    int32_t* __restrict__ spans = ap->spans;
    const int64_t span_cap = ap->span_cap;
    uint2* items = (uint2*)wave_lds[wave];
    constexpr int kItemCap = kSbWaveLds / 8;
    uint32_t sd_end = 0;   // start of the walk that is alive at the end of the text
    for (int ib = 0; ib < nitems || ib == 0; ib += kItemCap) {
      __builtin_amdgcn_wave_barrier();
      uint32_t k = (uint32_t)(ibase - ib);   // window index of my next item (may lie outside the window)
      uint32_t d = (uint32_t)rel;            // index of my next match within the wavefront
      uint32_t sd = 0;                       // start of the walk that is alive
      uint32_t wpos = (uint32_t)(-16 * wshift);   // text position of register word i (a running value: 64 per-word
      asm volatile("" : "+v"(sd), "+v"(wpos));    // constants derived from wshift would cost 128 scalar registers)
      if (!(dbg & 1))
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        uint32_t F = bm[i];
        asm volatile("" : "+v"(F), "+v"(wpos));   // (opaque: nothing of this pass is to be hoisted out of the window loop)
        const uint32_t em = F & 0xAAAAAAAAu, ns = F & 0x55555555u;
        // (words in front of the text's first hold no events)
        if (em != 0u && k < (uint32_t)kItemCap) items[k] = make_uint2(F, sd | (wpos << 6) | (d << 16));
        k += em != 0u;
        d += __builtin_popcount(em);
        if (ns) sd = wpos + ((31u - (uint32_t)__builtin_clz(ns)) >> 1);
        wpos += 16u;
      }
      sd_end = sd;
      if (ib == 0) {
        if (wave == 0) {   // hand the next round's first group to the other wavefronts
          const uint32_t nf = __builtin_amdgcn_readfirstlane((uint32_t)tk_next);
          if (lane == 0) {
            // the slot about to be overwritten carried round - 3's task: every wavefront must have taken it
            if (round >= 3u)
              for (int x = 1; x < kSbWaves; ++x)
                while (*(volatile int*)&wave_round[x] < (int)round - 3) __builtin_amdgcn_s_sleep(1);
            const uint64_t nfc = (uint64_t)nf < (uint64_t)nw ? nf : (uint32_t)nw;   // (nw fits 32 bits: checked on the host)
            *(volatile unsigned long long*)&blk_ticket[(round + 1u) & 3u] = ((unsigned long long)(round + 2u) << 32) | nfc;
          }
        }
        base = (dbg & 2) ? 0 : lookback_quiet(ctrl, w, nw, lane);
        if (trace && lane == 0) trace[4 * w + 2] = (int64_t)wall_clock64();
      }
      sb_wave_sync();
      if (base >= 0 && !(dbg & 4)) {
        const int wn = nitems - ib < kItemCap ? nitems - ib : kItemCap;
        for (int j = lane; j < wn; j += 64) {
          const uint2 it = items[j];
          const uint32_t ns = it.x & 0x55555555u;
          uint32_t emq = (it.x >> 1) & 0x55555555u;   // match end in front of byte b at bit 2b
          const uint32_t pos = ((it.y >> 10) & 63u) << 4;
          const uint32_t carried = it.y & 1023u;
          int64_t dst = base + (int64_t)(it.y >> 16);
          while (emq) {
            const int kq = __builtin_ctz(emq);
            const uint32_t nsb = ns & ((1u << kq) - 1u);   // walks begun at bytes in front of b
            const uint32_t st = nsb ? pos + ((31u - (uint32_t)__builtin_clz(nsb)) >> 1) : carried;
            if (dst < span_cap) mrx_stg_span(spans + 2 * dst, (int)st, (int)(pos + ((uint32_t)kq >> 1)));
            ++dst;
            emq &= emq - 1u;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    int64_t* __restrict__ prefix = ap->prefix;
    if (base < 0) {   // gave up waiting (never seen; the host reports the error word as well): poison the totals
      if (lane == 0) { prefix[n] = -1; *ap->total_out = -1; }
    } else if (!(dbg & 8)) {
      // the match that runs to the end of the text: the last of its text
      if (tail && base + rel + cnt < span_cap) mrx_stg_span(spans + 2 * (base + rel + cnt), (int)sd_end, my_len);
      if (live) prefix[my_text] = base + rel;
      if (my_text == n - 1) { prefix[n] = base + (int64_t)(incl & 0xFFFFu); *ap->total_out = base + (int64_t)(incl & 0xFFFFu); }
    }
    if (trace && lane == 0) trace[4 * w + 3] = (int64_t)wall_clock64();
  }
}

std::atomic<int> g_sb_mode{-1};
std::atomic<int64_t*> g_sb_trace{nullptr};   // -1: read MRX_STREAM_BITS once; 0 off, 1 on (default)

}  // namespace

int stream_bits_mode() {
  int m = g_sb_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("MRX_STREAM_BITS");
    m = (e && e[0] == '1') ? 1 : 0;
    g_sb_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}
void stream_bits_set_mode(int on) { g_sb_mode.store(on ? 1 : 0, std::memory_order_relaxed); }
void stream_bits_set_trace(int64_t* d_trace) { g_sb_trace.store(d_trace, std::memory_order_relaxed); }

bool stream_bits_eligible(const DevPlan& p, const uint8_t* data, int64_t stride, int64_t max_len, int64_t n) {
  if (!stream_bits_mode()) return false;
  if (!(p.flags & PF_STREAMABLE) || p.st_kind != 1) return false;   // the register-resident automaton forms only
  if (p.st_fixed_len > 0) return false;   // (exact-literal automata derive the start from the end: the other form has that)
  if (max_len > 1024 || max_len < 0 || n <= 0) return false;
  if ((n + 63) / 64 >= (int64_t(1) << 31)) return false;           // task numbers travel as 32 bits
  if (stride % 16 != 0 || ((uintptr_t)data % 16) != 0 || stride * 64 >= (int64_t(1) << 31)) return false;
  return true;
}

size_t stream_bits_ctrl_words(int64_t n) {
  const int64_t nw = (n + 63) / 64;
  return (size_t)((2 + nw + 2 * ((nw + 63) / 64) + 1) & ~int64_t(1));   // a 16-byte multiple
}
size_t stream_bits_args_bytes() { return sizeof(SbArgs); }

// workgroups that are resident at once (the occupancy the kernel was built for), queried once per device and variant
static int sb_grid_cap(int dev, int variant, const void* fn) {
  static std::atomic<int> cap[64][6];
  if (dev < 0 || dev >= 64) dev = 0;
  int c = cap[dev][variant].load(std::memory_order_relaxed);
  if (c <= 0) {
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * kSbWaves, 0) != hipSuccess || per_cu <= 0) per_cu = 3;
    c = cus * per_cu;
    if (getenv("MRX_SB_VERBOSE")) fprintf(stderr, "k_stream_bits variant %d: %d CUs x %d workgroups\n", variant, cus, per_cu);
    cap[dev][variant].store(c, std::memory_order_relaxed);
  }
  if (const char* e = getenv("MRX_SB_BPC")) { const int v = atoi(e); if (v > 0) c = (c / 3) * v; }   // measurement (3 = built-for occupancy)
  return c;
}

int stream_bits_init(int64_t n, int64_t max_len, int64_t* d_prefix, int32_t* d_spans, int64_t span_cap, int64_t* d_total, void* d_ctrl,
                     void* d_args, void* stream) {
  SbArgs a;
  a.ctrl = (unsigned long long*)d_ctrl; a.prefix = d_prefix; a.spans = d_spans; a.span_cap = span_cap; a.total_out = d_total;
  a.debug = 0;
  a.trace = g_sb_trace.load(std::memory_order_relaxed);
  if (const char* e = getenv("MRX_SB_DEBUG")) a.debug = atoi(e);
  a.phase_mode = 1;
  if (const char* e = getenv("MRX_SB_PHASE")) a.phase_mode = atoi(e);
  // a third of a task: 64 texts x max_len bytes per wavefront, ~3072 wavefronts sharing ~6 TB/s; 3.9 us per sleep period
  {
    const double task_us = 64.0 * (double)(max_len > 0 ? max_len : 1) * 3072.0 / 6.0e6;
    int ps = (int)(task_us / 3.0 / 3.9 + 0.5);
    if (const char* e = getenv("MRX_SB_PHASE_SLEEPS")) ps = atoi(e);
    a.phase_sleeps = ps < 0 ? 0 : ps > 64 ? 64 : ps;
  }
  const size_t words = stream_bits_ctrl_words(n);
  hipLaunchKernelGGL(k_sb_init, dim3((unsigned)((words + 256 * 8 - 1) / (256 * 8))), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long*)d_ctrl, (int64_t)words, (SbArgs*)d_args, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return internal_fail(MRX_E_NO_DEVICE, std::string("k_sb_init: ") + hipGetErrorString(e));
  return MRX_OK;
}

int stream_bits_scan(const DevPlan& p, const uint8_t* d_blob, const uint8_t* data, int64_t stride, const int32_t* lens,
                     int32_t len, int64_t max_len, int64_t n, const void* d_args, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int64_t nw = (n + 63) / 64;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool code = p.off_stcol32 >= 0;
  SbPlan sp;
  sp.off_col = code ? p.off_stcol32 : p.off_stcol;
  sp.reset_byte = p.st_reset_byte;
  sp.acc = code ? p.st_acc32 : p.st_accept_mask;
  sp.fixed_len = p.st_fixed_len;
#define SB_GO(AUTO, NWORDS, VARIANT)                                                                          \
  do {                                                                                                        \
    int64_t g = (nw + kSbWaves - 1) / kSbWaves;                                                               \
    const int cap = sb_grid_cap(dev, VARIANT, (const void*)k_stream_bits<AUTO, NWORDS>);                      \
    if (g > cap) g = cap;                                                                                     \
    hipLaunchKernelGGL((k_stream_bits<AUTO, NWORDS>), dim3((unsigned)g), dim3(64 * kSbWaves), 0, s, sp, d_blob, data, (uint32_t)stride, \
                       lens, len, n, (const SbArgs*)d_args);                                                  \
  } while (0)
  if (max_len <= 256) { if (code) SB_GO(5, 16, 0); else SB_GO(1, 16, 1); }
  else if (max_len <= 512) { if (code) SB_GO(5, 32, 2); else SB_GO(1, 32, 3); }
  else { if (code) SB_GO(5, 64, 4); else SB_GO(1, 64, 5); }
#undef SB_GO
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return internal_fail(MRX_E_NO_DEVICE, std::string("k_stream_bits: ") + hipGetErrorString(e));
  return MRX_OK;
}

}  // namespace mrx
