// Device helpers shared by the streaming kernels (mrx_kernels.hip, mrx_stream_bits.hip): non-temporal
// accesses and the decoupled look-back over per-task span counts.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mrx {

// The text stream is read exactly once: non-temporal loads keep it from displacing the event
// records (written here, read back by k_decode) in L2 / Infinity Cache.  -DMRX_NT_LOADS=0 to compare.
#ifndef MRX_NT_LOADS
#define MRX_NT_LOADS 1
#endif
typedef unsigned int mrx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 mrx_ldg(const uint4* p) {
#if MRX_NT_LOADS
  const mrx_u32x4 v = __builtin_nontemporal_load((const mrx_u32x4*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
#define MRX_LDG(P) mrx_ldg(P)
typedef int mrx_i32x2 __attribute__((ext_vector_type(2)));
// result spans are written once and not read again by this library
__device__ __forceinline__ void mrx_stg_span(int32_t* p, int a, int b) {
#if MRX_NT_LOADS
  mrx_i32x2 v; v.x = a; v.y = b;
  __builtin_nontemporal_store(v, (mrx_i32x2*)p);
#else
  *(int2*)p = make_int2(a, b);
#endif
}



constexpr unsigned long long kDescValid = 1ull << 62, kDescVal = (1ull << 62) - 1ull;
// group word: bits 0..39 = spans of the group's tasks that have reported, bits 40..47 = how many have
constexpr int kGroupCountShift = 40;
constexpr unsigned long long kGroupSumMask = (1ull << kGroupCountShift) - 1ull;
constexpr uint32_t kLookbackSpinLimit = 1u << 22;   // insurance only: a predecessor is always a running wavefront

// The end of a task's scan: its span count goes out at once, as the task's own descriptor and added
// to its group's word (64 consecutive tasks form a group; fire-and-forget atomic, one word per group).
// Every word is one 8-byte relaxed agent-scope access whose value IS the flag, so no fence is involved.
__device__ __forceinline__ void fused_publish(unsigned long long* ctrl, int64_t w, int64_t nw, uint32_t total, int lane) {
  if (lane == 0) {
    unsigned long long* desc = ctrl + 2;
    __hip_atomic_store(desc + w, kDescValid | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_add(desc + nw + (w >> 6), (unsigned long long)total + (1ull << kGroupCountShift),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Spans of all tasks before task w = (tasks of my group before me, one window of task descriptors)
// + (whole groups before mine, windows of 64 group words, cut short at the nearest group whose first
// task has already published the running total at the group's start).  Nothing here waits for another
// wavefront's look-back -- only for scans, and this runs one task late, so as a rule nothing waits at all.
template <bool QUIET = false>
__device__ __forceinline__ int64_t fused_lookback(unsigned long long* ctrl, int64_t w, int64_t nw, int lane) {
  unsigned long long* desc = ctrl + 2;
  unsigned long long* gsum = desc + nw;
  unsigned long long* ginc = gsum + ((nw + 63) >> 6);
  const int64_t G = w >> 6;
  const int r = (int)(w & 63);
  int64_t base = 0;
  bool level1 = true;
  uint32_t spins = 0;
  for (int64_t g0 = G - 1;;) {
    unsigned long long d1 = kDescValid;
    if (level1 && lane < r) d1 = __hip_atomic_load(desc + (w - 1 - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t g = g0 - lane;
    unsigned long long gs = 64ull << kGroupCountShift, gi = kDescValid;   // in front of group 0: nothing, total 0
    if (g >= 0) {
      gs = __hip_atomic_load(gsum + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      gi = __hip_atomic_load(ginc + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint64_t inc_m = __ballot((gi & kDescValid) != 0ull);
    const int F = inc_m ? __builtin_ctzll(inc_m) : 64;   // nearest group whose starting total is known
    const bool need = lane <= F;
    const bool ok = (d1 & kDescValid) != 0ull && (!need || (gs >> kGroupCountShift) == 64ull);
    if (!__all(ok)) {   // a scan in front of me has not reported yet
      if (++spins > kLookbackSpinLimit) {   // gave up (never seen): no spans are stored, the totals are poisoned
        if (lane == 0) __hip_atomic_fetch_or(ctrl + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
      }
      if (QUIET) {
        // Wait on ONE of the missing words with ONE lane, then read the window again.  (Every lane re-reading its
        // three words until all are there is 192 memory-side requests per wavefront and try: a few thousand
        // wavefronts waiting for the stragglers of their round took the fabric away from the scans they were
        // waiting for -- measured 0.62 ms against 0.32 ms without the look-back, profiles/r04_stream_bits.md.)
        const uint64_t m1 = __ballot((d1 & kDescValid) == 0ull);
        const uint64_t m2 = __ballot(need && (gs >> kGroupCountShift) != 64ull);
        const unsigned long long* wp = m1 ? desc + (w - 1 - __builtin_ctzll(m1)) : gsum + (g0 - (m2 ? __builtin_ctzll(m2) : 0));
        const bool is_desc = m1 != 0ull;
        if (lane == 0) {
          for (uint32_t k = 0; k < (1u << 16); ++k) {
            const unsigned long long v = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (is_desc ? (v & kDescValid) != 0ull : (v >> kGroupCountShift) == 64ull) break;
            __builtin_amdgcn_s_sleep(8);
          }
        }
      } else {
        __builtin_amdgcn_s_sleep(2);
      }
      continue;
    }
    int64_t v = (level1 && lane < r ? (int64_t)(d1 & kDescVal) : 0) + (need ? (int64_t)(gs & kGroupSumMask) : 0) +
                (lane == F ? (int64_t)(gi & kDescVal) : 0);
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    base += v;
    level1 = false;
    if (F < 64) break;
    g0 -= 64;
  }
  if (r == 0 && lane == 0)   // the running total at the start of my group, for the groups behind
    __hip_atomic_store(ginc + G, kDescValid | (unsigned long long)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return base;
}

// The look-back of k_stream_bits: the same words as above, read as little as possible.  A task that is not the first
// of its group waits for the running total at the group's start -- ONE word, polled by ONE lane -- which the group's
// first task publishes once its own look-back (windows of 64 group words, as above) is through; then the tasks of
// its group in front of it (one window).  Whatever is missing is polled word by word by one lane with a sleep
// between tries: a wavefront that waits costs the memory system one 8-byte load per microsecond instead of 192
// per try (a few thousand wavefronts waiting for the stragglers of their round had taken the fabric away from the
// scans they were waiting for: 0.62 ms against 0.32 ms without any look-back, profiles/r04_stream_bits.md).
__device__ __forceinline__ bool lookback_poll(const unsigned long long* wp, bool group_word, int lane) {
  bool ok = true;
  if (lane == 0) {
    ok = false;
    for (uint32_t k = 0; k < (1u << 20); ++k) {
      const unsigned long long v = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (group_word ? (v >> kGroupCountShift) == 64ull : (v & kDescValid) != 0ull) { ok = true; break; }
      __builtin_amdgcn_s_sleep(16);
    }
  }
  return __shfl((int)ok, 0) != 0;
}
__device__ __forceinline__ int64_t lookback_quiet(unsigned long long* ctrl, int64_t w, int64_t nw, int lane) {
  unsigned long long* desc = ctrl + 2;
  unsigned long long* gsum = desc + nw;
  unsigned long long* ginc = gsum + ((nw + 63) >> 6);
  const int64_t G = w >> 6;
  const int r = (int)(w & 63);
  int64_t base = 0;
  bool failed = false;
  if (r == 0) {
    // spans of all groups before mine: windows of 64 group words, cut short at the nearest known running total
    for (int64_t g0 = G - 1; g0 >= 0 && !failed;) {
      const int64_t g = g0 - lane;
      unsigned long long gs = 64ull << kGroupCountShift, gi = kDescValid;   // in front of group 0: nothing, total 0
      if (g >= 0) {
        gs = __hip_atomic_load(gsum + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gi = __hip_atomic_load(ginc + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const uint64_t inc_m = __ballot((gi & kDescValid) != 0ull);
      const int F = inc_m ? __builtin_ctzll(inc_m) : 64;
      const bool need = lane <= F;
      const uint64_t miss = __ballot(need && (gs >> kGroupCountShift) != 64ull);
      if (miss) {   // a scan of an earlier group has not reported: wait for that group's word, then look again
        if (!lookback_poll(gsum + (g0 - __builtin_ctzll(miss)), true, lane)) failed = true;
        continue;
      }
      int64_t v = (need ? (int64_t)(gs & kGroupSumMask) : 0) + (lane == F ? (int64_t)(gi & kDescVal) : 0);
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      base += v;
      if (F < 64) break;
      g0 -= 64;
    }
    if (!failed && lane == 0)
      __hip_atomic_store(ginc + G, kDescValid | (unsigned long long)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (!lookback_poll(ginc + G, false, lane)) failed = true;
    unsigned long long gi = 0ull;
    if (lane == 0) gi = __hip_atomic_load(ginc + G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = (int64_t)((unsigned long long)__shfl((long long)gi, 0) & kDescVal);
    while (!failed) {   // the tasks of my group in front of me
      unsigned long long d1 = kDescValid;
      if (lane < r) d1 = __hip_atomic_load(desc + (w - 1 - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint64_t miss = __ballot((d1 & kDescValid) == 0ull);
      if (miss) {
        if (!lookback_poll(desc + (w - 1 - __builtin_ctzll(miss)), false, lane)) failed = true;
        continue;
      }
      int64_t v = lane < r ? (int64_t)(d1 & kDescVal) : 0;
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      base += v;
      break;
    }
  }
  if (failed) {   // gave up (never seen): no spans are stored, the totals are poisoned
    if (lane == 0) __hip_atomic_fetch_or(ctrl + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return -1;
  }
  return base;
}

}  // namespace mrx
