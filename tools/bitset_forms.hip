// A/B for the bitset-NFA union pass (the structure of k_bscan in mrx_kernels.hip): the SAME program
// (\d{10}: 10 CLASS positions + MATCH, PikeVM numbering), the SAME text tile in LDS (64 texts x 128 B
// per wavefront and window, coalesced 16-byte loads), two thread mappings:
//   A  one lane per TEXT   -- a lane keeps its text's live-position set in a register and does one
//                             mask read + follow-table reads per byte; 64 texts advance per step
//   B  one lane per STATE  -- the wavefront walks ONE text; lane p is position p; a step is
//                             fire = active & consumes[byte] (scalar), next_p = (pred[p] & fire) != 0
//                             or p in the start set on a candidate byte, active = ballot(next_p)
// Both count, per text, the positions where the union automaton holds MATCH; counts must agree.
// build: hipcc -O3 --offload-arch=gfx950 tools/bitset_forms.hip -o tools/bitset_forms.bin
// run:   tools/bitset_forms.bin [texts=262144] [len=1024]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int CH = 128, PITCH = CH + 16, WAVES = 4, NPOS = 11;

__global__ __launch_bounds__(64 * WAVES) void k_fill(uint8_t* d, size_t nbytes, uint32_t seed) {
  // "Call DDDDDDDDDD or DDDDDDDDD today. " style: runs of 9..11 digits between short words
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbytes; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t blk = (uint32_t)(i / 18), off = (uint32_t)(i % 18);
    uint32_t h = (blk * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const uint32_t run = 9 + h % 3;   // digits in this 18-byte block
    uint32_t r = ((uint32_t)i * 2654435761u) ^ (seed * 31u); r ^= r >> 16; r *= 2246822519u; r ^= r >> 13;
    d[i] = off < run ? (uint8_t)('0' + r % 10) : (uint8_t)("Call or "[off % 8]);
  }
}

template <int FORM>
__global__ __launch_bounds__(64 * WAVES) void k_union(const uint8_t* __restrict__ data, int64_t n, int len, int* __restrict__ counts) {
  __shared__ __align__(16) uint8_t tiles[WAVES][64 * PITCH];
  __shared__ uint32_t mask_t[256];       // positions that consume the byte
  __shared__ uint32_t fol8[2][256];      // follow sets by 8-bit chunk of the firing set
  __shared__ uint32_t pred_t[32];        // B: positions q with p in follow[q]
  const uint32_t start = 1u, match = 1u << 10;
  for (int b = threadIdx.x; b < 256; b += blockDim.x) {
    mask_t[b] = (b >= '0' && b <= '9') ? 0x3FFu : 0u;
    for (int j = 0; j < 2; ++j) {
      uint32_t u = 0;
      for (int k = 0; k < 8; ++k) if ((b >> k) & 1) { const int q = 8 * j + k; if (q < 10) u |= 1u << (q + 1); }
      fol8[j][b] = u;
    }
    if (b < 32) pred_t[b] = (b >= 1 && b <= 10) ? 1u << (b - 1) : 0u;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* tile = tiles[wave];
  const int seg = lane & 7, rsub = lane >> 3;
  const int64_t nw = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * WAVES + wave; w < nw; w += (int64_t)gridDim.x * WAVES) {
    const uint8_t* wbase = data + (w << 6) * (int64_t)len;
    uint32_t U = 0;           // A: my text's set
    int cnt = 0;              // A: my text's count; B: lane 0 .. 63 hold text r's count in lane r
    uint64_t actB[1];         // (B keeps one active mask per text across windows: in LDS-free form, 64 scalars)
    (void)actB;
    uint32_t act_save = 0;    // B: lane r keeps text r's active set between windows
    for (int wb = 0; wb < len; wb += CH) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 8 * j + rsub;
        *(uint4*)(tile + r * PITCH + seg * 16) = *(const uint4*)(wbase + (int64_t)r * len + wb + seg * 16);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (FORM == 0) {
        const uint8_t* row = tile + lane * PITCH;
        for (int g = 0; g < CH / 16; ++g) {
          const uint4 wv = *(const uint4*)(row + g * 16);
          const uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const uint32_t b = (words[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            const uint32_t m = mask_t[b];
            const uint32_t x = (U & m) | (start & m);
            U = fol8[0][x & 0xFFu] | fol8[1][(x >> 8) & 0xFFu];
            cnt += (U & match) ? 1 : 0;
          }
        }
      } else {
        for (int r = 0; r < 64; ++r) {   // the wavefront walks text r of the tile
          uint32_t active = __builtin_amdgcn_readlane(act_save, r);
          const uint8_t* row = tile + r * PITCH;
          const uint32_t pred = lane < 32 ? pred_t[lane] : 0u;
          int c = 0;
          for (int k = 0; k < CH; ++k) {
            const uint32_t b = row[k];                      // same address in every lane: broadcast
            const uint32_t m = mask_t[b];
            const uint32_t fire = (active & m) | (start & m);
            const bool nxt = (pred & fire) != 0u;
            active = (uint32_t)__ballot(nxt);
            c += (active & match) ? 1 : 0;
          }
          if (lane == r) { act_save = active; cnt += c; }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    const int64_t i = (w << 6) + lane;
    if (i < n) counts[i] = cnt;
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 262144;
  const int len = argc > 2 ? atoi(argv[2]) : 1024;
  uint8_t* d; int *c0, *c1;
  CK(hipMalloc(&d, (size_t)n * len)); CK(hipMalloc(&c0, n * 4)); CK(hipMalloc(&c1, n * 4));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d, (size_t)n * len, 12345u);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms[2] = {0, 0};
  for (int form = 0; form < 2; ++form) {
    const int reps = form == 0 ? 10 : 2;
    for (int it = 0; it < reps + 1; ++it) {
      if (it == 1) CK(hipEventRecord(a, 0));
      if (form == 0) hipLaunchKernelGGL(k_union<0>, dim3(1024), dim3(64 * WAVES), 0, 0, d, n, len, c0);
      else hipLaunchKernelGGL(k_union<1>, dim3(1024), dim3(64 * WAVES), 0, 0, d, n, len, c1);
    }
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms[form], a, b)); ms[form] /= reps;
  }
  std::vector<int> h0(n), h1(n);
  CK(hipMemcpy(h0.data(), c0, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), c1, n * 4, hipMemcpyDeviceToHost));
  long long tot = 0, bad = 0;
  for (int64_t i = 0; i < n; ++i) { tot += h0[i]; bad += h0[i] != h1[i]; }
  const double gb = (double)n * len / 1e9;
  printf("{\"texts\": %lld, \"len\": %d, \"match_ends\": %lld, \"mismatching_texts\": %lld, "
         "\"lane_per_text_ms\": %.3f, \"lane_per_text_GBps\": %.1f, \"lane_per_state_ms\": %.3f, \"lane_per_state_GBps\": %.1f}\n",
         (long long)n, len, tot, bad, ms[0], gb / ms[0] * 1e3, ms[1], gb / ms[1] * 1e3);
  return bad ? 1 : 0;
}
