// Ablation 2: chunk size (64 vs 128 bytes per text and stage), launch bounds, prefetch depth.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kWaves = 4;
struct EvRec { uint32_t F; int32_t start, pos_base; uint32_t meta; };

template <int CH, int LB, int NOSTORE>
__global__ __launch_bounds__(64 * kWaves, LB) void k(const uint16_t* __restrict__ cols,
                                                   const uint8_t* __restrict__ data, int64_t stride,
                                                   int32_t len, int64_t n, int32_t* __restrict__ counts,
                                                   EvRec* __restrict__ recs, int64_t rec_row) {
  constexpr int PITCH = CH + 16, LPR = CH / 16 /*lanes per row*/, RPI = 64 / LPR /*rows per instr*/, NL = 64 / RPI;
  __shared__ __align__(16) uint8_t tiles[kWaves][64 * PITCH];
  __shared__ __align__(16) uint16_t col_lds[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) col_lds[i] = cols[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, seg = lane % LPR, rsub = lane / LPR;
  uint8_t* tile = tiles[wave];
  const int64_t nw = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * kWaves + wave; w < nw; w += (int64_t)gridDim.x * kWaves) {
    const int64_t base_text = w << 6, my_text = base_text + lane;
    const uint8_t* rowb = data + (base_text + rsub) * stride + seg * 16;
    uint32_t q4 = 0; int start = 0, cnt = 0, wrec = 0;
    EvRec* wave_recs = recs + base_text * rec_row;
    uint4 v[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) v[j] = *(const uint4*)(rowb + (int64_t)j * RPI * stride);
    uint8_t* wr = tile + rsub * PITCH + seg * 16;
    for (int cbase = 0; cbase < len; cbase += CH) {
#pragma unroll
      for (int j = 0; j < NL; ++j) *(uint4*)(wr + j * RPI * PITCH) = v[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (cbase + CH < len) {
#pragma unroll
        for (int j = 0; j < NL; ++j) v[j] = *(const uint4*)(rowb + (int64_t)j * RPI * stride + cbase + CH);
      }
#pragma unroll
      for (int g = 0; g < CH / 16; ++g) {
        const uint4 wv = *(const uint4*)(tile + lane * PITCH + g * 16);
        const uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t cv[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) cv[k2] = col_lds[(words[k2 >> 2] >> ((k2 & 3) * 8)) & 0xFFu];
        uint32_t F = 0;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) { const uint32_t e = cv[k2] >> q4; q4 = e & 0xCu; F = __builtin_amdgcn_alignbit(e, F, 2); }
        const uint32_t em = F & 0xAAAAAAAAu, ns = F & 0x55555555u;
        const int gbase = cbase + g * 16;
        const uint64_t has = __ballot(em != 0);
        if (has) {
          if (em && !NOSTORE) {
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
            EvRec r; r.F = F; r.start = start; r.pos_base = gbase; r.meta = ((uint32_t)lane << 26) | cnt;
            wave_recs[wrec + rank] = r;
          }
          wrec += __builtin_popcountll(has);
        }
        cnt += __builtin_popcount(em);
        if (ns) start = gbase + ((31 - __builtin_clz(ns)) >> 1);
      }
      __builtin_amdgcn_wave_barrier();
    }
    counts[my_text] = cnt + start + wrec;
  }
}

int main() {
  const int64_t n = 1 << 20; const int len = 1024; const int64_t stride = len;
  std::vector<uint8_t> h((size_t)n * len);
  uint32_t x = 12345;
  for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; uint32_t r = x >> 24;
    h[i] = r < 150 ? 'a' + r % 26 : r < 200 ? '0' + r % 10 : ' '; }
  for (int64_t t = 0; t < n; ++t) { const int kind = (int)(t % 10); uint8_t* row = &h[(size_t)t * len];
    if (kind < 4) { const int ksplit = 1 + (int)((t * 2654435761u) % (len - 1)); for (int j = 0; j < len; ++j) row[j] = j < ksplit ? 'a' + (j * 7 + t) % 26 : '0' + (j + t) % 10; }
    else if (kind == 4) { for (int j = 0; j < len; ++j) row[j] = 'a' + (j * 11 + t) % 26; row[len - 1] = '!'; }
    else if (kind < 7) { x = (uint32_t)t * 747796405u + 1; for (int j = 0; j < len; ++j) { x = x * 1664525u + 1013904223u; row[j] = 32 + (x >> 24) % 95; } } }
  std::vector<uint16_t> cols(256);
  for (int c = 0; c < 256; ++c) {
    const bool L = c >= 'a' && c <= 'z', D = c >= '0' && c <= '9';
    auto ent = [&](int q) { int t, em = 0, nsf = 0;
      if (q == 0) { t = L ? 1 : 0; nsf = L; } else if (q == 1) { t = L ? 1 : D ? 2 : 0; }
      else { if (D) t = 2; else { em = 1; t = L ? 1 : 0; nsf = L; } }
      return (t << 2) | (em << 1) | nsf; };
    cols[c] = (uint16_t)(ent(0) | (ent(1) << 4) | (ent(2) << 8));
  }
  uint8_t* d; uint16_t* dc; int32_t* dcount; EvRec* drec;
  const int64_t rec_row = len / 16 + 2;
  CK(hipMalloc(&d, h.size() + 64)); CK(hipMalloc(&dc, 512)); CK(hipMalloc(&dcount, n * 4));
  CK(hipMalloc(&drec, sizeof(EvRec) * rec_row * n));
  CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dc, cols.data(), 512, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
#define RUN(CH, LB, NS, GRID, NAME) do { for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<CH, LB, NS>), dim3(GRID), dim3(256), 0, 0, dc, d, stride, len, n, dcount, drec, rec_row); \
    CK(hipEventRecord(a)); for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k<CH, LB, NS>), dim3(GRID), dim3(256), 0, 0, dc, d, stride, len, n, dcount, drec, rec_row); \
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10; \
    printf("%-40s %.4f ms  %.0f GB/s\n", NAME, ms, (double)n * len / ms / 1e6); } while (0)
  RUN(64, 1, 0, 2048, "chunk 64 lb1 grid2048");
  RUN(64, 8, 0, 2048, "chunk 64 lb8 grid2048");
  RUN(64, 8, 0, 4096, "chunk 64 lb8 grid4096");
  RUN(64, 8, 0, 1024, "chunk 64 lb8 grid1024");
  RUN(64, 8, 1, 2048, "chunk 64 lb8 nostore");
  RUN(128, 1, 0, 2048, "chunk 128 lb1 grid2048");
  RUN(128, 4, 0, 2048, "chunk 128 lb4 grid2048");
  RUN(128, 4, 0, 1024, "chunk 128 lb4 grid1024");
  RUN(128, 4, 0, 4096, "chunk 128 lb4 grid4096");
  RUN(128, 4, 1, 2048, "chunk 128 lb4 nostore");
  RUN(32, 8, 0, 2048, "chunk 32 lb8 grid2048");
  return 0;
}
