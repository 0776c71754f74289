// Ablation microbenchmark for the streaming scan loop (not part of the product).
// Builds variants of the k_stream_findall inner loop with pieces switched off to
// find what bounds it.  hipcc -O3 --offload-arch=gfx950 tools/stream_ablate.hip -o /tmp/ablate
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kChunk = 64, kRowPitch = 80, kTileBytes = 64 * kRowPitch, kWaves = 4;
struct EvRec { uint32_t F; int32_t start, pos_base; uint32_t meta; };

// VAR bits: 1 = no column lookup (cv from the byte itself), 2 = no record store,
// 4 = no global loads after the first chunk (reuse registers), 8 = no LDS tile (walk v0 directly),
// 16 = no event handling at all, 32 = old per-lane scattered record rows
// LB = waves per SIMD requested through __launch_bounds__
template <int VAR, int LB>
__global__ __launch_bounds__(64 * kWaves, LB) void k(const uint16_t* __restrict__ cols,
                                                   const uint8_t* __restrict__ data, int64_t stride,
                                                   int32_t len, int64_t n, int32_t* __restrict__ counts,
                                                   EvRec* __restrict__ recs, int64_t rec_row) {
  __shared__ __align__(16) uint8_t tiles[kWaves][kTileBytes];
  __shared__ __align__(16) uint16_t col_lds[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) col_lds[i] = cols[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, seg = lane & 3;
  uint8_t* tile = tiles[wave];
  const int64_t nw = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * kWaves + wave; w < nw; w += (int64_t)gridDim.x * kWaves) {
    const int64_t base_text = w << 6, my_text = base_text + lane;
    const int64_t t0 = base_text + (lane >> 2);
    const uint8_t* row0 = data + t0 * stride;
    const uint8_t* row1 = data + (t0 + 16) * stride;
    const uint8_t* row2 = data + (t0 + 32) * stride;
    const uint8_t* row3 = data + (t0 + 48) * stride;
    uint32_t q4 = 0; int start = 0, cnt = 0, nrec = 0, wrec = 0;
    EvRec* myrec = recs + my_text * rec_row;
    EvRec* wave_recs = recs + base_text * rec_row;
    uint4 v0, v1, v2, v3;
#define LOADC(CB) do { int64_t b_ = (CB) + seg * 16; v0 = *(const uint4*)(row0 + b_); v1 = *(const uint4*)(row1 + b_); \
                       v2 = *(const uint4*)(row2 + b_); v3 = *(const uint4*)(row3 + b_); } while (0)
    LOADC(0);
    uint8_t* wr = tile + (lane >> 2) * kRowPitch + seg * 16;
    for (int cbase = 0; cbase < len; cbase += kChunk) {
      if (!(VAR & 8)) {
        *(uint4*)(wr) = v0; *(uint4*)(wr + 16 * kRowPitch) = v1;
        *(uint4*)(wr + 32 * kRowPitch) = v2; *(uint4*)(wr + 48 * kRowPitch) = v3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (!(VAR & 4)) { if (cbase + kChunk < len) LOADC(cbase + kChunk); }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint4 wv;
        if (VAR & 8) { wv = g == 0 ? v0 : g == 1 ? v1 : g == 2 ? v2 : v3; wv.x += cbase; }
        else wv = *(const uint4*)(tile + lane * kRowPitch + g * 16);
        const uint32_t words[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t cv[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
          const uint32_t b = (words[k2 >> 2] >> ((k2 & 3) * 8)) & 0xFFu;
          cv[k2] = (VAR & 1) ? (b * 0x0101u) : col_lds[b];
        }
        uint32_t F = 0;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
          const uint32_t e = cv[k2] >> q4;
          q4 = e & 0xCu;
          F = __builtin_amdgcn_alignbit(e, F, 2);
        }
        if (VAR & 16) { cnt += F; continue; }
        const uint32_t em = F & 0xAAAAAAAAu, ns = F & 0x55555555u;
        const int gbase = cbase + g * 16;
        if (VAR & 32) {
          if (em) {
            if (!(VAR & 2)) { EvRec r; r.F = F; r.start = start; r.pos_base = gbase; r.meta = cnt; myrec[nrec] = r; }
            ++nrec;
          }
        } else {
          const uint64_t has = __ballot(em != 0);
          if (has) {
            if (em && !(VAR & 2)) {
              const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0));
              EvRec r; r.F = F; r.start = start; r.pos_base = gbase; r.meta = ((uint32_t)lane << 26) | cnt;
              wave_recs[wrec + rank] = r;
            }
            wrec += __builtin_popcountll(has);
          }
        }
        cnt += __builtin_popcount(em);
        if (ns) start = gbase + ((31 - __builtin_clz(ns)) >> 1);
      }
      __builtin_amdgcn_wave_barrier();
    }
    counts[my_text] = cnt + nrec + start + wrec;
  }
}

int main() {
  const int64_t n = 1 << 20; const int len = 1024; const int64_t stride = len;
  std::vector<uint8_t> h((size_t)n * len);
  uint32_t x = 12345;
  for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; uint32_t r = x >> 24;
    h[i] = r < 150 ? 'a' + r % 26 : r < 200 ? '0' + r % 10 : ' '; }
  // make the first 40% of texts 'full' ([a-z]{k}[0-9]{len-k}) and 10% adversarial like the bench mix
  for (int64_t t = 0; t < n; ++t) { const int kind = (int)(t % 10); uint8_t* row = &h[(size_t)t * len];
    if (kind < 4) { const int ksplit = 1 + (int)((t * 2654435761u) % (len - 1)); for (int j = 0; j < len; ++j) row[j] = j < ksplit ? 'a' + (j * 7 + t) % 26 : '0' + (j + t) % 10; }
    else if (kind == 4) { for (int j = 0; j < len; ++j) row[j] = 'a' + (j * 11 + t) % 26; row[len - 1] = '!'; }
    else if (kind < 7) { x = (uint32_t)t * 747796405u + 1; for (int j = 0; j < len; ++j) { x = x * 1664525u + 1013904223u; row[j] = 32 + (x >> 24) % 95; } } }
  // [a-z]+\d+ search automaton columns: states 0 idle, 1 letters, 2 digits(acc)
  std::vector<uint16_t> cols(256);
  for (int c = 0; c < 256; ++c) {
    const bool L = c >= 'a' && c <= 'z', D = c >= '0' && c <= '9';
    auto ent = [&](int q) { int t, em = 0, nsf = 0;
      if (q == 0) { t = L ? 1 : 0; nsf = L; }
      else if (q == 1) { t = L ? 1 : D ? 2 : 0; }
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
  const int grid = 2048;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
#define RUN2(V, LB, NAME) do { for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<V, LB>), dim3(grid), dim3(256), 0, 0, dc, d, stride, len, n, dcount, drec, rec_row); \
    CK(hipEventRecord(a)); for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k<V, LB>), dim3(grid), dim3(256), 0, 0, dc, d, stride, len, n, dcount, drec, rec_row); \
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10; \
    printf("%-34s %.4f ms  %.0f GB/s\n", NAME, ms, (double)n * len / ms / 1e6); } while (0)
#define RUN(V, NAME) RUN2(V, 1, NAME)
  RUN2(0, 1, "full (dense records) lb1");
  RUN2(0, 5, "full lb5");
  RUN2(0, 6, "full lb6");
  RUN2(0, 7, "full lb7");
  RUN2(0, 8, "full lb8");
  RUN(32, "old scattered record rows");
  RUN(1, "no col lookup");
  RUN(2, "no record store");
  RUN(3, "no col lookup, no store");
  RUN(4, "no global loads (after first)");
  RUN(6, "no loads, no store");
  RUN(7, "no loads, no store, no lookup");
  RUN(16, "no event handling");
  RUN(8 | 4, "no tile, no loads");
  RUN(8 | 4 | 16 | 1, "chain only");
  return 0;
}
