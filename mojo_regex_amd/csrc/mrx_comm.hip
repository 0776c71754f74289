// mrx_comm.hip -- results exchange between the ranks of a sharded batch (include/mrx_comm.h), on RCCL
// directly.  The reference is single-process and has no counterpart; contract: SURVEY.md 8(e).
//
// RCCL is opened with dlopen at the first mrx_comm_* call: the matching library itself has no RCCL
// dependency, and a process that already holds librccl.so.1 (PyTorch's nccl backend) shares it.
// xGMI is point to point (7 links per GPU): the all-gatherv is one ncclBroadcast per rank inside ONE
// group, so that RCCL schedules all of them at once over the links instead of ring after ring.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only; every function is resolved with dlsym

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mrx.h"
#include "../../include/mrx_comm.h"
#include "mrx_internal.hpp"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> names;
    if (const char* e = getenv("MRX_RCCL_LIB")) names.push_back(e);
    names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
    for (const std::string& n : names)
      if ((r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.lib) { r.why = std::string("RCCL not found (librccl.so.1): ") + dlerror(); return; }
#define MRX_SYM(field, name)                                                    \
    if (!(*(void**)(&r.field) = dlsym(r.lib, name))) { r.why = std::string("RCCL lacks ") + name; r.lib = nullptr; return; }
    MRX_SYM(GetUniqueId, "ncclGetUniqueId");
    MRX_SYM(CommInitRank, "ncclCommInitRank");
    MRX_SYM(CommDestroy, "ncclCommDestroy");
    MRX_SYM(AllGather, "ncclAllGather");
    MRX_SYM(Broadcast, "ncclBroadcast");
    MRX_SYM(GroupStart, "ncclGroupStart");
    MRX_SYM(GroupEnd, "ncclGroupEnd");
    MRX_SYM(GetErrorString, "ncclGetErrorString");
#undef MRX_SYM
  });
  return r;
}

#define CC_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return mrx::internal_fail(MRX_E_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define CC_NCCL(expr)                                                                                  \
  do {                                                                                                 \
    ncclResult_t e_ = (expr);                                                                          \
    if (e_ != ncclSuccess) return mrx::internal_fail(MRX_E_NO_DEVICE, std::string(#expr) + ": " + R.GetErrorString(e_)); \
  } while (0)

constexpr int kCommBlock = 256;

// What every rank tells every other rank before anything else moves (kMeta int64 words per rank): {texts, spans,
// capacity of its global offsets buffer, capacity of its global spans buffer}.  texts == -1: this rank's own
// arguments are invalid.  Every decision that ends a call early is taken from the GATHERED words, so all ranks take
// it together -- a rank that returned on its own would leave the others blocked in the next collective.
constexpr int kMeta = 4;
__global__ void k_comm_meta(const int64_t* __restrict__ prefix, int64_t n_local, int64_t cap_a, int64_t cap_b,
                            int64_t* __restrict__ meta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    meta[0] = n_local;
    meta[1] = (prefix && n_local >= 0) ? prefix[n_local] : 0;
    meta[2] = cap_a;
    meta[3] = cap_b;
  }
}

// out[i] = prefix[i + 1] + base: the inclusive ends of my texts in the global span numbering
// (base = spans of the ranks before me; meta_all = {n, total} of every rank, on the device)
__global__ __launch_bounds__(kCommBlock) void k_comm_shift(const int64_t* __restrict__ prefix, int64_t n_local,
                                                           const int64_t* __restrict__ meta_all, int mstride, int rank,
                                                           int64_t* __restrict__ out, int64_t pad_to) {
  int64_t base = 0;
  for (int r = 0; r < rank; ++r) base += meta_all[mstride * r + 1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pad_to; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = i < n_local ? prefix[i + 1] + base : 0;
}

// PADDED form: stage_prefix[G][P] (shifted inclusive ends) and stage_spans[G][cap] -> the global CSR
__global__ __launch_bounds__(kCommBlock) void k_comm_compact(const int64_t* __restrict__ meta_all, int mstride, int nranks,
                                                             const int64_t* __restrict__ stage_prefix, int64_t P,
                                                             const int2* __restrict__ stage_spans, int64_t cap,
                                                             int64_t* __restrict__ gprefix, int64_t gprefix_cap,
                                                             int2* __restrict__ gspans, int64_t gspans_cap,
                                                             int32_t* __restrict__ status) {
  const int r = blockIdx.y;
  int64_t tbase = 0, sbase = 0, N = 0, T = 0;
  bool over = false, invalid = false;
  for (int q = 0; q < nranks; ++q) {
    const int64_t n_q = meta_all[mstride * q], t_q = meta_all[mstride * q + 1];
    if (n_q < 0) { invalid = true; continue; }
    if (q < r) { tbase += n_q; sbase += t_q; }
    N += n_q; T += t_q;
    over = over || t_q > cap || n_q > P;
  }
  over = over || N + 1 > gprefix_cap || T > gspans_cap;
  // (the capacities of EVERY rank, so that all ranks write or none does)
  if (mstride >= 4)
    for (int q = 0; q < nranks; ++q) over = over || N + 1 > meta_all[mstride * q + 2] || T > meta_all[mstride * q + 3];
  if (over || invalid) {   // nothing is written: some rank's arguments or capacities do not hold this result
    if (status && r == 0 && blockIdx.x == 0 && threadIdx.x == 0) *status = invalid ? MRX_E_ARGUMENT : MRX_E_CAPACITY;
    return;
  }
  const int64_t n_r = meta_all[mstride * r], t_r = meta_all[mstride * r + 1];
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < n_r; i += step) gprefix[tbase + 1 + i] = stage_prefix[(int64_t)r * P + i];
  for (int64_t k = tid; k < t_r; k += step) gspans[sbase + k] = stage_spans[(int64_t)r * cap + k];
  if (r == 0 && tid == 0) { gprefix[0] = 0; if (status) *status = MRX_OK; }
}

}  // namespace

struct mrx_comm {
  ncclComm_t nccl = nullptr;
  int nranks = 0, rank = 0, device = 0;
  int64_t* d_meta = nullptr;      // [kMeta] mine + [kMeta * nranks] everyone's
  int64_t* h_meta = nullptr;      // pinned, [kMeta * nranks]
  uint8_t* d_stage = nullptr;     // grow-only staging (shifted prefix to send; padded form: gathered rows)
  size_t stage_bytes = 0;
  int ensure_stage(size_t bytes) {
    if (bytes <= stage_bytes) return MRX_OK;
    if (d_stage) { CC_HIP(hipDeviceSynchronize()); CC_HIP(hipFree(d_stage)); d_stage = nullptr; stage_bytes = 0; }
    CC_HIP(hipMalloc((void**)&d_stage, bytes));
    stage_bytes = bytes;
    return MRX_OK;
  }
};

extern "C" {

int mrx_comm_unique_id(uint8_t id[MRX_COMM_ID_BYTES]) {
  Rccl& R = rccl();
  if (!R.lib) return mrx::internal_fail(MRX_E_NO_DEVICE, R.why);
  if (!id) return mrx::internal_fail(MRX_E_ARGUMENT, "null argument");
  static_assert(sizeof(ncclUniqueId) == MRX_COMM_ID_BYTES, "id size");
  ncclUniqueId u;
  CC_NCCL(R.GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return MRX_OK;
}

int mrx_comm_init(const uint8_t id[MRX_COMM_ID_BYTES], int nranks, int rank, mrx_comm** out) {
  Rccl& R = rccl();
  if (!R.lib) return mrx::internal_fail(MRX_E_NO_DEVICE, R.why);
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return mrx::internal_fail(MRX_E_ARGUMENT, "bad communicator arguments");
  *out = nullptr;
  mrx_comm* c = new mrx_comm();
  c->nranks = nranks; c->rank = rank;
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; return mrx::internal_fail(MRX_E_NO_DEVICE, "no HIP device"); }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclResult_t e = R.CommInitRank(&c->nccl, nranks, u, rank);
  if (e != ncclSuccess) { delete c; return mrx::internal_fail(MRX_E_NO_DEVICE, std::string("ncclCommInitRank: ") + R.GetErrorString(e)); }
  if (hipMalloc((void**)&c->d_meta, sizeof(int64_t) * kMeta * (1 + (size_t)nranks)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_meta, sizeof(int64_t) * kMeta * (size_t)nranks) != hipSuccess) {
    mrx_comm_free(c);
    return mrx::internal_fail(MRX_E_NO_DEVICE, "communicator scratch allocation failed");
  }
  *out = c;
  return MRX_OK;
}

void mrx_comm_free(mrx_comm* c) {
  if (!c) return;
  Rccl& R = rccl();
  if (c->nccl && R.lib) (void)R.CommDestroy(c->nccl);
  if (c->d_meta) (void)hipFree(c->d_meta);
  if (c->h_meta) (void)hipHostFree(c->h_meta);
  if (c->d_stage) (void)hipFree(c->d_stage);
  delete c;
}

int mrx_comm_rank(const mrx_comm* c) { return c ? c->rank : -1; }
int mrx_comm_size(const mrx_comm* c) { return c ? c->nranks : 0; }

int mrx_allgather_fixed(mrx_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream) {
  Rccl& R = rccl();
  if (!c || !R.lib) return mrx::internal_fail(MRX_E_ARGUMENT, "no communicator");
  if (bytes_per_rank == 0) return MRX_OK;
  if (!d_send || !d_recv) return mrx::internal_fail(MRX_E_ARGUMENT, "null argument");
  CC_NCCL(R.AllGather(d_send, d_recv, bytes_per_rank, ncclChar, c->nccl, (hipStream_t)stream));
  return MRX_OK;
}

int mrx_allgatherv_rows(mrx_comm* c, const void* d_send, int64_t rows_local, size_t row_bytes,
                        void* d_out, int64_t out_cap_rows, int64_t* rows_total, void* stream) {
  Rccl& R = rccl();
  if (!c || !R.lib) return mrx::internal_fail(MRX_E_ARGUMENT, "no communicator");
  // A rank whose own arguments are bad still takes part in the size exchange (as "invalid"), so that every rank
  // learns it and all return together -- returning here would leave the others blocked in the all-gather.
  const bool bad = rows_local < 0 || row_bytes == 0 || (!d_send && rows_local > 0) || !d_out || out_cap_rows < 0;
  hipStream_t s = (hipStream_t)stream;
  const int G = c->nranks;
  int64_t* mine = c->d_meta;
  int64_t* all = c->d_meta + kMeta;
  hipLaunchKernelGGL(k_comm_meta, dim3(1), dim3(64), 0, s, (const int64_t*)nullptr, bad ? int64_t(-1) : rows_local,
                     out_cap_rows, int64_t(0), mine);
  CC_NCCL(R.AllGather(mine, all, kMeta * sizeof(int64_t), ncclChar, c->nccl, s));
  CC_HIP(hipMemcpyAsync(c->h_meta, all, sizeof(int64_t) * kMeta * G, hipMemcpyDeviceToHost, s));
  CC_HIP(hipStreamSynchronize(s));   // the one host synchronisation of the exact form
  int64_t total = 0, min_cap = INT64_MAX;
  bool any_bad = false;
  for (int r = 0; r < G; ++r) {
    if (c->h_meta[kMeta * r] < 0) { any_bad = true; continue; }
    total += c->h_meta[kMeta * r];
    if (c->h_meta[kMeta * r + 2] < min_cap) min_cap = c->h_meta[kMeta * r + 2];
  }
  if (rows_total) *rows_total = total;
  // (every rank sees the same words and takes the same branch)
  if (any_bad) return mrx::internal_fail(MRX_E_ARGUMENT, bad ? "bad arguments" : "another rank passed bad arguments");
  if (total > min_cap) return mrx::internal_fail(MRX_E_CAPACITY, "gather output too small on at least one rank");
  CC_NCCL(R.GroupStart());
  int64_t off = 0;
  ncclResult_t bad_nccl = ncclSuccess;
  for (int r = 0; r < G; ++r) {
    const int64_t rows = c->h_meta[kMeta * r];
    uint8_t* dst = (uint8_t*)d_out + (size_t)off * row_bytes;
    if (rows > 0) {
      ncclResult_t e = R.Broadcast(r == c->rank ? d_send : dst, dst, (size_t)rows * row_bytes, ncclChar, r, c->nccl, s);
      if (e != ncclSuccess) bad_nccl = e;
    }
    off += rows;
  }
  CC_NCCL(R.GroupEnd());
  if (bad_nccl != ncclSuccess) return mrx::internal_fail(MRX_E_NO_DEVICE, std::string("ncclBroadcast: ") + R.GetErrorString(bad_nccl));
  return MRX_OK;
}

size_t mrx_comm_spans_staging_bytes(const mrx_comm* c, int64_t n_global, int64_t cap_spans_per_rank) {
  if (!c || n_global < 0 || cap_spans_per_rank < 0) return 0;
  const int G = c->nranks;
  const int64_t P = (n_global + G - 1) / G + 1;
  return sizeof(int64_t) * (size_t)P * (size_t)(G + 1) + sizeof(int2) * (size_t)cap_spans_per_rank * (size_t)G + 64;
}

int mrx_comm_reserve(mrx_comm* c, size_t bytes) {
  if (!c) return mrx::internal_fail(MRX_E_ARGUMENT, "no communicator");
  return c->ensure_stage(bytes);
}

int mrx_allgatherv_spans(mrx_comm* c, const int64_t* d_prefix, int64_t n_local, const int32_t* d_spans,
                         int64_t cap_spans_per_rank, int64_t n_global,
                         int64_t* d_gprefix, int64_t gprefix_cap, int32_t* d_gspans, int64_t gspans_cap,
                         int64_t* N_total, int64_t* T_total, int32_t* d_status, void* stream) {
  Rccl& R = rccl();
  if (!c || !R.lib) return mrx::internal_fail(MRX_E_ARGUMENT, "no communicator");
  hipStream_t s = (hipStream_t)stream;
  const int G = c->nranks;
  // Purely local checks first.  A rank that fails them does NOT return yet: it goes through the same collectives as
  // the others with "invalid" in its size words (and nothing to send), every rank sees that in the gathered words,
  // and all of them report MRX_E_ARGUMENT -- on the host in the exact form, in *d_status in the padded form.
  // cap_spans_per_rank and n_global size the collectives themselves and must be the same on every rank.
  const int64_t P = cap_spans_per_rank > 0 ? (n_global + G - 1) / G + 1 : 0;   // prefix slots per rank, padded form
  bool bad = !d_prefix || n_local < 0 || !d_gprefix || !d_gspans || (!d_spans && cap_spans_per_rank > 0) ||
             gprefix_cap < 0 || gspans_cap < 0 || n_global < 0;
  const char* why = "bad arguments";
  if (!bad && cap_spans_per_rank > 0) {
    if (n_global < n_local) { bad = true; why = "n_global: the number of texts of all ranks"; }
    else if (n_local > P) { bad = true; why = "n_local exceeds a contiguous shard of n_global texts"; }
  }
  // staging before the first collective (it may have to grow: mrx_comm_reserve() at set-up keeps that, and its
  // device synchronisation, out of the calls); a rank that cannot get it is the one failure the others cannot
  // be told about
  const size_t send_b = sizeof(int64_t) * (size_t)P;
  const size_t stage_p = sizeof(int64_t) * (size_t)P * G, stage_s = sizeof(int2) * (size_t)(cap_spans_per_rank > 0 ? cap_spans_per_rank : 0) * G;
  if (cap_spans_per_rank > 0) {
    if (n_global >= 0) {
      if (int rc = c->ensure_stage(send_b + stage_p + stage_s + 64)) return rc;
    }
  } else {
    if (int rc = c->ensure_stage(sizeof(int64_t) * (size_t)(n_local > 0 ? n_local : 1))) return rc;
  }
  if (cap_spans_per_rank > 0 && n_global < 0) return mrx::internal_fail(MRX_E_ARGUMENT, "n_global is negative on this rank: it sizes the collectives and must be the same everywhere");
  int64_t* mine = c->d_meta;
  int64_t* all = c->d_meta + kMeta;
  hipLaunchKernelGGL(k_comm_meta, dim3(1), dim3(64), 0, s, bad ? (const int64_t*)nullptr : d_prefix, bad ? int64_t(-1) : n_local,
                     gprefix_cap, gspans_cap, mine);
  CC_NCCL(R.AllGather(mine, all, kMeta * sizeof(int64_t), ncclChar, c->nccl, s));

  if (cap_spans_per_rank > 0) {
    // ---- PADDED form: sizes stay on the device --------------------------------------------------
    int64_t* send_p = (int64_t*)c->d_stage;
    int64_t* st_p = (int64_t*)(c->d_stage + send_b);
    int2* st_s = (int2*)(c->d_stage + send_b + stage_p);
    const int gp = (int)((P + kCommBlock - 1) / kCommBlock < 1024 ? (P + kCommBlock - 1) / kCommBlock : 1024);
    hipLaunchKernelGGL(k_comm_shift, dim3(gp), dim3(kCommBlock), 0, s, bad ? (const int64_t*)nullptr : d_prefix,
                       bad ? int64_t(0) : n_local, all, kMeta, c->rank, send_p, P);
    CC_NCCL(R.GroupStart());
    ncclResult_t e1 = R.AllGather(send_p, st_p, send_b, ncclChar, c->nccl, s);
    // (an invalid rank still ships cap_spans_per_rank slots -- of its staging, nobody reads them)
    ncclResult_t e2 = R.AllGather(bad || !d_spans ? (const void*)st_s : (const void*)d_spans, st_s,
                                  sizeof(int2) * (size_t)cap_spans_per_rank, ncclChar, c->nccl, s);
    CC_NCCL(R.GroupEnd());
    if (e1 != ncclSuccess || e2 != ncclSuccess) return mrx::internal_fail(MRX_E_NO_DEVICE, "ncclAllGather failed");
    if (d_gprefix && d_gspans) {
      const int64_t work = cap_spans_per_rank > P ? cap_spans_per_rank : P;
      const int gx = (int)((work + kCommBlock - 1) / kCommBlock < 2048 ? (work + kCommBlock - 1) / kCommBlock : 2048);
      hipLaunchKernelGGL(k_comm_compact, dim3(gx, G), dim3(kCommBlock), 0, s, all, kMeta, G, st_p, P, st_s, cap_spans_per_rank,
                         d_gprefix, gprefix_cap, (int2*)d_gspans, gspans_cap, d_status);
      CC_HIP(hipGetLastError());
    }
    if (bad) return mrx::internal_fail(MRX_E_ARGUMENT, why);
    return MRX_OK;
  }

  // ---- EXACT form: one read-back of the size words -----------------------------------------------
  CC_HIP(hipMemcpyAsync(c->h_meta, all, sizeof(int64_t) * kMeta * G, hipMemcpyDeviceToHost, s));
  CC_HIP(hipStreamSynchronize(s));
  int64_t N = 0, T = 0, min_pcap = INT64_MAX, min_scap = INT64_MAX;
  bool any_bad = false;
  for (int r = 0; r < G; ++r) {
    if (c->h_meta[kMeta * r] < 0) { any_bad = true; continue; }
    N += c->h_meta[kMeta * r]; T += c->h_meta[kMeta * r + 1];
    if (c->h_meta[kMeta * r + 2] < min_pcap) min_pcap = c->h_meta[kMeta * r + 2];
    if (c->h_meta[kMeta * r + 3] < min_scap) min_scap = c->h_meta[kMeta * r + 3];
  }
  if (N_total) *N_total = N;
  if (T_total) *T_total = T;
  // (every rank has read the same words: all leave here together, or none does)
  if (any_bad) return mrx::internal_fail(MRX_E_ARGUMENT, bad ? why : "another rank passed bad arguments");
  if (N + 1 > min_pcap || T > min_scap) return mrx::internal_fail(MRX_E_CAPACITY, "gather output too small on at least one rank");
  int64_t* send_p = (int64_t*)c->d_stage;
  if (n_local > 0) {
    const int gp = (int)((n_local + kCommBlock - 1) / kCommBlock < 1024 ? (n_local + kCommBlock - 1) / kCommBlock : 1024);
    hipLaunchKernelGGL(k_comm_shift, dim3(gp), dim3(kCommBlock), 0, s, d_prefix, n_local, all, kMeta, c->rank, send_p, n_local);
  }
  CC_HIP(hipMemsetAsync(d_gprefix, 0, sizeof(int64_t), s));
  CC_NCCL(R.GroupStart());
  int64_t toff = 0, soff = 0;
  ncclResult_t bad_nccl = ncclSuccess;
  for (int r = 0; r < G; ++r) {
    const int64_t n_r = c->h_meta[kMeta * r], t_r = c->h_meta[kMeta * r + 1];
    int64_t* pdst = d_gprefix + 1 + toff;
    int32_t* sdst = d_gspans + 2 * soff;
    if (n_r > 0) {
      ncclResult_t e = R.Broadcast(r == c->rank ? (const void*)send_p : (const void*)pdst, pdst, sizeof(int64_t) * (size_t)n_r, ncclChar, r, c->nccl, s);
      if (e != ncclSuccess) bad_nccl = e;
    }
    if (t_r > 0) {
      ncclResult_t e = R.Broadcast(r == c->rank ? (const void*)d_spans : (const void*)sdst, sdst, sizeof(int2) * (size_t)t_r, ncclChar, r, c->nccl, s);
      if (e != ncclSuccess) bad_nccl = e;
    }
    toff += n_r; soff += t_r;
  }
  CC_NCCL(R.GroupEnd());
  if (bad_nccl != ncclSuccess) return mrx::internal_fail(MRX_E_NO_DEVICE, std::string("ncclBroadcast: ") + R.GetErrorString(bad_nccl));
  if (d_status) CC_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
  return MRX_OK;
}

// Testing hook (include/mrx_testing.h): the compaction step of the padded form on staging the caller filled
// as ncclAllGather would have -- the multi-rank path of k_comm_compact can be checked on one GPU.
int mrx_testing_comm_compact(const int64_t* d_meta_all, int meta_stride, int nranks, const int64_t* d_stage_prefix, int64_t P,
                             const int32_t* d_stage_spans, int64_t cap, int64_t* d_gprefix, int64_t gprefix_cap,
                             int32_t* d_gspans, int64_t gspans_cap, int32_t* d_status, void* stream) {
  if (nranks < 1 || !d_meta_all || (meta_stride != 2 && meta_stride != kMeta)) return mrx::internal_fail(MRX_E_ARGUMENT, "bad arguments");
  const int64_t work = cap > P ? cap : P;
  const int gx = (int)((work + kCommBlock - 1) / kCommBlock < 2048 ? (work + kCommBlock - 1) / kCommBlock : 2048);
  hipLaunchKernelGGL(k_comm_compact, dim3(gx > 0 ? gx : 1, nranks), dim3(kCommBlock), 0, (hipStream_t)stream, d_meta_all, meta_stride, nranks,
                     d_stage_prefix, P, (const int2*)d_stage_spans, cap, d_gprefix, gprefix_cap, (int2*)d_gspans,
                     gspans_cap, d_status);
  CC_HIP(hipGetLastError());
  return MRX_OK;
}

// the shift step likewise (what a rank sends: its inclusive ends in the global numbering, zero padded)
int mrx_testing_comm_shift(const int64_t* d_prefix, int64_t n_local, const int64_t* d_meta_all, int meta_stride, int rank,
                           int64_t* d_out, int64_t pad_to, void* stream) {
  const int gp = (int)((pad_to + kCommBlock - 1) / kCommBlock < 1024 ? (pad_to + kCommBlock - 1) / kCommBlock : 1024);
  hipLaunchKernelGGL(k_comm_shift, dim3(gp > 0 ? gp : 1), dim3(kCommBlock), 0, (hipStream_t)stream, d_prefix, n_local, d_meta_all,
                     meta_stride, rank, d_out, pad_to);
  CC_HIP(hipGetLastError());
  return MRX_OK;
}

}  // extern "C"
