/* Plain C client of include/mrx.h + include/mrx_comm.h: the device-resident findall of one rank followed by the
 * results exchange (RCCL behind the C ABI), at world size 1 -- what a rank of a sharded batch does, minus the other
 * ranks.  Device buffers come from the HIP C API (hipMalloc / hipMemcpy); tests/test_comm.py checks that it builds
 * everywhere and that its output is right on a GPU box. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <string.h>

#include "mrx.h"
#include "mrx_comm.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s\n", (int)e_, #x); return 10; } } while (0)
#define CHECK_MRX(x) do { int rc_ = (x); if (rc_ != MRX_OK) { printf("mrx error %d at %s: %s\n", rc_, #x, mrx_last_error()); return 11; } } while (0)

int main(void) {
  const char* pat = "[a-z]+\\d+";
  mrx_handle* h = NULL;
  CHECK_MRX(mrx_compile(pat, strlen(pat), &h));
  const char* texts = "hello123 world456none" "QQab12ZZ";
  const int64_t off[4] = {0, 17, 21, 29};
  uint8_t* d_data; int64_t *d_off, *d_prefix, *d_gprefix; int32_t *d_spans, *d_gspans, *d_s, *d_e, *d_gs, *d_status;
  CHECK_HIP(hipMalloc((void**)&d_data, 64));
  CHECK_HIP(hipMalloc((void**)&d_off, sizeof off));
  CHECK_HIP(hipMalloc((void**)&d_prefix, 4 * sizeof(int64_t)));
  CHECK_HIP(hipMalloc((void**)&d_gprefix, 4 * sizeof(int64_t)));
  CHECK_HIP(hipMalloc((void**)&d_spans, 16 * 2 * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void**)&d_gspans, 16 * 2 * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void**)&d_s, 3 * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void**)&d_e, 3 * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void**)&d_gs, 3 * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void**)&d_status, sizeof(int32_t)));
  CHECK_HIP(hipMemcpy(d_data, texts, 29, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_off, off, sizeof off, hipMemcpyHostToDevice));

  uint8_t id[MRX_COMM_ID_BYTES];
  mrx_comm* comm = NULL;
  CHECK_MRX(mrx_comm_unique_id(id));              /* rank 0; the other ranks receive these 128 bytes */
  CHECK_MRX(mrx_comm_init(id, 1, 0, &comm));
  printf("comm rank=%d size=%d\n", mrx_comm_rank(comm), mrx_comm_size(comm));

  int64_t total = 0;
  CHECK_MRX(mrx_findall_dev(h, d_data, d_off, 3, d_prefix, d_spans, 16, &total, NULL));
  printf("local total=%lld\n", (long long)total);

  /* exact form: sizes read back once */
  int64_t N = 0, T = 0;
  CHECK_MRX(mrx_allgatherv_spans(comm, d_prefix, 3, d_spans, 0, 3, d_gprefix, 4, d_gspans, 16, &N, &T, NULL, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  int64_t gp[4]; int32_t gs[32];
  CHECK_HIP(hipMemcpy(gp, d_gprefix, sizeof gp, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gs, d_gspans, sizeof(int32_t) * 2 * (size_t)T, hipMemcpyDeviceToHost));
  printf("exact N=%lld T=%lld\n", (long long)N, (long long)T);
  for (int i = 0; i < 3; ++i)
    for (int64_t k = gp[i]; k < gp[i + 1]; ++k) printf("exact text%d [%d,%d)\n", i, gs[2 * k], gs[2 * k + 1]);

  /* padded form: nothing is read back by the library */
  CHECK_HIP(hipMemset(d_gprefix, 0xFF, 4 * sizeof(int64_t)));
  CHECK_HIP(hipMemset(d_gspans, 0xFF, 32 * sizeof(int32_t)));
  CHECK_MRX(mrx_allgatherv_spans(comm, d_prefix, 3, d_spans, 16, 3, d_gprefix, 4, d_gspans, 16, NULL, NULL, d_status, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  int32_t status = -1;
  CHECK_HIP(hipMemcpy(&status, d_status, sizeof status, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gp, d_gprefix, sizeof gp, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gs, d_gspans, sizeof(int32_t) * 2 * (size_t)gp[3], hipMemcpyDeviceToHost));
  printf("padded status=%d T=%lld\n", (int)status, (long long)gp[3]);
  for (int i = 0; i < 3; ++i)
    for (int64_t k = gp[i]; k < gp[i + 1]; ++k) printf("padded text%d [%d,%d)\n", i, gs[2 * k], gs[2 * k + 1]);

  /* fixed-size results: search */
  CHECK_MRX(mrx_search_dev(h, d_data, d_off, 3, d_s, d_e, NULL));
  CHECK_MRX(mrx_allgather_fixed(comm, d_s, d_gs, 3 * sizeof(int32_t), NULL));
  CHECK_HIP(hipDeviceSynchronize());
  int32_t s[3];
  CHECK_HIP(hipMemcpy(s, d_gs, sizeof s, hipMemcpyDeviceToHost));
  printf("search starts %d %d %d\n", s[0], s[1], s[2]);

  mrx_comm_free(comm);
  mrx_free(h);
  printf("done\n");
  return 0;
}
