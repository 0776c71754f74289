/* Plain C client of include/mrx.h: compile a pattern, run findall / search / sub on a small host
 * batch through the *_batch wrappers, print the results.  Built with gcc (no HIP headers needed on
 * the client side); tests/test_c_abi.py checks that it compiles and links everywhere and that its
 * output is right on a GPU box. */
#include <stdio.h>
#include <string.h>

#include "mrx.h"

int main(void) {
  const char* pat = "[a-z]+\\d+";
  mrx_handle* h = NULL;
  if (mrx_compile(pat, strlen(pat), &h) != MRX_OK) {
    printf("compile failed: %s\n", mrx_last_error());
    return 2;
  }
  printf("engine=%s\n", mrx_engine_type(h));
  const char* texts = "hello123 world456none" "QQab12ZZ";
  const int64_t off[4] = {0, 17, 21, 29};
  int64_t prefix[4], total = 0;
  int32_t spans[2 * 16];
  int rc = mrx_findall_batch(h, (const uint8_t*)texts, off, 3, prefix, spans, 16, &total);
  if (rc != MRX_OK) {
    printf("findall failed (%d): %s\n", rc, mrx_last_error());
    mrx_free(h);
    return rc == MRX_E_NO_DEVICE ? 3 : 4;
  }
  printf("total=%lld\n", (long long)total);
  for (int i = 0; i < 3; ++i)
    for (int64_t k = prefix[i]; k < prefix[i + 1]; ++k) printf("text%d [%d,%d)\n", i, spans[2 * k], spans[2 * k + 1]);
  int32_t s[3], e[3];
  if (mrx_search_batch(h, (const uint8_t*)texts, off, 3, s, e) != MRX_OK) return 5;
  for (int i = 0; i < 3; ++i) printf("search%d %d %d\n", i, s[i], e[i]);
  int64_t out_off[4], out_total = 0;
  uint8_t out[128];
  if (mrx_sub_batch(h, "#", 1, 0, (const uint8_t*)texts, off, 3, out_off, out, sizeof out, &out_total) != MRX_OK) return 6;
  printf("sub=%.*s\n", (int)out_total, (const char*)out);
  mrx_handle* bad = NULL;
  if (mrx_compile("[abc", 4, &bad) == MRX_E_SYNTAX) printf("syntax: %s\n", mrx_last_error());
  mrx_free(h);
  return 0;
}
