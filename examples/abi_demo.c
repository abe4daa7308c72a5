/* Plain-C caller of the drop-in boundary (include/xcx.h): the KATs of xcorr/util/cc.py that
 * SURVEY.md 8c records, and one hour-long window pair, through libxcx.so -- no Python, no torch.
 *
 *   gcc -std=c99 -O2 -I include examples/abi_demo.c -L xcorr_amd/lib -lxcx \
 *       -Wl,-rpath,$PWD/xcorr_amd/lib -lm -o /tmp/abi_demo && /tmp/abi_demo
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "xcx.h"

#define CHECK(call)                                              \
  do {                                                           \
    int rc_ = (call);                                            \
    if (rc_ != XCX_OK) {                                         \
      char msg[512];                                             \
      xcx_last_error(msg, sizeof msg);                           \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, msg);  \
      return 1;                                                  \
    }                                                            \
  } while (0)

int main(void) {
  xcx_ctx* ctx = NULL;
  CHECK(xcx_init(0, &ctx));

  /* cc([1,2,3,4], [0,1,.5,0], normalize=False, pad=True) = [0, .5, 2, 3.5, 5, 4, 0] */
  {
    const float x[4] = {1.f, 2.f, 3.f, 4.f}, y[4] = {0.f, 1.f, .5f, 0.f};
    float out[7];
    xcx_plan* plan = NULL;
    CHECK(xcx_plan_cc1d(ctx, 4, 0, 0, 0, 0, &plan));
    CHECK(xcx_cc1d(plan, x, y, 1, out, 0));
    printf("kat:");
    for (int i = 0; i < 7; ++i) printf(" %.6f", out[i]);
    printf("\n");
    CHECK(xcx_plan_destroy(plan));
  }

  /* one hour at 50 Hz: b = roll(a, 250) + noise -> normalised CCF peaks at lag -250 samples */
  {
    const int n = 180000, L = 2 * n - 1;
    float* a = (float*)malloc(sizeof(float) * n);
    float* b = (float*)malloc(sizeof(float) * n);
    float* out = (float*)malloc(sizeof(float) * L);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {                      /* xorshift, uniform in [-1, 1) */
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      a[i] = (float)((double)(s >> 11) / 4503599627370496.0 - 1.0);
    }
    for (int i = 0; i < n; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      b[i] = a[(i - 250 + n) % n] + 0.5f * (float)((double)(s >> 11) / 4503599627370496.0 - 1.0);
    }
    xcx_plan* plan = NULL;
    CHECK(xcx_plan_cc1d(ctx, n, 0, 0, 0, XCX_NORMALIZE, &plan));
    CHECK(xcx_cc1d(plan, a, b, 1, out, 0));
    int arg = 0;
    for (int i = 1; i < L; ++i) if (out[i] > out[arg]) arg = i;
    printf("hour: L %d argmax %d lag %d max %.6f\n", L, arg, arg - (n - 1), out[arg]);
    CHECK(xcx_plan_destroy(plan));
    free(a); free(b); free(out);
  }
  CHECK(xcx_destroy(ctx));
  return 0;
}
