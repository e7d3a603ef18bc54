/* Is (float)((double)x / C) == (float)((double)x * (1.0 / C)) for EVERY float x with |x| <= lim?  (jade_shade.h, sample_hdr: the HIP
 * module multiplies where the reference - and the oracle - divide, PathTrace.cu:689-690.)  Exhaustive: prints the mismatch count.
 * usage: dpdiv_exhaustive   ->  two lines "C R inputs mismatches" */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "jade_fpmath.h"
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(void) {
  const double C[2] = {2.0 * JADE_PI_D, JADE_PI_D};
  const float lim[2] = {3.2f, 1.6f};
  int rc = 0;
  for (int c = 0; c < 2; ++c) {
    const double R = 1.0 / C[c];
    uint64_t bad = 0, n = 0;
    for (uint32_t u = 0; u < 0x7f800000u; ++u) {
      const float x = u2f(u);
      if (x > lim[c]) break;
      for (int sgn = 0; sgn < 2; ++sgn) {
        const float xs = sgn ? -x : x;
        const float a = (float)((double)xs / C[c]), b = (float)((double)xs * R);
        ++n;
        if (memcmp(&a, &b, 4)) ++bad;
      }
    }
    printf("%.17g %.17g %llu %llu\n", C[c], R, (unsigned long long)n, (unsigned long long)bad);
    if (bad) rc = 1;
  }
  return rc;
}
