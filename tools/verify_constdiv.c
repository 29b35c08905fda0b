// Exhaustive check of tempo_div_r64 (unproject_math.h): for EVERY fp32 numerator a,
//   RN32((double)a * r) == a / b   with r = 1.0 / (double)b  and with r from two Newton
// steps on an fp32 estimate that is a few ulp off (what the kernel's v_rcp_f32 gives).
// Mismatches are counted apart for results in the fp32 denormal range, where the form
// may miss exact ties (the kernels redo those: normalise_tile) -- everywhere else the
// count must be 0.  normalise_tile recognises such a result by its DENORMAL class; a product
// that rounds to zero while the true quotient rounds to 2^-149 (a quotient within 2^-50 of
// 2^-150) would escape that test.  It needs a divisor within 2^-50, relatively, of an even
// integer multiple of the numerator's 2^-149 count; the kernels' divisors there are
// count + 1e-6, so it cannot occur, and this program counts such results with the ties.
//   gcc -O2 -fopenmp -ffp-contract=off tools/verify_constdiv.c -o /tmp/vcd -lm && /tmp/vcd 960 512 239 127 ...
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double newton(float b, float est) {
  const double d = (double)b;
  double r = (double)est;
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

int main(int argc, char** argv) {
  int fail = 0;
  for (int i = 1; i < argc; i++) {
    const float b = strtof(argv[i], 0);
    const double r0 = 1.0 / (double)b;
    float est = 1.0f / b;
    uint32_t eu;
    memcpy(&eu, &est, 4);
    eu += 3;  // a sloppy estimate
    memcpy(&est, &eu, 4);
    const double r1 = newton(b, est);
    unsigned long bad = 0, bad_den = 0, badn = 0, badn_den = 0;
#pragma omp parallel for reduction(+ : bad, bad_den, badn, badn_den)
    for (uint64_t u = 0; u <= 0xffffffffull; u++) {
      const uint32_t v = (uint32_t)u;
      float a;
      memcpy(&a, &v, 4);
      const float ref = a / b;
      const float g0 = (float)((double)a * r0), g1 = (float)((double)a * r1);
      const int den = ref != 0.0f && fabsf(ref) < 0x1p-126f;
      const int e0 = ref != ref ? g0 == g0 : memcmp(&ref, &g0, 4) != 0;
      const int e1 = ref != ref ? g1 == g1 : memcmp(&ref, &g1, 4) != 0;
      // a wrong result at a denormal tie is itself denormal (or the tie's other side)
      if (e0) { if (den || (g0 != 0.0f && fabsf(g0) < 0x1p-126f)) bad_den++; else bad++; }
      if (e1) { if (den || (g1 != 0.0f && fabsf(g1) < 0x1p-126f)) badn_den++; else badn++; }
    }
    printf("b=%.9g  host reciprocal: %lu wrong (+%lu at denormal ties)   newton: %lu wrong (+%lu at denormal ties)\n",
           (double)b, bad, bad_den, badn, badn_den);
    fail |= bad != 0 || badn != 0;
  }
  return fail;
}
