/* Test-only: exports detmath.h functions over arrays so tests/test_detmath.py can bound them against libm. */
#include "../../atm-raytracer_amd/csrc/detmath.h"
#include <stddef.h>
#define V1(name) void t_##name(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_##name(x[i]); }
V1(sin) V1(cos) V1(tan) V1(asin) V1(atan) V1(exp) V1(log) V1(sqrt) V1(floor)
void t_atan2(const double* a, const double* b, double* y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_atan2(a[i], b[i]); }
void t_pow(const double* a, const double* b, double* y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_pow(a[i], b[i]); }
void t_div_r_seq(const double* a, const double* b, double* y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_div_r_seq(a[i], b[i], 1.0 / b[i]); }
/* host counterparts of atmrt_math_probe (tests/test_gpu_detmath.py): on the host dm_div / dm_div_r / dm_sqrt_inrange ARE the IEEE operations */
void t_div(const double* a, const double* b, double* y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_div(a[i], b[i]); }
void t_div_r(const double* a, const double* b, double* y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_div_r(a[i], b[i], 1.0 / b[i]); }
V1(sqrt_inrange)
void t_sincos(const double* x, double* s, double* c, size_t n) { for (size_t i = 0; i < n; i++) dm_sincos(x[i], &s[i], &c[i]); }
void t_pow3(const double* a, const double* b, double* y0, double* y1, size_t n) {
  for (size_t i = 0; i < n; i++) {
    y0[i] = dm_pow(a[i], b[i]);
    y1[i] = dm_pow(a[i] * 0.99999981, b[i]) + dm_pow(a[i] * 1.00000019, b[i]);
  }
}
void t_div3(const double* a, const double* b, double* y0, double* y1, size_t n) {
  for (size_t i = 0; i < n; i++) {
    double q0;
    dm_div3(a[i], b[i], a[i], b[i] * 0.99999976158142090, a[i], b[i] * 1.00000047683715820, &q0, &y0[i], &y1[i]);
  }
}
void t_div3_seeded(const double* a, const double* b, double* y0, double* y1, size_t n) {
  for (size_t i = 0; i < n; i++) {
    double q0;
    dm_div3_seeded(a[i], b[i], a[i], b[i] * 0.99999976158142090, a[i], b[i] * 1.00000023841857910, 0, 0.0, &q0, &y0[i], &y1[i]);
  }
}
void t_div3_seed_z(const double* a, const double* b, double* y0, double* y1, size_t n) {
  for (size_t i = 0; i < n; i++) {
    double q1;
    dm_div3_seeded(a[i], b[i], a[i], b[i] * 0.99999976158142090, a[i], b[i] * 1.00000023841857910, 1, 2.0 - b[i], &y0[i], &q1, &y1[i]);
  }
}
void t_div_seed_n(const double* a, const double* b, double* y, size_t n) {
  for (size_t i = 0; i < n; i++) y[i] = dm_div_seeded(a[i], 1.0 + b[i], 1.0 - b[i]);
}
void t_pow3_shared(const double* a, const double* b, double* y0, double* y1, size_t n) {
  for (size_t i = 0; i < n; i++) {
    y0[i] = dm_pow(a[i], b[i]);
    y1[i] = dm_pow(a[i] * 0.99999976158142090, b[i]) + dm_pow(a[i] * 1.00000023841857910, b[i]);
  }
}
