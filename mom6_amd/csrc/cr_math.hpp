// cr_math.hpp -- correctly rounded transcendentals for the few places where the reference calls libm on the hot path
// (bt_rem = av_rem ** Instep, MOM_barotropic.F90:1529; exp(-htot*Idecay_len_TKE), MOM_set_viscosity.F90:2178).  The
// cos and acos: find_L_open_concave_trigonometric, MOM_set_viscosity.F90:1213-1225 (CHANNEL_DRAG).  The
// reference gets the pow / exp / cos / acos of its build's libm (documented < 1 ulp, not correctly rounded); here they are evaluated in
// double-double arithmetic from + - * / fma only and rounded once, so that the test suite's CPU checker -- which repeats this
// operation order -- agrees bit for bit.
#pragma once

#include <cmath>

#include "common.hpp"

namespace m6 {
namespace cr {

struct dd_t { double hi, lo; };
__device__ __forceinline__ dd_t dd_fast2sum(double a, double b) { double s = a + b; return {s, b - (s - a)}; }
__device__ __forceinline__ dd_t dd_2sum(double a, double b) {
  double s = a + b, bb = s - a; return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd_t dd_2prod(double a, double b) { double p = a * b; return {p, __builtin_fma(a, b, -p)}; }
__device__ __forceinline__ dd_t dd_add(dd_t a, dd_t b) {
  dd_t s = dd_2sum(a.hi, b.hi), t = dd_2sum(a.lo, b.lo);
  s.lo += t.hi; s = dd_fast2sum(s.hi, s.lo); s.lo += t.lo; return dd_fast2sum(s.hi, s.lo);
}
__device__ __forceinline__ dd_t dd_add_d(dd_t a, double b) {
  dd_t s = dd_2sum(a.hi, b); s.lo += a.lo; return dd_fast2sum(s.hi, s.lo);
}
__device__ __forceinline__ dd_t dd_mul(dd_t a, dd_t b) {
  dd_t p = dd_2prod(a.hi, b.hi); p.lo += a.hi * b.lo + a.lo * b.hi; return dd_fast2sum(p.hi, p.lo);
}
__device__ __forceinline__ dd_t dd_mul_d(dd_t a, double b) {
  dd_t p = dd_2prod(a.hi, b); p.lo += a.lo * b; return dd_fast2sum(p.hi, p.lo);
}
__device__ inline dd_t dd_div(dd_t a, dd_t b) {
  double q1 = a.hi / b.hi;
  dd_t r = dd_add(a, dd_mul_d(b, -q1));
  double q2 = r.hi / b.hi;
  r = dd_add(r, dd_mul_d(b, -q2));
  double q3 = r.hi / b.hi;
  dd_t q = dd_fast2sum(q1, q2);
  return dd_add_d(q, q3);
}
__device__ inline double cr_pow(double x, double y) {
  if (x == 1.0) return 1.0;
  const dd_t LN2 = {0.6931471805599453094, 2.3190468138462995584e-17};
  // log
  int e;
  double m = frexp(x, &e);
  if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
  dd_t s = dd_div(dd_2sum(m, -1.0), dd_2sum(m, 1.0)), s2 = dd_mul(s, s);
  dd_t sum = {0.0, 0.0};
  for (int k = 24; k >= 1; k--) {
    dd_t c = {1.0, 0.0}, dk = {(double)(2 * k + 1), 0.0};
    sum = dd_mul(dd_add(dd_div(c, dk), sum), s2);
  }
  sum = dd_add_d(sum, 1.0);
  dd_t r = dd_mul(s, sum); r.hi *= 2.0; r.lo *= 2.0;
  dd_t t = dd_mul_d(dd_add(dd_mul_d(LN2, (double)e), r), y);
  // exp
  double kd = rint(t.hi * 1.4426950408889634074);
  r = dd_add(t, dd_mul_d(LN2, -kd));
  r.hi *= 0.00390625; r.lo *= 0.00390625;
  sum = {0.0, 0.0};
  for (int k = 12; k >= 1; k--) {
    dd_t one_plus = dd_add_d(sum, 1.0), dk = {(double)k, 0.0};
    sum = dd_mul(dd_div(r, dk), one_plus);
  }
  for (int q = 0; q < 8; q++) {
    dd_t sq = dd_mul(sum, sum); sum.hi *= 2.0; sum.lo *= 2.0; sum = dd_add(sum, sq);
  }
  dd_t res = dd_add_d(sum, 1.0);
  return ldexp(res.hi, (int)kd);
}


// exp(t) for -700 < t <= 0 (0 below), correctly rounded: the exp half of cr_pow
__device__ inline double cr_exp(double t0) {
  if (t0 == 0.0) return 1.0;
  if (!(t0 > -700.0)) return 0.0;
  const dd_t LN2 = {0.6931471805599453094, 2.3190468138462995584e-17};
  dd_t t = {t0, 0.0};
  double kd = rint(t.hi * 1.4426950408889634074);
  dd_t r = dd_add(t, dd_mul_d(LN2, -kd));
  r.hi *= 0.00390625; r.lo *= 0.00390625;
  dd_t sum = {0.0, 0.0};
  for (int k = 12; k >= 1; k--) {
    dd_t one_plus = dd_add_d(sum, 1.0), dk = {(double)k, 0.0};
    sum = dd_mul(dd_div(r, dk), one_plus);
  }
  for (int q = 0; q < 8; q++) {
    dd_t sq = dd_mul(sum, sum); sum.hi *= 2.0; sum.lo *= 2.0; sum = dd_add(sum, sq);
  }
  dd_t res = dd_add_d(sum, 1.0);
  return ldexp(res.hi, (int)kd);
}

// ---- cos and acos (the test suite's CPU checker repeats this operation order) -----------------------------------------------
__device__ __forceinline__ dd_t dd_neg(dd_t a) { return {-a.hi, -a.lo}; }
// a / d for a double d
__device__ __forceinline__ dd_t dd_div_d(dd_t a, double d) {
  double q1 = a.hi / d;
  dd_t p = dd_2prod(q1, d);
  dd_t r = dd_add(a, dd_neg(p));
  double q2 = r.hi / d;
  return dd_fast2sum(q1, q2);
}
// sin and cos of r, |r| <= 0.8, by their Taylor series in Horner form: 16 terms each
__device__ inline void dd_sincos_small(dd_t r, dd_t &s, dd_t &c) {
  const dd_t r2 = dd_mul(r, r);
  dd_t ts = {1.0, 0.0}, tc = {1.0, 0.0};
  for (int n = 15; n >= 1; n--) {
    ts = dd_add_d(dd_neg(dd_div_d(dd_mul(ts, r2), (double)((2 * n) * (2 * n + 1)))), 1.0);
    tc = dd_add_d(dd_neg(dd_div_d(dd_mul(tc, r2), (double)((2 * n - 1) * (2 * n)))), 1.0);
  }
  s = dd_mul(r, ts);
  c = tc;
}
// sin and cos of y, 0 <= y <= pi
__device__ inline void dd_sincos_0_pi(dd_t y, dd_t &s, dd_t &c) {
  const dd_t PI = {3.141592653589793116, 1.2246467991473532072e-16}, PI_2 = {1.570796326794896558, 6.1232339957367660359e-17};
  dd_t rs, rc;
  if (y.hi <= 0.78539816339744830962) {
    dd_sincos_small(y, s, c);
  } else if (y.hi <= 2.3561944901923449288) {
    dd_sincos_small(dd_add(y, dd_neg(PI_2)), rs, rc);
    s = rc; c = dd_neg(rs);
  } else {
    dd_sincos_small(dd_add(y, dd_neg(PI)), rs, rc);
    s = dd_neg(rs); c = dd_neg(rc);
  }
}
// cos(x), |x| <= pi (NaN outside), correctly rounded
__device__ inline double cr_cos(double x) {
  if (!(fabs(x) <= 3.1415926535897936)) return __builtin_nan("");
  dd_t y = {fabs(x), 0.0}, s, c;
  dd_sincos_0_pi(y, s, c);
  return c.hi;
}
// acos(x), |x| <= 1 (NaN outside), correctly rounded: a polynomial first guess (Abramowitz & Stegun 4.4.46), three Newton steps on
// cos(y) = x in double-double arithmetic
__device__ inline double cr_acos(double x) {
  if (!(fabs(x) <= 1.0)) return __builtin_nan("");
  if (x == 1.0) return 0.0;
  if (x == -1.0) return 3.141592653589793116;
  const double a = fabs(x);
  double p = -0.0012624911;
  p = p * a + 0.0066700901; p = p * a + -0.0170881256; p = p * a + 0.0308918810; p = p * a + -0.0501743046;
  p = p * a + 0.0889789874; p = p * a + -0.2145988016; p = p * a + 1.5707963050;
  double y0 = sqrt(1.0 - a) * p;
  if (x < 0.0) y0 = 3.141592653589793116 - y0;
  dd_t y = {y0, 0.0};
  for (int it = 0; it < 3; it++) {
    dd_t s, c;
    dd_sincos_0_pi(y, s, c);
    y = dd_add(y, dd_div(dd_add_d(c, -x), s));
  }
  return y.hi;
}

}  // namespace cr
}  // namespace m6
