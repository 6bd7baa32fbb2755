// cr_math.hpp -- correctly rounded transcendentals for the few places where the reference calls libm on the hot path
// (bt_rem = av_rem ** Instep, MOM_barotropic.F90:1529; exp(-htot*Idecay_len_TKE), MOM_set_viscosity.F90:2178).  The
// reference gets the pow / exp of its build's libm (documented < 1 ulp, not correctly rounded); here they are evaluated in
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

}  // namespace cr
}  // namespace m6
