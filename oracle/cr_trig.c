/*
 * cr_trig.c -- correctly rounded cos and acos for the CPU oracle (TEST INFRASTRUCTURE, see mom6_oracle.h).
 *
 * The reference calls libm's cos and acos in find_L_open_concave_trigonometric (MOM_set_viscosity.F90:1213-1225, CHANNEL_DRAG with
 * TRIG_CHANNEL_DRAG_WIDTHS, its default).  libm's results are documented < 1 ulp, not correctly rounded, and the device's math
 * library differs from the host's; both sides of the parity tests therefore evaluate these two functions in double-double
 * arithmetic from + - * / fma and sqrt only, in the operation order below (mom6_amd/csrc/cr_math.hpp repeats it), and round once.
 * tests/test_set_viscosity.py checks them against 200-bit arithmetic.
 */
#include <math.h>

#include "mom6_oracle.h"

typedef struct { double hi, lo; } dd_t;
static inline dd_t dd_fast2sum(double a, double b) { double s = a + b; dd_t r = {s, b - (s - a)}; return r; }
static inline dd_t dd_2sum(double a, double b) {
  double s = a + b, bb = s - a; dd_t r = {s, (a - (s - bb)) + (b - bb)}; return r;
}
static inline dd_t dd_2prod(double a, double b) { double p = a * b; dd_t r = {p, fma(a, b, -p)}; return r; }
static inline dd_t dd_add(dd_t a, dd_t b) {
  dd_t s = dd_2sum(a.hi, b.hi), t = dd_2sum(a.lo, b.lo);
  s.lo += t.hi; s = dd_fast2sum(s.hi, s.lo); s.lo += t.lo; return dd_fast2sum(s.hi, s.lo);
}
static inline dd_t dd_add_d(dd_t a, double b) {
  dd_t s = dd_2sum(a.hi, b); s.lo += a.lo; return dd_fast2sum(s.hi, s.lo);
}
static inline dd_t dd_neg(dd_t a) { dd_t r = {-a.hi, -a.lo}; return r; }
static inline dd_t dd_mul(dd_t a, dd_t b) {
  dd_t p = dd_2prod(a.hi, b.hi); p.lo += a.hi * b.lo + a.lo * b.hi; return dd_fast2sum(p.hi, p.lo);
}
static inline dd_t dd_mul_d(dd_t a, double b) {
  dd_t p = dd_2prod(a.hi, b); p.lo += a.lo * b; return dd_fast2sum(p.hi, p.lo);
}
static inline dd_t dd_div(dd_t a, dd_t b) {
  double q1 = a.hi / b.hi;
  dd_t r = dd_add(a, dd_mul_d(b, -q1));
  double q2 = r.hi / b.hi;
  r = dd_add(r, dd_mul_d(b, -q2));
  double q3 = r.hi / b.hi;
  dd_t q = dd_fast2sum(q1, q2);
  return dd_add_d(q, q3);
}
/* a / d for a double d */
static inline dd_t dd_div_d(dd_t a, double d) {
  double q1 = a.hi / d;
  dd_t p = dd_2prod(q1, d);
  dd_t r = dd_add(a, dd_neg(p));
  double q2 = r.hi / d;
  return dd_fast2sum(q1, q2);
}

static const dd_t DD_PI = {3.141592653589793116, 1.2246467991473532072e-16};
static const dd_t DD_PI_2 = {1.570796326794896558, 6.1232339957367660359e-17};

/* sin and cos of r, |r| <= 0.8, by their Taylor series in Horner form: 16 terms each (the first neglected one is below 2^-118) */
static void dd_sincos_small(dd_t r, dd_t *s, dd_t *c) {
  const dd_t r2 = dd_mul(r, r);
  dd_t ts = {1.0, 0.0}, tc = {1.0, 0.0};
  for (int n = 15; n >= 1; n--) {
    ts = dd_add_d(dd_neg(dd_div_d(dd_mul(ts, r2), (double)((2 * n) * (2 * n + 1)))), 1.0);
    tc = dd_add_d(dd_neg(dd_div_d(dd_mul(tc, r2), (double)((2 * n - 1) * (2 * n)))), 1.0);
  }
  *s = dd_mul(r, ts);
  *c = tc;
}

/* sin and cos of y, 0 <= y <= pi (a little beyond is harmless) */
static void dd_sincos_0_pi(dd_t y, dd_t *s, dd_t *c) {
  dd_t rs, rc;
  if (y.hi <= 0.78539816339744830962) {
    dd_sincos_small(y, s, c);
  } else if (y.hi <= 2.3561944901923449288) {      /* cos(y) = -sin(y - pi/2), sin(y) = cos(y - pi/2) */
    dd_sincos_small(dd_add(y, dd_neg(DD_PI_2)), &rs, &rc);
    *s = rc; *c = dd_neg(rs);
  } else {                                         /* cos(y) = -cos(y - pi), sin(y) = -sin(y - pi) */
    dd_sincos_small(dd_add(y, dd_neg(DD_PI)), &rs, &rc);
    *s = dd_neg(rs); *c = dd_neg(rc);
  }
}

/* cos(x), |x| <= pi (NaN outside: the one caller's argument is between -2 pi / 3 and -pi / 3) */
double orc_cr_cos(double x) {
  if (!(fabs(x) <= 3.1415926535897936)) return NAN;
  dd_t y = {fabs(x), 0.0}, s, c;
  dd_sincos_0_pi(y, &s, &c);
  return c.hi;
}

/* acos(x), -1 <= x <= 1 (NaN outside, as libm): a polynomial first guess good to 2e-8 (Abramowitz & Stegun 4.4.46), then three
 * Newton steps on cos(y) = x in double-double arithmetic */
double orc_cr_acos(double x) {
  if (!(fabs(x) <= 1.0)) return NAN;
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
    dd_sincos_0_pi(y, &s, &c);
    y = dd_add(y, dd_div(dd_add_d(c, -x), s));
  }
  return y.hi;
}
