/* oracle/coms.c -- TEST INFRASTRUCTURE: a C restatement of the extended-fixed-point (EFP) order-invariant sums of MOM_coms
 * (src/framework/MOM_coms.F90): reproducing_sum_3d :318-497 with real_to_ints :508, ints_to_real :545, increment_ints :558,
 * increment_ints_faster :589, carry_overflow :620, regularize_ints :643.  One PE (sum_across_PEs is the identity).
 * The reference holds no known-answer vectors for these routines; the tests pin this file against exact integer
 * arithmetic instead (an EFP sum of multiples of 2^-138 is exact -- Hallberg & Adcroft 2014), which is a stronger statement
 * than a vector. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "mom6_oracle.h"

#define NI 6
static const int64_t prec = (int64_t)1 << 46;                     /* :28 */
static const double r_prec = 70368744177664.0;                    /* 2.0**46 :29 */
static const double I_prec = 1.0 / 70368744177664.0;              /* :30 */
static const int max_count_prec = (1 << (63 - 46)) - 1;           /* :31 */

typedef struct { int overflow_error, NaN_error; double pr[NI], I_pr[NI], max_efp_float; } efp_state;

static void efp_init(efp_state *s)
{
  s->overflow_error = 0; s->NaN_error = 0;
  s->pr[0] = r_prec*r_prec; s->pr[1] = r_prec; s->pr[2] = 1.0; s->pr[3] = 1.0/r_prec;                          /* :39 */
  s->pr[4] = 1.0/(r_prec*r_prec); s->pr[5] = 1.0/(r_prec*r_prec*r_prec);
  s->I_pr[0] = 1.0/(r_prec*r_prec); s->I_pr[1] = 1.0/r_prec; s->I_pr[2] = 1.0; s->I_pr[3] = r_prec;            /* :42 */
  s->I_pr[4] = r_prec*r_prec; s->I_pr[5] = r_prec*r_prec*r_prec;
  s->max_efp_float = s->pr[0] * (9223372036854775808.0 - 1.0);   /* pr(1) * (2.**63 - 1.) :44 */
}

static int64_t iabs64(int64_t x) { return x < 0 ? -x : x; }

/* increment_ints_faster :589 */
static void increment_ints_faster(efp_state *s, int64_t *int_sum, double r, double *max_mag_term)
{
  if ((r >= 1e30) == (r < 1e30)) { s->NaN_error = 1; return; }
  const int sgn = r < 0.0 ? -1 : 1;
  double rs = fabs(r);
  if (rs > fabs(*max_mag_term)) *max_mag_term = r;
  if (rs > s->max_efp_float) { s->overflow_error = 1; return; }
  for (int i = 0; i < NI; i++) {
    const int64_t ival = (int64_t)(rs*s->I_pr[i]);
    rs = rs - (double)ival*s->pr[i];
    int_sum[i] += sgn*ival;
  }
}

/* real_to_ints :508 with prec_error and no overflow argument: too large a value is the reference's FATAL (here rc 1) */
static int real_to_ints(efp_state *s, double r, int64_t prec_error, int64_t *ints)
{
  memset(ints, 0, NI*sizeof(int64_t));
  if ((r >= 1e30) == (r < 1e30)) { s->NaN_error = 1; return 0; }
  const int sgn = r < 0.0 ? -1 : 1;
  double rs = fabs(r);
  if (!(rs < (double)prec_error*s->pr[0])) return 1;
  for (int i = 0; i < NI; i++) {
    const int64_t ival = (int64_t)(rs*s->I_pr[i]);
    rs = rs - (double)ival*s->pr[i];
    ints[i] = sgn*ival;
  }
  return 0;
}

/* increment_ints :558 */
static void increment_ints(efp_state *s, int64_t *int_sum, const int64_t *int2, int64_t prec_error)
{
  for (int i = NI-1; i >= 1; i--) {
    int_sum[i] += int2[i];
    if (int_sum[i] > prec) { int_sum[i] -= prec; int_sum[i-1] += 1; }
    else if (int_sum[i] < -prec) { int_sum[i] += prec; int_sum[i-1] -= 1; }
  }
  int_sum[0] += int2[0];
  if (iabs64(int_sum[0]) > prec_error) s->overflow_error = 1;
}

/* carry_overflow :620 */
static void carry_overflow(efp_state *s, int64_t *int_sum, int64_t prec_error)
{
  for (int i = NI-1; i >= 1; i--) if (iabs64(int_sum[i]) >= prec) {
    const int num_carry = (int)((double)int_sum[i] * I_prec);
    int_sum[i] -= (int64_t)num_carry*prec;
    int_sum[i-1] += num_carry;
  }
  if (iabs64(int_sum[0]) > prec_error) s->overflow_error = 1;
}

/* regularize_ints :643 */
void orc_efp_regularize(int64_t *int_sum)
{
  for (int i = NI-1; i >= 1; i--) if (iabs64(int_sum[i]) >= prec) {
    const int num_carry = (int)((double)int_sum[i] * I_prec);
    int_sum[i] -= (int64_t)num_carry*prec;
    int_sum[i-1] += num_carry;
  }
  int positive = 1;
  for (int i = 0; i < NI; i++) if (iabs64(int_sum[i]) > 0) { if (int_sum[i] < 0) positive = 0; break; }
  if (positive) {
    for (int i = NI-1; i >= 1; i--) if (int_sum[i] < 0) { int_sum[i] += prec; int_sum[i-1] -= 1; }
  } else {
    for (int i = NI-1; i >= 1; i--) if (int_sum[i] > 0) { int_sum[i] -= prec; int_sum[i-1] += 1; }
  }
}

/* ints_to_real :545 */
double orc_efp_to_real(const int64_t *ints)
{
  efp_state s; efp_init(&s);
  double r = 0.0;
  for (int i = 0; i < NI; i++) r = r + s.pr[i]*(double)ints[i];
  return r;
}

/* reproducing_sum_3d(array(is:ie, js:je, 1:ke), sums, EFP_sum, EFP_lay_sums, err) :318 on one PE.  `a` is the (ke, ncol, nrow)
 * C array, summed over rows j0..j1 and points i0..i1 (0-based, inclusive) of every layer.  lay_sums / efp_lay (6 per layer)
 * select the by-layer branch (:389-447) when either is given, as `present(sums) .or. present(EFP_lay_sums)` does; efp_sum
 * (6) may be NULL.  *err is the reference's code (0, +1 a term too large, +2 overflow, +2 NaN); a NULL err turns those
 * into the return code 1 (the reference's FATAL).  Returns 0 and the sum in *sum. */
int orc_reproducing_sum_3d(const double *a, int nrow, int ncol, int ke, int i0, int i1, int j0, int j1, double *sum,
                           double *lay_sums, int64_t *efp_sum, int64_t *efp_lay, int *err)
{
  efp_state s; efp_init(&s);
  const int64_t prec_error = INT64_MAX;                            /* (2**62 + (2**62 - 1)) / num_PEs() :362 */
  const int isz = i1 + 1 - i0, jsz = j1 + 1 - j0;
  const size_t plane = (size_t)nrow*ncol;
  double max_mag_term = 0.0;
#define A3(i,j,k) a[plane*(k) + (size_t)(j)*nrow + (i)]
  const int by_layer = lay_sums != NULL || efp_lay != NULL;
  const int nacc = by_layer ? ke : 1;
  int64_t *ints = (int64_t*)calloc((size_t)NI*nacc, sizeof(int64_t));
  int64_t tmp[NI];
  for (int k = 0; k < ke; k++) {
    int64_t *acc = ints + (by_layer ? (size_t)NI*k : 0);
    if ((long)jsz*isz < max_count_prec) {                          /* :393 / :451 */
      for (int j = j0; j <= j1; j++) for (int i = i0; i <= i1; i++) increment_ints_faster(&s, acc, A3(i,j,k), &max_mag_term);
      carry_overflow(&s, acc, prec_error);
    } else if (isz < max_count_prec) {
      for (int j = j0; j <= j1; j++) {
        for (int i = i0; i <= i1; i++) increment_ints_faster(&s, acc, A3(i,j,k), &max_mag_term);
        carry_overflow(&s, acc, prec_error);
      }
    } else {
      for (int j = j0; j <= j1; j++) for (int i = i0; i <= i1; i++) {
        if (real_to_ints(&s, A3(i,j,k), prec_error, tmp)) { free(ints); return 1; }
        increment_ints(&s, acc, tmp, prec_error);
      }
    }
  }
  int e = 0;
  if (fabs(max_mag_term) >= (double)prec_error*s.pr[0]) e += 1;
  if (s.overflow_error) e += 2;
  if (s.NaN_error) e += 2;
  if (err) { *err = e; if (e > 0) memset(ints, 0, sizeof(int64_t)*NI*nacc); }
  else if (e > 0) { free(ints); return 1; }
  if (by_layer) {
    double total = 0.0;
    for (int k = 0; k < ke; k++) {
      orc_efp_regularize(ints + (size_t)NI*k);
      const double val = orc_efp_to_real(ints + (size_t)NI*k);
      if (lay_sums) lay_sums[k] = val;
      total = total + val;
    }
    if (efp_lay) memcpy(efp_lay, ints, sizeof(int64_t)*NI*ke);
    if (efp_sum) {
      memset(efp_sum, 0, sizeof(int64_t)*NI);
      s.overflow_error = 0;
      for (int k = 0; k < ke; k++) increment_ints(&s, efp_sum, ints + (size_t)NI*k, prec);     /* no prec_error argument :433 */
    }
    *sum = total;
  } else {
    orc_efp_regularize(ints);
    *sum = orc_efp_to_real(ints);
    if (efp_sum) memcpy(efp_sum, ints, sizeof(int64_t)*NI);
  }
#undef A3
  free(ints);
  return 0;
}
