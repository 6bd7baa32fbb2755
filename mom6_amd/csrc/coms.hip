// coms.hip -- the order-invariant extended-fixed-point (EFP) sums of MOM_coms on the device: reproducing_sum of a field that
// lives in HBM (src/framework/MOM_coms.F90: reproducing_sum_3d :318, real_to_ints :508, carry_overflow :620, regularize_ints
// :643; Hallberg & Adcroft 2014).  Every value is split into six 46-bit integer limbs exactly as increment_ints_faster
// (:589) does; integer addition is associative, so the limbs can be accumulated in any order -- per thread over a strip of
// rows, across the wave with DPP shuffles, across the grid with 64-bit atomics -- and the regularised result is the
// reference's, bit for bit, on any tiling of the domain.
//
// Accumulators.  A limb of one value is below 2^46 in magnitude (the first limb excepted), so a thread's strip (<= 32 rows)
// stays below 2^51 and a wave's sum below 2^57.  The grid-wide sums would leave 64 bits (the reason the reference carries
// after every row), so limbs 2..6 are accumulated as two 64-bit atomics each -- the low 32 bits and the arithmetic high
// part -- which is exact for up to 2^31 waves; the host recombines them in 128-bit integers and carries.  The first limb
// is accumulated modulo 2^64, which is the reference's own arithmetic for it.
#include "efp.hpp"

namespace {
using namespace m6efp;

typedef __int128 i128;
inline long long iabs(long long x) { return x < 0 ? -x : x; }

// carry_overflow :620 on 128-bit partial sums: limbs 2..6 into (-prec, prec), the first limb modulo 2^64
void carry(const i128 *t, long long *ints) {
  i128 v[NI];
  for (int n = 0; n < NI; n++) v[n] = t[n];
  for (int n = NI - 1; n >= 1; n--) {
    const i128 num_carry = v[n] / PREC;                     // truncates towards zero, as int() does
    v[n] -= num_carry * PREC; v[n - 1] += num_carry;
  }
  ints[0] = (long long)(unsigned long long)v[0];
  for (int n = 1; n < NI; n++) ints[n] = (long long)v[n];
}

// regularize_ints :643
void regularize(long long *s) {
  const double I_prec = 1.0 / 70368744177664.0;
  for (int i = NI - 1; i >= 1; i--) if (iabs(s[i]) >= PREC) {
    const int num_carry = (int)((double)s[i] * I_prec);
    s[i] -= (long long)num_carry * PREC; s[i - 1] += num_carry;
  }
  bool positive = true;
  for (int i = 0; i < NI; i++) if (iabs(s[i]) > 0) { if (s[i] < 0) positive = false; break; }
  if (positive) { for (int i = NI - 1; i >= 1; i--) if (s[i] < 0) { s[i] += PREC; s[i - 1] -= 1; } }
  else          { for (int i = NI - 1; i >= 1; i--) if (s[i] > 0) { s[i] -= PREC; s[i - 1] += 1; } }
}

// ints_to_real :545
double to_real(const EfpConst &c, const long long *s) {
  double r = 0.0;
  for (int i = 0; i < NI; i++) r = r + c.pr[i] * (double)s[i];
  return r;
}

// increment_ints :558 without prec_error
void increment(long long *sum, const long long *add) {
  for (int i = NI - 1; i >= 1; i--) {
    sum[i] += add[i];
    if (sum[i] > PREC) { sum[i] -= PREC; sum[i - 1] += 1; }
    else if (sum[i] < -PREC) { sum[i] += PREC; sum[i - 1] -= 1; }
  }
  sum[0] += add[0];
}

// sum_across_PEs of 64-bit integers through the 32-bit exchange the domain offers: four signed 16-bit pieces per value
// (exact for up to 2^15 PEs, the reference's own limit being 2^17 - 1, :352)
int sum_across_PEs_i64(mom6hip_ctx *ctx, long long *v, int n) {
  if (!m6::multi_tile(ctx)) return 0;
  std::vector<int32_t> w((size_t)4 * n);
  for (int q = 0; q < n; q++) {
    const bool neg = v[q] < 0;
    const unsigned long long m = neg ? 0ull - (unsigned long long)v[q] : (unsigned long long)v[q];
    for (int p = 0; p < 4; p++) { const int32_t piece = (int32_t)((m >> (16 * p)) & 0xffffull); w[4 * q + p] = neg ? -piece : piece; }
  }
  if (int rc = m6::sum_across_PEs(ctx, w.data(), 4 * n)) return rc;
  for (int q = 0; q < n; q++) {
    unsigned long long t = 0ull;                            // modulo 2^64, like the integers it stands for
    for (int p = 0; p < 4; p++) t += (unsigned long long)(long long)w[4 * q + p] << (16 * p);
    v[q] = (long long)t;
  }
  return 0;
}


}  // namespace

int m6efp::efp_finish(mom6hip_ctx *ctx, const std::vector<unsigned long long> &res, int nk, long long npts2d, double *sum, double *lay_sums,
                      int64_t *efp_sum, int64_t *efp_lay, int64_t *npoints, int32_t *err, bool pe_sum_unregularized) {
  const EfpConst c = efp_const();
  // the number of PEs, for prec_error = (2**63 - 1) / num_PEs() :362 (asked of the domain once)
  if (ctx->num_PEs == 0) {
    int32_t one = 1;
    if (m6::multi_tile(ctx)) { if (int rc = m6::sum_across_PEs(ctx, &one, 1)) return rc; }
    ctx->num_PEs = one;
  }
  const long long prec_error = INT64_MAX / ctx->num_PEs;

  const bool by_layer = lay_sums != nullptr || efp_lay != nullptr;          // :389
  const int nsum = by_layer ? nk : 1;
  std::vector<long long> ints((size_t)NI * nsum, 0ll);
  bool overflow_error = (res[(size_t)ACC * nk + 1] & 2ull) != 0;
  const bool NaN_error = (res[(size_t)ACC * nk + 1] & 1ull) != 0;
  {
    i128 tot[NI] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < nk; k++) {
      const unsigned long long *o = &res[(size_t)ACC * k];
      i128 t[NI];
      t[0] = (i128)(long long)o[0];
      for (int n = 1; n < NI; n++) t[n] = (i128)o[2 * n - 1] + (i128)(long long)o[2 * n] * (i128)4294967296ll;
      if (by_layer) {
        carry(t, &ints[(size_t)NI * k]);
        if (iabs(ints[(size_t)NI * k]) > prec_error) overflow_error = true;
      } else {
        for (int n = 0; n < NI; n++) tot[n] += t[n];
      }
    }
    if (!by_layer) {
      carry(tot, ints.data());
      if (iabs(ints[0]) > prec_error) overflow_error = true;
    }
  }
  double max_mag; { const unsigned long long b = res[(size_t)ACC * nk]; memcpy(&max_mag, &b, 8); }
  int e = 0;                                                                // :411-416 / :469-474
  if (max_mag >= (double)prec_error * c.pr[0]) e += 1;
  if (overflow_error) e += 2;
  if (NaN_error) e += 2;
  if (err) {
    *err = e;
    if (e > 0) std::fill(ints.begin(), ints.end(), 0ll);
  } else {
    M6_REQUIRE(!NaN_error, "NaN in input field of reproducing_sum(_3d).");
    M6_REQUIRE(!(max_mag >= (double)prec_error * c.pr[0]), "Overflow in reproducing_sum(_3d) conversion of %13.5E", max_mag);
    M6_REQUIRE(!overflow_error, "Overflow in reproducing_sum(_3d).");
  }

  if (pe_sum_unregularized) {      // reproducing_EFP_sum_2d(only_on_PE) :139-213 ends in regularize_ints on every PE ...
    for (int q = 0; q < nsum; q++) regularize(&ints[(size_t)NI * q]);
  }
  // sum_across_PEs of the limbs (and of the number of points, for the means of MOM_checksums' subStats)
  {
    std::vector<long long> x(ints);
    x.push_back(npts2d);
    if (int rc = sum_across_PEs_i64(ctx, x.data(), (int)x.size())) return rc;
    std::copy(x.begin(), x.end() - 1, ints.begin());
    if (npoints) *npoints = (int64_t)x.back() * nk;
  }

  if (pe_sum_unregularized) {      // ... and EFP_sum_across_PEs (:789-835) only carries the overflows of the sum
    const double I_prec = 1.0 / 70368744177664.0;
    for (int i = NI - 1; i >= 1; i--) if (iabs(ints[i]) >= PREC) {      // carry_overflow :620
      const int num_carry = (int)((double)ints[i] * I_prec);
      ints[i] -= (long long)num_carry * PREC; ints[i - 1] += num_carry;
    }
    M6_REQUIRE(iabs(ints[0]) <= prec_error, "Overflow in EFP_list_sum_across_PEs.");
    *sum = to_real(c, ints.data());
    if (efp_sum) for (int n = 0; n < NI; n++) efp_sum[n] = ints[n];
    return 0;
  }
  if (by_layer) {
    double total = 0.0;
    for (int k = 0; k < nk; k++) {
      regularize(&ints[(size_t)NI * k]);
      const double val = to_real(c, &ints[(size_t)NI * k]);
      if (lay_sums) lay_sums[k] = val;
      total = total + val;
    }
    if (efp_lay) for (size_t q = 0; q < (size_t)NI * nk; q++) efp_lay[q] = ints[q];
    if (efp_sum) {                                                          // :431-434
      long long s[NI] = {0, 0, 0, 0, 0, 0};
      for (int k = 0; k < nk; k++) increment(s, &ints[(size_t)NI * k]);
      for (int n = 0; n < NI; n++) efp_sum[n] = s[n];
    }
    *sum = total;
  } else {
    regularize(ints.data());
    *sum = to_real(c, ints.data());
    if (efp_sum) for (int n = 0; n < NI; n++) efp_sum[n] = ints[n];
  }
  return 0;
}

extern "C" int mom6hip_reproducing_sum(mom6hip_ctx_t *ctx, const double *field, int32_t pos, int32_t nk, double *sum, double *lay_sums,
                                       int64_t *efp_sum, int64_t *efp_lay, int64_t *npoints, int32_t *err, int32_t memspace) {
  M6_REQUIRE(ctx && field && sum, "mom6hip_reproducing_sum: null argument");
  M6_REQUIRE(pos >= MOM6HIP_POS_H && pos <= MOM6HIP_POS_Q && nk >= 1, "mom6hip_reproducing_sum: bad staggering or layer count");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "mom6hip_reproducing_sum: bad memspace");
  const m6::GridDev g = ctx->g;
  const int xs = (pos == MOM6HIP_POS_U || pos == MOM6HIP_POS_Q) ? 1 : 0, ys = (pos == MOM6HIP_POS_V || pos == MOM6HIP_POS_Q) ? 1 : 0;
  const int nrow = g.nih + xs, ncol = g.njh + ys;
  // the h-point computational domain in the field's own indexing, whatever the staggering (MOM_checksums.F90:1079)
  const int i0 = g.isc - (g.isd - xs), i1 = g.iec - (g.isd - xs), j0 = g.jsc - (g.jsd - ys), j1 = g.jec - (g.jsd - ys);
  const int isz = i1 + 1 - i0, jsz = j1 + 1 - j0;
  M6_REQUIRE(isz >= 1 && jsz >= 1, "mom6hip_reproducing_sum: empty computational domain");
  m6::Stager st(ctx, memspace);
  const double *d = st.in(field, sizeof(double) * (size_t)nrow * ncol * nk);
  M6_REQUIRE(!st.failed() && d, "mom6hip_reproducing_sum: staging failed");
  const long plane = (long)nrow * ncol;
  std::vector<unsigned long long> res;
  if (int rc = efp_reduce(ctx, [=] __device__(int i, int j, int k) { return d[plane * k + (long)j * nrow + i]; }, i0, i1, j0, j1, nk, res)) return rc;
  if (int rc = efp_finish(ctx, res, nk, (long long)isz * jsz, sum, lay_sums, efp_sum, efp_lay, npoints, err)) return rc;
  return st.finish();
}
