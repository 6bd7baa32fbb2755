// efp.hpp -- the device side of MOM_coms' extended-fixed-point sums, shared by coms.hip (reproducing_sum of a field) and
// sum_output.hip (the integrals of write_energy, whose integrands are formed on the fly).  See coms.hip for the scheme.
#pragma once
#include "common.hpp"

namespace m6efp {

constexpr int NI = 6;                         // MOM_coms.F90:36
constexpr int STRIP = 32;                     // rows per thread
constexpr long long PREC = 1ll << 46;         // :28
constexpr int ACC = 1 + 2 * (NI - 1);         // per layer: limb 1, then (low, high) of limbs 2..6

struct EfpConst { double pr[NI], I_pr[NI], max_efp_float; };
inline EfpConst efp_const() {
  EfpConst c;
  const double r_prec = 70368744177664.0;     // 2.0**46 :29
  c.pr[0] = r_prec * r_prec; c.pr[1] = r_prec; c.pr[2] = 1.0; c.pr[3] = 1.0 / r_prec;                         // :39
  c.pr[4] = 1.0 / (r_prec * r_prec); c.pr[5] = 1.0 / (r_prec * r_prec * r_prec);
  c.I_pr[0] = 1.0 / (r_prec * r_prec); c.I_pr[1] = 1.0 / r_prec; c.I_pr[2] = 1.0; c.I_pr[3] = r_prec;         // :42
  c.I_pr[4] = r_prec * r_prec; c.I_pr[5] = r_prec * r_prec * r_prec;
  c.max_efp_float = c.pr[0] * (9223372036854775808.0 - 1.0);                                                  // :44
  return c;
}

__device__ __forceinline__ long long shfl_down_ll(long long x, int off) {
  return (long long)__shfl_down((unsigned long long)x, off);
}

// acc[ACC * k ...] += the limbs of f(i, j, k) over i0..i1, j0..j1 (f: the value at a point of the caller's index space);
// misc[0] = max |term| (as bits: non-negative doubles order as integers), misc[1] |= 1 for a NaN, 2 for a term with no
// EFP representation
template <class F>
__global__ __launch_bounds__(256) void efp_sum_kernel(F f, int i0, int i1, int j0, int j1, EfpConst c, unsigned long long *__restrict__ acc,
                                                      unsigned long long *__restrict__ misc) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x, jb = j0 + blockIdx.y * STRIP, k = blockIdx.z;
  long long s[NI] = {0, 0, 0, 0, 0, 0};
  unsigned long long mag = 0ull, bad = 0ull;
  if (i <= i1) {
    const int je = jb + STRIP - 1 < j1 ? jb + STRIP - 1 : j1;
    for (int j = jb; j <= je; j++) {
      const double r = f(i, j, k);
      if ((r >= 1e30) == (r < 1e30)) { bad |= 1ull; continue; }        // increment_ints_faster :603
      double rs = fabs(r);
      const unsigned long long b = (unsigned long long)__double_as_longlong(rs);
      mag = b > mag ? b : mag;
      if (rs > c.max_efp_float) { bad |= 2ull; continue; }             // :609
      const bool neg = r < 0.0;
#pragma unroll
      for (int n = 0; n < NI; n++) {
        const long long ival = (long long)(rs * c.I_pr[n]);            // :615-617
        rs = rs - (double)ival * c.pr[n];
        s[n] += neg ? -ival : ival;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int n = 0; n < NI; n++) s[n] += shfl_down_ll(s[n], off);
    const unsigned long long m2 = __shfl_down(mag, off);
    mag = m2 > mag ? m2 : mag;
    bad |= __shfl_down(bad, off);
  }
  if ((threadIdx.x & 63) == 0) {
    unsigned long long *o = acc + (size_t)ACC * k;
    if (s[0]) atomicAdd(&o[0], (unsigned long long)s[0]);
#pragma unroll
    for (int n = 1; n < NI; n++) if (s[n]) {
      atomicAdd(&o[2 * n - 1], (unsigned long long)s[n] & 0xffffffffull);
      atomicAdd(&o[2 * n], (unsigned long long)(s[n] >> 32));
    }
    if (mag) atomicMax(&misc[0], mag);
    if (bad) atomicOr(&misc[1], bad);
  }
}

// The limbs of sum f(i, j, k) over the ranges, per layer, into `res` (ACC * nk accumulators, then max |term| and the flags):
// launches on the context's stream and waits for the result.
template <class F>
int efp_reduce(mom6hip_ctx *ctx, F f, int i0, int i1, int j0, int j1, int nk, std::vector<unsigned long long> &res) {
  const size_t nacc = (size_t)ACC * nk + 2;
  M6_REQUIRE(ctx->efp_acc.reserve(nacc * sizeof(unsigned long long)) == 0, "reproducing_sum: out of device memory");
  unsigned long long *acc = (unsigned long long *)ctx->efp_acc.p;
  M6_HIP(hipMemsetAsync(acc, 0, nacc * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(efp_sum_kernel<F>, dim3((i1 - i0 + 256) / 256, (j1 - j0 + STRIP) / STRIP, nk), dim3(256), 0, ctx->stream, f, i0, i1, j0, j1,
                     efp_const(), acc, acc + (size_t)ACC * nk);
  M6_HIP(hipGetLastError());
  res.resize(nacc);
  M6_HIP(hipMemcpyAsync(res.data(), acc, nacc * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  M6_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// The host side of reproducing_sum_3d (:389-497) from the accumulators of efp_reduce: carry, the error code, the sum over
// PEs, regularize_ints, ints_to_real.  npts2d: the points of one layer on this PE.  pe_sum_unregularized: the form of
// reproducing_sum_EFP(only_on_PE) + EFP_sum_across_PEs (write_energy's salt and heat): every PE's integers are
// regularised BEFORE the sum across PEs and only carried after it (:789-835).
int efp_finish(mom6hip_ctx *ctx, const std::vector<unsigned long long> &res, int nk, long long npts2d, double *sum, double *lay_sums,
               int64_t *efp_sum, int64_t *efp_lay, int64_t *npoints, int32_t *err, bool pe_sum_unregularized = false);

}  // namespace m6efp
