// mixedlayer_restrat.hip -- mixedlayer_restrat of src/parameterizations/lateral/MOM_mixed_layer_restrat.F90 on gfx950: the driver
// :135-172, mixedlayer_restrat_OM4 :175-720 (Fox-Kemper et al. 2008 in general coordinates), the shape function mu :723-757 and
// mixedlayer_restrat_BML :1209-1486 (the bulk mixed layer); find_ustar (MOM_forcing_type.F90:1236) with forces%ustar, Boussinesq.
//
// One lane per column, lanes along i (every global access i-contiguous):
//   mle_column_kernel<BML>  the cells is-1 .. ie+1: the mixed layer depth (from the density difference to the surface, or h_MLD),
//                           its two running means (state of the control structure), the thickness and the mean buoyancy of the
//                           fast and the slow mixed layer -- four 2-D fields for the faces;
//   mle_face_kernel<d,BML>  the faces: the overturning timescale, the two transports, then the k loops of the reference -- the
//                           profile a(k) and the limiter of the fast transport, b(k) and the limiter of the slow one, the layer
//                           transports.  a(k), b(k) depend on the thicknesses only, so the later loops recompute them (bit for
//                           bit the same values) instead of keeping nk-long arrays per lane;
//   mle_update_kernel       the thickness tendency :690-696.
// Algorithmic traffic: h, T, S once for the columns (only down to the mixed layer base for T, S), h three times for the faces,
// uhml, vhml written and read, uhtr, vhtr, h updated: ~150 B per cell and call (one call per thermodynamic step, MOM.F90:1335).
#include <cmath>

#include "common.hpp"
#include "cr_math.hpp"
#include "eos.hpp"

namespace {

using m6::max2;
using m6::min2;
using namespace m6::eos;

// mu :723-757; x**(1+2 dh) correctly rounded (0**y = 0, x**1 = x, 1**y = 1 exactly)
__device__ __forceinline__ double mle_mu(double sigma, double dh) {
  const double s21 = 2. * sigma + 1.;
  const double mu = max2(0., (1. - s21 * s21) * (1. + (5. / 21.) * (s21 * s21)));
  const double xp = max2(0., min2(1., (-sigma - 0.5) * 2. / (1. + 2. * dh)));
  const double base = max2(1. - (xp * xp) * (3. - 2. * xp), 0.);
  const double ex = 1. + 2. * dh;
  double dd;
  if (base == 0.0) dd = 0.0;
  else if (ex == 1.0 || base == 1.0) dd = base;
  else dd = m6::cr::cr_pow(base, ex);
  const double bottop = 0.5 * (1. - copysign(1., sigma + 0.5));
  return max2(mu, dd * bottop);
}

__global__ void mle_mu_kernel(double sigma, double dh, double *out) { *out = mle_mu(sigma, dh); }

struct MLEArgs {
  m6::GridDev g;
  EosDev E;
  int nkml, use_PBL_MLD, res_upscale;
  double ml_restrat_coef, ml_restrat_coef2, I_LFront, vonKar_x_pi2, decay_time, decay_time2, density_diff, tail_dh, stretch, ustar_min, dt;
  const double *h, *T, *S, *ustar, *h_MLD, *Rd_dx_h;
  double *MLD_filtered, *MLD_filtered_slow;
  double *htot_fast, *htot_slow, *Rml_fast, *Rml_slow;      // 2-D scratch, h points
  double *hml, *htr;                                        // of the direction of the launch
};

template <bool BML> __global__ __launch_bounds__(64) void mle_column_kernel(MLEArgs A) {
  const m6::GridDev &g = A.g;
  const int i = g.isc - 1 + blockIdx.x * 64 + threadIdx.x, j = g.jsc - 1 + blockIdx.y;
  if (i > g.iec + 1) return;
  const int nz = g.nk;
  const long hpl = (long)g.nih * g.njh, c = g.h2(i, j);
  const double g_Rho0 = g.H_to_Z * g.g_Earth / g.Rho0, h_neglect = g.H_subroundoff;
  if (BML) {      // :1306-1321
    double ht = 0.0, rho_int = 0.0;
    for (int k = 0; k < A.nkml; k++) {
      const double hk = A.h[c + hpl * k];
      const double Rho_ml = eos_density(A.E, A.T[c + hpl * k], A.S[c + hpl * k], 0.0);
      rho_int = rho_int + hk * Rho_ml;
      ht = ht + hk;
    }
    A.htot_fast[c] = ht;
    A.Rml_fast[c] = (g_Rho0 * rho_int) / (ht + h_neglect);
    return;
  }
  double mld;
  if (A.density_diff > 0.) {      // :283-327
    double dK = 0.5 * A.h[c], dKm1;
    const double rhoSurf = eos_density(A.E, A.T[c], A.S[c], 0.0);
    double deltaRhoAtK = 0., deltaRhoAtKm1;
    mld = 0.;
    for (int k = 1; k < nz; k++) {
      dKm1 = dK;
      dK = dK + 0.5 * (A.h[c + hpl * k] + A.h[c + hpl * (k - 1)]);
      deltaRhoAtKm1 = deltaRhoAtK;
      deltaRhoAtK = eos_density(A.E, A.T[c + hpl * k], A.S[c + hpl * k], 0.0);
      deltaRhoAtK = deltaRhoAtK - rhoSurf;
      const double ddRho = deltaRhoAtK - deltaRhoAtKm1;
      if ((mld == 0.) && (ddRho > 0.) && (deltaRhoAtKm1 < A.density_diff) && (deltaRhoAtK >= A.density_diff)) {
        const double aFac = (A.density_diff - deltaRhoAtKm1) / ddRho;
        mld = dK * aFac + dKm1 * (1. - aFac);
      }
    }
    mld = A.stretch * mld;
    if ((mld == 0.) && (deltaRhoAtK < A.density_diff)) mld = dK;
  } else {
    mld = A.stretch * A.h_MLD[c];
  }
  if (A.decay_time > 0.) {      // :330-345
    const double aFac = A.decay_time / (A.dt + A.decay_time), bFac = A.dt / (A.dt + A.decay_time);
    const double f = max2(mld, bFac * mld + aFac * A.MLD_filtered[c]);
    A.MLD_filtered[c] = f;
    mld = f;
  }
  double mld_slow = mld;
  if (A.decay_time2 > 0.) {      // :348-367
    const double aFac = A.decay_time2 / (A.dt + A.decay_time2), bFac = A.dt / (A.dt + A.decay_time2);
    const double f = max2(mld, bFac * mld + aFac * A.MLD_filtered_slow[c]);
    A.MLD_filtered_slow[c] = f;
    mld_slow = f;
  }
  // :392-428
  double hf = 0.0, hs = 0.0, Rf = 0.0, Rs = 0.0;
  for (int k = 0; k < nz; k++) {
    if (!(hf < mld || hs < mld_slow)) break;      // (nothing below adds anything: the reference's keep_going, per column)
    const double hk = A.h[c + hpl * k];
    const double rho_ml = eos_density(A.E, A.T[c + hpl * k], A.S[c + hpl * k], 0.0);
    if (hf < mld) { const double dh = min2(hk, mld - hf); Rf = Rf + dh * rho_ml; hf = hf + dh; }
    if (hs < mld_slow) { const double dh = min2(hk, mld_slow - hs); Rs = Rs + dh * rho_ml; hs = hs + dh; }
  }
  A.htot_fast[c] = hf; A.htot_slow[c] = hs;
  A.Rml_fast[c] = -(g_Rho0 * Rf) / (hf + h_neglect);
  A.Rml_slow[c] = -(g_Rho0 * Rs) / (hs + h_neglect);
}

// :523-529 (the same lines at every face, for the fast and the slow depth)
__device__ __forceinline__ double mle_timescale(double vonKar_x_pi2, double u_star, double absf, double h_vel, double h_neglect, double coef) {
  const double mom_mixrate = vonKar_x_pi2 * (u_star * u_star) / (absf * (h_vel * h_vel) + 4.0 * (h_vel + h_neglect) * u_star);
  double timescale = 0.0625 * (absf + 2.0 * mom_mixrate) / (absf * absf + mom_mixrate * mom_mixrate);
  timescale = timescale * coef;
  return timescale;
}

template <int DIR, bool BML> __global__ __launch_bounds__(64) void mle_face_kernel(MLEArgs A) {
  const m6::GridDev &g = A.g;
  const int i = (DIR ? g.isc : g.isc - 1) + blockIdx.x * 64 + threadIdx.x, j = (DIR ? g.jsc - 1 : g.jsc) + blockIdx.y;
  if (i > g.iec) return;
  const int nz = g.nk;
  const long hpl = (long)g.nih * g.njh;
  const long c0 = g.h2(i, j), c1 = DIR ? g.h2(i, j + 1) : g.h2(i + 1, j);
  const long f2 = DIR ? g.v2(i, j) : g.u2(i, j), fpl = DIR ? (long)g.nih * (g.njh + 1) : (long)(g.nih + 1) * g.njh;
  const double h_neglect = g.H_subroundoff, I4dt = 0.25 / A.dt;
  const double aT0 = g.areaT[c0], aT1 = g.areaT[c1];
  const double u_star = max2(A.ustar_min, 0.5 * (g.Z_to_H * A.ustar[c0] + g.Z_to_H * A.ustar[c1]));
  const double absf = DIR ? 0.5 * (fabs(g.CoriolisBu[g.q2(i - 1, j)]) + fabs(g.CoriolisBu[g.q2(i, j)]))
                          : 0.5 * (fabs(g.CoriolisBu[g.q2(i, j - 1)]) + fabs(g.CoriolisBu[g.q2(i, j)]));
  // timescale * G%OBCmaskCu * G%dyCu * G%IdxCu * (...) * (h_vel**2) associates from the left (MOM_mixed_layer_restrat.F90:523-524, :538-539,
  // :1371-1372): the three metric factors multiply the timescale one after the other (found by running the reference's own module beside
  // the oracle, round 5: as a product of their own they moved 5-10 % of the face transports by an ulp)
  const double gm = DIR ? g.mask2dCv[f2] : g.mask2dCu[f2], gl = DIR ? g.dxCv[f2] : g.dyCu[f2], gi = DIR ? g.IdyCv[f2] : g.IdxCu[f2];
  auto h_avail = [&](long c, double aT, int k) { return max2(I4dt * aT * (A.h[c + hpl * k] - g.Angstrom_H), 0.0); };
  if (BML) {      // :1328-1374 / :1378-1424
    const double ht0 = A.htot_fast[c0], ht1 = A.htot_fast[c1];
    const double h_vel = 0.5 * (ht0 + ht1);
    const double timescale = mle_timescale(A.vonKar_x_pi2, u_star, absf, h_vel, h_neglect, A.ml_restrat_coef);
    double Dml = timescale * gm * gl * gi * (A.Rml_fast[c1] - A.Rml_fast[c0]) * (h_vel * h_vel);
    if (Dml == 0) {
      for (int k = 0; k < A.nkml; k++) A.hml[f2 + fpl * k] = 0.0;
    } else {
      const double I2htot = 1.0 / (ht0 + ht1 + h_neglect);
      double z_topx2 = 0.0;
      for (int k = 0; k < A.nkml; k++) {
        const double hx2 = (A.h[c0 + hpl * k] + A.h[c1 + hpl * k] + h_neglect);
        const double a = (hx2 * I2htot) * (2.0 - 4.0 * (z_topx2 + 0.5 * hx2) * I2htot);
        z_topx2 = z_topx2 + hx2;
        if (a * Dml > 0.0) {
          const double av = h_avail(c0, aT0, k);
          if (a * Dml > av) Dml = av / a;
        } else {
          const double av = h_avail(c1, aT1, k);
          if (-a * Dml > av) Dml = -av / a;
        }
      }
      z_topx2 = 0.0;
      for (int k = 0; k < A.nkml; k++) {
        const double hx2 = (A.h[c0 + hpl * k] + A.h[c1 + hpl * k] + h_neglect);
        const double a = (hx2 * I2htot) * (2.0 - 4.0 * (z_topx2 + 0.5 * hx2) * I2htot);
        z_topx2 = z_topx2 + hx2;
        const double t = a * Dml;
        A.hml[f2 + fpl * k] = t;
        A.htr[f2 + fpl * k] = A.htr[f2 + fpl * k] + t * A.dt;
      }
    }
    for (int k = A.nkml; k < nz; k++) A.hml[f2 + fpl * k] = 0.0;      // :1467-1470
    return;
  }
  const double hf0 = A.htot_fast[c0], hf1 = A.htot_fast[c1], hs0 = A.htot_slow[c0], hs1 = A.htot_slow[c1];
  double res_scaling_fac = 0.0;
  if (A.res_upscale) {
    const double dx = DIR ? g.dxCv[f2] : g.dxCu[f2], dy = DIR ? g.dyCv[f2] : g.dyCu[f2];
    res_scaling_fac = (sqrt(0.5 * (dx * dx + dy * dy)) * A.I_LFront) * min2(1., 0.5 * (A.Rd_dx_h[c0] + A.Rd_dx_h[c1]));
  }
  double h_vel = 0.5 * ((hf0 + hf1) + h_neglect);
  double timescale = mle_timescale(A.vonKar_x_pi2, u_star, absf, h_vel, h_neglect, A.ml_restrat_coef);
  if (A.res_upscale) timescale = timescale * res_scaling_fac;
  double Dml = timescale * gm * gl * gi * (A.Rml_fast[c1] - A.Rml_fast[c0]) * (h_vel * h_vel);
  h_vel = 0.5 * ((hs0 + hs1) + h_neglect);
  timescale = mle_timescale(A.vonKar_x_pi2, u_star, absf, h_vel, h_neglect, A.ml_restrat_coef2);
  if (A.res_upscale) timescale = timescale * res_scaling_fac;
  double Dml_slow = timescale * gm * gl * gi * (A.Rml_slow[c1] - A.Rml_slow[c0]) * (h_vel * h_vel);

  if (Dml + Dml_slow == 0.) {
    for (int k = 0; k < nz; k++) A.hml[f2 + fpl * k] = 0.0;
    return;
  }
  const double IhTot = 2.0 / ((hf0 + hf1) + h_neglect), IhTot_slow = 2.0 / ((hs0 + hs1) + h_neglect);
  const double tdh = A.tail_dh;
  // a(k) and the limiter of the fast transport :556-567
  double zpa = 0.0, mu_up = mle_mu(zpa, tdh);
  for (int k = 0; k < nz; k++) {
    const double hAtVel = 0.5 * (A.h[c0 + hpl * k] + A.h[c1 + hpl * k]);
    zpa = zpa - (hAtVel * IhTot);
    const double mu_dn = mle_mu(zpa, tdh);
    const double a = mu_up - mu_dn;
    mu_up = mu_dn;
    if (a * Dml > 0.0) {
      const double av = h_avail(c0, aT0, k);
      if (a * Dml > av) Dml = av / a;
    } else if (a * Dml < 0.0) {
      const double av = h_avail(c1, aT1, k);
      if (-a * Dml > av) Dml = -av / a;
    }
  }
  // b(k) and the limiter of the slow transport :568-582
  zpa = 0.0; mu_up = mle_mu(zpa, tdh);
  double zpb = 0.0, mub_up = mu_up;
  for (int k = 0; k < nz; k++) {
    const double hAtVel = 0.5 * (A.h[c0 + hpl * k] + A.h[c1 + hpl * k]);
    zpa = zpa - (hAtVel * IhTot); zpb = zpb - (hAtVel * IhTot_slow);
    const double mu_dn = mle_mu(zpa, tdh), mub_dn = mle_mu(zpb, tdh);
    const double a = mu_up - mu_dn, b = mub_up - mub_dn;
    mu_up = mu_dn; mub_up = mub_dn;
    if (b * Dml_slow > 0.0) {
      const double av = h_avail(c0, aT0, k);
      if (b * Dml_slow > av - a * Dml) Dml_slow = max2(0., av - a * Dml) / b;
    } else if (b * Dml_slow < 0.0) {
      const double av = h_avail(c1, aT1, k);
      if (-b * Dml_slow > av + a * Dml) Dml_slow = -max2(0., av + a * Dml) / b;
    }
  }
  // the layer transports :583-586
  zpa = 0.0; zpb = 0.0; mu_up = mle_mu(zpa, tdh); mub_up = mu_up;
  for (int k = 0; k < nz; k++) {
    const double hAtVel = 0.5 * (A.h[c0 + hpl * k] + A.h[c1 + hpl * k]);
    zpa = zpa - (hAtVel * IhTot); zpb = zpb - (hAtVel * IhTot_slow);
    const double mu_dn = mle_mu(zpa, tdh), mub_dn = mle_mu(zpb, tdh);
    const double a = mu_up - mu_dn, b = mub_up - mub_dn;
    mu_up = mu_dn; mub_up = mub_dn;
    const double t = a * Dml + b * Dml_slow;
    A.hml[f2 + fpl * k] = t;
    A.htr[f2 + fpl * k] = A.htr[f2 + fpl * k] + t * A.dt;
  }
}

__global__ __launch_bounds__(256) void mle_update_kernel(m6::GridDev g, const double *__restrict__ uhml, const double *__restrict__ vhml,
                                                         double *__restrict__ h, double dt, int nk_upd) {
  const int i = g.isc + blockIdx.x * 64 + threadIdx.x, j = g.jsc + blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i > g.iec || j > g.jec || k >= nk_upd) return;
  const double h_min = 0.5 * g.Angstrom_H;
  const long n = g.h3(i, j, k);
  double hn = h[n] - dt * g.IareaT[g.h2(i, j)] * ((uhml[g.u3(i, j, k)] - uhml[g.u3(i - 1, j, k)]) + (vhml[g.v3(i, j, k)] - vhml[g.v3(i, j - 1, k)]));
  if (hn < h_min) hn = h_min;
  h[n] = hn;
}

}  // namespace

extern "C" double mom6hip_mixedlayer_restrat_mu(double sigma, double dh) {
  double *d = nullptr, out = NAN;
  if (hipMalloc((void **)&d, 8) != hipSuccess) return NAN;
  hipLaunchKernelGGL(mle_mu_kernel, dim3(1), dim3(1), 0, 0, sigma, dh, d);
  if (hipMemcpy(&out, d, 8, hipMemcpyDeviceToHost) != hipSuccess) out = NAN;
  (void)hipFree(d);
  return out;
}

extern "C" int mom6hip_mixedlayer_restrat(mom6hip_ctx_t *ctx, const mom6hip_mixedlayer_restrat_cs_t *cs, double *h, double *uhtr, double *vhtr,
                                          const double *T, const double *S, const mom6hip_eos_t *eos, const double *ustar, double dt,
                                          const double *h_MLD, double *uhml, double *vhml, int32_t memspace) {
  static const char *names[8] = {"USE_BODNER23", "USE_STANLEY_ML", "non-Boussinesq mode", "(reserved)", "(reserved)", "(reserved)", "(reserved)",
                                 "(reserved)"};
  M6_REQUIRE(ctx != nullptr && cs != nullptr && cs->initialized, "mixedlayer_restrat: Module must be initialized before it is used.");
  M6_REQUIRE(h && uhtr && vhtr && ustar, "mixedlayer_restrat: null argument");
  M6_REQUIRE(memspace == MOM6HIP_MEM_HOST || memspace == MOM6HIP_MEM_DEVICE, "mixedlayer_restrat: bad memspace");
  for (int n = 0; n < 8; n++) M6_REQUIRE(!cs->unsupported[n], "mixedlayer_restrat: %s is not provided by libmom6hip", names[n]);
  const bool bml = cs->nkml > 0;
  if (bml && ((cs->nkml < 2) || (cs->ml_restrat_coef <= 0.0))) return 0;      // :1284
  M6_REQUIRE(eos && T && S, "mixedlayer_restrat: An equation of state must be used with this module.");
  M6_REQUIRE(dt > 0.0, "mixedlayer_restrat: dt must be positive");
  const m6::GridDev g = ctx->g;
  M6_REQUIRE(!bml || cs->nkml <= g.nk, "mixedlayer_restrat: nkml exceeds the number of layers");
  if (!bml) {
    M6_REQUIRE(!(cs->front_length > 0.) || cs->Rd_dx_h, "mixedlayer_restrat_OM4: The resolution argument, Rd/dx, was not associated.");
    M6_REQUIRE(cs->MLE_density_diff > 0. || cs->MLE_use_PBL_MLD, "mixedlayer_restrat_OM4: No MLD to use for MLE parameterization.");
    M6_REQUIRE(cs->MLE_density_diff > 0. || h_MLD, "mixedlayer_restrat_OM4: MLE_USE_PBL_MLD needs h_MLD");
    M6_REQUIRE(!(cs->MLE_MLD_decay_time > 0.) || cs->MLD_filtered, "mixedlayer_restrat_OM4: MLE_MLD_DECAY_TIME needs MLD_filtered");
    M6_REQUIRE(!(cs->MLE_MLD_decay_time2 > 0.) || cs->MLD_filtered_slow, "mixedlayer_restrat_OM4: MLE_MLD_DECAY_TIME2 needs MLD_filtered_slow");
  }
  M6_REQUIRE(g.isc - g.isd >= 1 && g.jsc - g.jsd >= 1, "mixedlayer_restrat: the halo must be at least 1 point wide");
  M6_REQUIRE(g.areaT && g.IareaT && g.CoriolisBu && g.dxCu && g.dyCu && g.IdxCu && g.dxCv && g.dyCv && g.IdyCv && g.mask2dCu && g.mask2dCv,
             "mixedlayer_restrat: a required grid metric is missing");
  const int nz = g.nk;
  const size_t nH = (size_t)g.nih * g.njh, nU = (size_t)(g.nih + 1) * g.njh, nV = (size_t)g.nih * (g.njh + 1);
  const size_t bH2 = 8 * nH, bH = bH2 * nz, bU = 8 * nU * nz, bV = 8 * nV * nz;
  hipStream_t s = ctx->stream;
  m6::Stager st(ctx, memspace);
  MLEArgs A;
  A.g = g;
  A.E.form = eos->form; A.E.Rho_T0_S0 = eos->Rho_T0_S0; A.E.dRho_dT = eos->dRho_dT; A.E.dRho_dS = eos->dRho_dS;
  A.nkml = cs->nkml; A.use_PBL_MLD = cs->MLE_use_PBL_MLD; A.res_upscale = (!bml && cs->front_length > 0.) ? 1 : 0;
  A.ml_restrat_coef = cs->ml_restrat_coef; A.ml_restrat_coef2 = cs->ml_restrat_coef2;
  A.I_LFront = A.res_upscale ? 1. / cs->front_length : 0.0;
  A.vonKar_x_pi2 = cs->vonKar * 9.8696;
  A.decay_time = cs->MLE_MLD_decay_time; A.decay_time2 = cs->MLE_MLD_decay_time2; A.density_diff = cs->MLE_density_diff;
  A.tail_dh = cs->MLE_tail_dh; A.stretch = cs->MLE_MLD_stretch; A.ustar_min = cs->ustar_min; A.dt = dt;
  double *d_h = st.inout(h, bH), *d_uhtr = st.inout(uhtr, bU), *d_vhtr = st.inout(vhtr, bV);
  A.h = d_h; A.T = st.in(T, bH); A.S = st.in(S, bH); A.ustar = st.in(ustar, bH2);
  A.h_MLD = (!bml && !(cs->MLE_density_diff > 0.)) ? st.in(h_MLD, bH2) : nullptr;
  A.Rd_dx_h = A.res_upscale ? st.in(cs->Rd_dx_h, bH2) : nullptr;
  A.MLD_filtered = (!bml && cs->MLE_MLD_decay_time > 0.) ? st.inout(cs->MLD_filtered, bH2) : nullptr;
  A.MLD_filtered_slow = (!bml && cs->MLE_MLD_decay_time2 > 0.) ? st.inout(cs->MLD_filtered_slow, bH2) : nullptr;
  double *d_uhml = uhml ? st.inout(uhml, bU) : (double *)st.scratch(bU), *d_vhml = vhml ? st.inout(vhml, bV) : (double *)st.scratch(bV);
  A.htot_fast = (double *)st.scratch(bH2); A.htot_slow = (double *)st.scratch(bH2);
  A.Rml_fast = (double *)st.scratch(bH2); A.Rml_slow = (double *)st.scratch(bH2);
  M6_REQUIRE(!st.failed(), "mixedlayer_restrat: staging failed");
  const int ni = g.iec - g.isc + 1, nj = g.jec - g.jsc + 1;
  const dim3 gc((ni + 2 + 63) / 64, nj + 2), gu((ni + 1 + 63) / 64, nj), gv((ni + 63) / 64, nj + 1);
  if (bml) {
    hipLaunchKernelGGL(mle_column_kernel<true>, gc, dim3(64), 0, s, A);
    A.hml = d_uhml; A.htr = d_uhtr;
    hipLaunchKernelGGL((mle_face_kernel<0, true>), gu, dim3(64), 0, s, A);
    A.hml = d_vhml; A.htr = d_vhtr;
    hipLaunchKernelGGL((mle_face_kernel<1, true>), gv, dim3(64), 0, s, A);
  } else {
    hipLaunchKernelGGL(mle_column_kernel<false>, gc, dim3(64), 0, s, A);
    A.hml = d_uhml; A.htr = d_uhtr;
    hipLaunchKernelGGL((mle_face_kernel<0, false>), gu, dim3(64), 0, s, A);
    A.hml = d_vhml; A.htr = d_vhtr;
    hipLaunchKernelGGL((mle_face_kernel<1, false>), gv, dim3(64), 0, s, A);
  }
  const int nk_upd = bml ? cs->nkml : nz;
  hipLaunchKernelGGL(mle_update_kernel, dim3((ni + 63) / 64, (nj + 3) / 4, nk_upd), dim3(64, 4), 0, s, g, d_uhml, d_vhml, d_h, dt, nk_upd);
  M6_HIP(hipGetLastError());
  return st.finish();
}
