// eos.hpp -- the equations of state the hot path evaluates on the device (src/equation_of_state/MOM_EOS_Wright.F90,
// MOM_EOS_UNESCO.F90, MOM_EOS_linear.F90), shared by pressure_force.hip and set_viscosity.hip.  Every expression keeps the reference's
// parenthesisation (the library is built with -ffp-contract=off).
#pragma once

#include "common.hpp"

namespace m6 {
namespace eos {

struct EosDev { int form; double Rho_T0_S0, dRho_dT, dRho_dS; };

// Wright 1997 coefficients, MOM_EOS_Wright.F90:23-38
constexpr double a0 = 7.057924e-4, a1 = 3.480336e-7, a2 = -1.112733e-7;
constexpr double b0 = 5.790749e8, b1 = 3.516535e6, b2 = -4.002714e4, b3 = 2.084372e2, b4 = 5.944068e5, b5 = -9.643486e3;
constexpr double c0 = 1.704853e5, c1 = 7.904722e2, c2 = -7.984422, c3 = 5.140652e-2, c4 = -2.302158e2, c5 = -3.079464;

// UNESCO (Jackett & McDougall 1995) coefficients, MOM_EOS_UNESCO.F90:14-66
namespace unesco {
constexpr double R00 = 999.842594, R01 = 6.793952e-2, R02 = -9.095290e-3, R03 = 1.001685e-4, R04 = -1.120083e-6, R05 = 6.536332e-9,
    R10 = 0.824493, R11 = -4.0899e-3, R12 = 7.6438e-5, R13 = -8.2467e-7, R14 = 5.3875e-9, R60 = -5.72466e-3, R61 = 1.0227e-4,
    R62 = -1.6546e-6, R20 = 4.8314e-4,
    S000 = 1.965933e4, S010 = 1.444304e2, S020 = -1.706103, S030 = 9.648704e-3, S040 = -4.190253e-5, S100 = 52.84855,
    S110 = -3.101089e-1, S120 = 6.283263e-3, S130 = -5.084188e-5, S600 = 3.886640e-1, S610 = 9.085835e-3, S620 = -4.619924e-4,
    S001 = 3.186519, S011 = 2.212276e-2, S021 = -2.984642e-4, S031 = 1.956415e-6, S101 = 6.704388e-3, S111 = -1.847318e-4,
    S121 = 2.059331e-7, S601 = 1.480266e-4,
    S002 = 2.102898e-4, S012 = -1.202016e-5, S022 = 1.394680e-7, S102 = -2.040237e-6, S112 = 6.128773e-8, S122 = 6.207323e-10;
__device__ __forceinline__ double density(double T, double S, double pressure) {      // density_elem :95-128
  const double p1 = pressure*1.0e-5, t1 = T;
  const double s1 = (S > 0.0 ? S : 0.0), s12 = sqrt(s1);
  const double sig0 = ( t1*(R01 + t1*(R02 + t1*(R03 + t1*(R04 + t1*R05)))) +
           s1*((R10 + t1*(R11 + t1*(R12 + t1*(R13 + t1*R14)))) +
               (s12*(R60 + t1*(R61 + t1*R62)) + s1*R20)) );
  const double rho0 = R00 + sig0;
  const double ks = (S000 + ( t1*(S010 + t1*(S020 + t1*(S030 + t1*S040))) +
                 s1*((S100 + t1*(S110 + t1*(S120 + t1*S130))) + s12*(S600 + t1*(S610 + t1*S620))) )) +
       p1*( (S001 + ( t1*(S011 + t1*(S021 + t1*S031)) +
                      s1*((S101 + t1*(S111 + t1*S121)) + s12*S601) )) +
            p1*(S002 + ( t1*(S012 + t1*S022) + s1*(S102 + t1*(S112 + t1*S122)) )) );
  return rho0*ks / (ks - p1);
}
__device__ __forceinline__ double density_anomaly(double T, double S, double pressure, double rho_ref) {      // :133-167
  const double p1 = pressure*1.0e-5, t1 = T;
  const double s1 = (S > 0.0 ? S : 0.0), s12 = sqrt(s1);
  const double sig0 = ( t1*(R01 + t1*(R02 + t1*(R03 + t1*(R04 + t1*R05)))) +
           s1*((R10 + t1*(R11 + t1*(R12 + t1*(R13 + t1*R14)))) +
               (s12*(R60 + t1*(R61 + t1*R62)) + s1*R20)) );
  const double ks = (S000 + ( t1*(S010 + t1*(S020 + t1*(S030 + t1*S040))) +
                 s1*((S100 + t1*(S110 + t1*(S120 + t1*S130))) + s12*(S600 + t1*(S610 + t1*S620))) )) +
       p1*( (S001 + ( t1*(S011 + t1*(S021 + t1*S031)) +
                      s1*((S101 + t1*(S111 + t1*S121)) + s12*S601) )) +
            p1*(S002 + ( t1*(S012 + t1*S022) + s1*(S102 + t1*(S112 + t1*S122)) )) );
  return ((R00 - rho_ref)*ks + (sig0*ks + p1*rho_ref)) / (ks - p1);
}
__device__ __forceinline__ void density_derivs(double T, double S, double pressure, double &drho_dT, double &drho_dS) {      // :236-296
  const double p1 = pressure*1.0e-5, t1 = T;
  const double s1 = (S > 0.0 ? S : 0.0), s12 = sqrt(s1);
  const double rho0 = R00 + ( t1*(R01 + t1*(R02 + t1*(R03 + t1*(R04 + t1*R05)))) +
                 s1*((R10 + t1*(R11 + t1*(R12 + t1*(R13 + t1*R14)))) +
                     (s12*(R60 + t1*(R61 + t1*R62)) + s1*R20)) );
  const double drho0_dT = R01 + ( t1*(2.0*R02 + t1*(3.0*R03 + t1*(4.0*R04 + t1*(5.0*R05)))) +
                     s1*(R11 + (t1*(2.0*R12 + t1*(3.0*R13 + t1*(4.0*R14))) +
                                s12*(R61 + t1*(2.0*R62)) )) );
  const double drho0_dS = R10 + ( t1*(R11 + t1*(R12 + t1*(R13 + t1*R14))) +
                     (1.5*(s12*(R60 + t1*(R61 + t1*R62))) + s1*(2.0*R20)) );
  const double ks = ( S000 + (t1*(S010 + t1*(S020 + t1*(S030 + t1*S040))) +
                 s1*((S100 + t1*(S110 + t1*(S120 + t1*S130))) + s12*(S600 + t1*(S610 + t1*S620)))) ) +
       p1*( (S001 + ( t1*(S011 + t1*(S021 + t1*S031)) +
                      s1*((S101 + t1*(S111 + t1*S121)) + s12*S601) )) +
            p1*(S002 + ( t1*(S012 + t1*S022) + s1*(S102 + t1*(S112 + t1*S122)) )) );
  const double dks_dT = ( S010 + (t1*(2.0*S020 + t1*(3.0*S030 + t1*(4.0*S040))) +
                     s1*((S110 + t1*(2.0*S120 + t1*(3.0*S130))) + s12*(S610 + t1*(2.0*S620)))) ) +
           p1*(((S011 + t1*(2.0*S021 + t1*(3.0*S031))) + s1*(S111 + t1*(2.0*S121)) ) +
               p1*(S012 + t1*(2.0*S022) + s1*(S112 + t1*(2.0*S122))) );
  const double dks_dS = ( S100 + (t1*(S110 + t1*(S120 + t1*S130)) + 1.5*(s12*(S600 + t1*(S610 + t1*S620)))) ) +
           p1*((S101 + t1*(S111 + t1*S121) + s12*(1.5*S601)) +
               p1*(S102 + t1*(S112 + t1*S122)) );
  const double I_denom = 1.0 / (ks - p1);
  drho_dT = (ks*drho0_dT - dks_dT*((rho0*p1)*I_denom)) * I_denom;
  drho_dS = (ks*drho0_dS - dks_dS*((rho0*p1)*I_denom)) * I_denom;
}
}  // namespace unesco

// WRIGHT_FULL and WRIGHT_REDUCED (MOM_EOS_Wright_full.F90, MOM_EOS_Wright_red.F90: one code, two sets of coefficients)
struct WrightCoefs { double a0, a1, a2, b0, b1, b2, b3, b4, b5, c0, c1, c2, c3, c4, c5; };
__device__ __forceinline__ WrightCoefs wright_coefs(int form) {
  if (form == MOM6HIP_EOS_WRIGHT_FULL) return WrightCoefs{7.133718e-4, 2.724670e-7, -1.646582e-7, 5.613770e8, 3.600337e6, -3.727194e4, 1.660557e2, 6.844158e5, -8.389457e3, 1.609893e5, 8.427815e2, -6.931554, 3.869318e-2, -1.664201e2, -2.765195};
  return WrightCoefs{7.057924e-4, 3.480336e-7, -1.112733e-7, 5.790749e8, 3.516535e6, -4.002714e4, 2.084372e2, 5.944068e5, -9.643486e3, 1.704853e5, 7.904722e2, -7.984422, 5.140652e-2, -2.302158e2, -3.079464};
}
__device__ __forceinline__ double wrightx_density(const WrightCoefs W, double T, double S, double pressure) {      // :73-89
  const double al0 = W.a0 + (W.a1*T + W.a2*S);
  const double p0 = W.b0 + ( W.b4*S + T * (W.b1 + (T*(W.b2 + W.b3*T) + W.b5*S)) );
  const double lambda = W.c0 + ( W.c4*S + T * (W.c1 + (T*(W.c2 + W.c3*T) + W.c5*S)) );
  return (pressure + p0) / (lambda + al0*(pressure + p0));
}
__device__ __forceinline__ double wrightx_density_anomaly(const WrightCoefs W, double T, double S, double pressure, double rho_ref) {      // :94-119
  const double pa_000 = W.b0*(1.0 - W.a0*rho_ref) - rho_ref*W.c0;
  const double al_TS = W.a1*T + W.a2*S;
  const double al0 = W.a0 + al_TS;
  const double p_TSp = pressure + (W.b4*S + T * (W.b1 + (T*(W.b2 + W.b3*T) + W.b5*S)));
  const double lam_TS = W.c4*S + T * (W.c1 + (T*(W.c2 + W.c3*T) + W.c5*S));
  return (pa_000 + (p_TSp - rho_ref*(p_TSp*al0 + (W.b0*al_TS + lam_TS)))) / ( (W.c0 + lam_TS) + al0*(W.b0 + p_TSp) );
}
__device__ __forceinline__ double wrightx_spv_anomaly(const WrightCoefs W, double T, double S, double pressure, double spv_ref) {      // :150-170
  const double lam_000 = W.c0 + (W.a0 - spv_ref)*W.b0;
  const double al_TS = W.a1*T + W.a2*S;
  const double p_TSp = pressure + (W.b4*S + T * (W.b1 + (T*(W.b2 + W.b3*T) + W.b5*S)));
  const double lambda = lam_000 + ( W.c4*S + T * (W.c1 + (T*(W.c2 + W.c3*T) + W.c5*S)) );
  return al_TS + (lambda + (W.a0 - spv_ref)*p_TSp) / (W.b0 + p_TSp);
}
__device__ __forceinline__ void wrightx_density_derivs(const WrightCoefs W, double T, double S, double pressure, double &DT, double &DS) {      // :175-200
  const double al0 = W.a0 + (W.a1*T + W.a2*S);
  const double p0 = W.b0 + ( W.b4*S + T * (W.b1 + (T*(W.b2 + W.b3*T) + W.b5*S)) );
  const double lambda = W.c0 + ( W.c4*S + T * (W.c1 + (T*(W.c2 + W.c3*T) + W.c5*S)) );
  const double den = (lambda + al0*(pressure + p0));
  const double I_denom2 = 1.0 / (den*den);
  DT = I_denom2 * (lambda * (W.b1 + (T*(2.0*W.b2 + 3.0*W.b3*T) + W.b5*S)) -
     (pressure+p0) * ( (pressure+p0)*W.a1 + (W.c1 + (T*(W.c2*2.0 + W.c3*3.0*T) + W.c5*S)) ));
  DS = I_denom2 * (lambda * (W.b4 + W.b5*T) -
     (pressure+p0) * ( (pressure+p0)*W.a2 + (W.c4 + W.c5*T) ));
}
__device__ __forceinline__ bool is_wrightx(int form) { return form == MOM6HIP_EOS_WRIGHT_FULL || form == MOM6HIP_EOS_WRIGHT_REDUCED; }

// density_elem :80-95
__device__ __forceinline__ double eos_density(const EosDev &E, double T, double S, double pressure) {
  if (E.form == MOM6HIP_EOS_LINEAR) return E.Rho_T0_S0 + E.dRho_dT * T + E.dRho_dS * S;
  if (E.form == MOM6HIP_EOS_UNESCO) return unesco::density(T, S, pressure);
  if (is_wrightx(E.form)) return wrightx_density(wright_coefs(E.form), T, S, pressure);
  const double al0 = (a0 + a1 * T) + a2 * S;
  const double p0 = (b0 + b4 * S) + T * (b1 + T * (b2 + b3 * T) + b5 * S);
  const double lambda = (c0 + c4 * S) + T * (c1 + T * (c2 + c3 * T) + c5 * S);
  return (pressure + p0) / (lambda + al0 * (pressure + p0));
}

// density_anomaly_elem :98-129
__device__ __forceinline__ double eos_density_anomaly(const EosDev &E, double T, double S, double pressure, double rho_ref) {
  if (E.form == MOM6HIP_EOS_LINEAR) return (E.Rho_T0_S0 - rho_ref) + (E.dRho_dT * T + E.dRho_dS * S);
  if (E.form == MOM6HIP_EOS_UNESCO) return unesco::density_anomaly(T, S, pressure, rho_ref);
  if (is_wrightx(E.form)) return wrightx_density_anomaly(wright_coefs(E.form), T, S, pressure, rho_ref);
  const double pa_000 = (b0 * (1.0 - a0 * rho_ref) - rho_ref * c0);
  const double al_TS = a1 * T + a2 * S;
  const double al0 = a0 + al_TS;
  const double p_TSp = pressure + (b4 * S + T * (b1 + (T * (b2 + b3 * T) + b5 * S)));
  const double lam_TS = c4 * S + T * (c1 + (T * (c2 + c3 * T) + c5 * S));
  return (pa_000 + (p_TSp - rho_ref * (p_TSp * al0 + (b0 * al_TS + lam_TS)))) / ((c0 + lam_TS) + al0 * (b0 + p_TSp));
}

// calculate_density_derivs_elem :178-206
__device__ __forceinline__ void eos_density_derivs(const EosDev &E, double T, double S, double pressure, double &drho_dT,
                                                   double &drho_dS) {
  if (E.form == MOM6HIP_EOS_LINEAR) { drho_dT = E.dRho_dT; drho_dS = E.dRho_dS; return; }
  if (E.form == MOM6HIP_EOS_UNESCO) { unesco::density_derivs(T, S, pressure, drho_dT, drho_dS); return; }
  if (is_wrightx(E.form)) { wrightx_density_derivs(wright_coefs(E.form), T, S, pressure, drho_dT, drho_dS); return; }
  const double al0 = (a0 + a1 * T) + a2 * S;
  const double p0 = (b0 + b4 * S) + T * (b1 + T * ((b2 + b3 * T)) + b5 * S);
  const double lambda = (c0 + c4 * S) + T * (c1 + T * ((c2 + c3 * T)) + c5 * S);
  double I_denom2 = 1.0 / (lambda + al0 * (pressure + p0));
  I_denom2 = I_denom2 * I_denom2;
  drho_dT = I_denom2 * (lambda * (b1 + T * (2.0 * b2 + 3.0 * b3 * T) + b5 * S) -
                        (pressure + p0) * ((pressure + p0) * a1 + (c1 + T * (c2 * 2.0 + c3 * 3.0 * T) + c5 * S)));
  drho_dS = I_denom2 * (lambda * (b4 + b5 * T) - (pressure + p0) * ((pressure + p0) * a2 + (c4 + c5 * T)));
}

// ---- specific-volume anomaly (calculate_spec_vol with spv_ref): MOM_EOS_Wright.F90:156-174, MOM_EOS_UNESCO.F90:208-240,
// MOM_EOS_linear.F90:102-113
__device__ __forceinline__ double eos_spec_vol_anomaly(const EosDev &E, double T, double S, double pressure, double spv_ref) {
  if (E.form == MOM6HIP_EOS_LINEAR)
    return ((1.0 - E.Rho_T0_S0*spv_ref) - spv_ref*(E.dRho_dT*T + E.dRho_dS*S)) / (E.Rho_T0_S0 + (E.dRho_dT*T + E.dRho_dS*S));
  if (is_wrightx(E.form)) return wrightx_spv_anomaly(wright_coefs(E.form), T, S, pressure, spv_ref);
  if (E.form == MOM6HIP_EOS_UNESCO) {
    using namespace unesco;
    const double p1 = pressure*1.0e-5, t1 = T;
    const double s1 = (S > 0.0 ? S : 0.0), s12 = sqrt(s1);
    const double rho0 = R00 + ( t1*(R01 + t1*(R02 + t1*(R03 + t1*(R04 + t1*R05)))) +
                   s1*((R10 + t1*(R11 + t1*(R12 + t1*(R13 + t1*R14)))) +
                       (s12*(R60 + t1*(R61 + t1*R62)) + s1*R20)) );
    const double ks = (S000 + ( t1*(S010 + t1*(S020 + t1*(S030 + t1*S040))) +
                   s1*((S100 + t1*(S110 + t1*(S120 + t1*S130))) + s12*(S600 + t1*(S610 + t1*S620))) )) +
         p1*( (S001 + ( t1*(S011 + t1*(S021 + t1*S031)) +
                        s1*((S101 + t1*(S111 + t1*S121)) + s12*S601) )) +
              p1*(S002 + ( t1*(S012 + t1*S022) + s1*(S102 + t1*(S112 + t1*S122)) )) );
    return (ks*(1.0 - (rho0*spv_ref)) - p1) / (rho0*ks);
  }
  const double al0 = (a0 + a1*T) + a2*S;
  const double p0 = (b0 + b4*S) + T * (b1 + T*((b2 + b3*T)) + b5*S);
  const double lambda = (c0 + c4*S) + T * (c1 + T*((c2 + c3*T)) + c5*S);
  return (lambda + (al0 - spv_ref)*(pressure + p0)) / (pressure + p0);
}

}  // namespace eos
}  // namespace m6
