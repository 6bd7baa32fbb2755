// eos.hpp -- the equations of state the hot path evaluates on the device (src/equation_of_state/MOM_EOS_Wright.F90,
// MOM_EOS_linear.F90), shared by pressure_force.hip and set_viscosity.hip.  Every expression keeps the reference's
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

// density_elem :80-95
__device__ __forceinline__ double eos_density(const EosDev &E, double T, double S, double pressure) {
  if (E.form == MOM6HIP_EOS_LINEAR) return E.Rho_T0_S0 + E.dRho_dT * T + E.dRho_dS * S;
  const double al0 = (a0 + a1 * T) + a2 * S;
  const double p0 = (b0 + b4 * S) + T * (b1 + T * (b2 + b3 * T) + b5 * S);
  const double lambda = (c0 + c4 * S) + T * (c1 + T * (c2 + c3 * T) + c5 * S);
  return (pressure + p0) / (lambda + al0 * (pressure + p0));
}

// density_anomaly_elem :98-129
__device__ __forceinline__ double eos_density_anomaly(const EosDev &E, double T, double S, double pressure, double rho_ref) {
  if (E.form == MOM6HIP_EOS_LINEAR) return (E.Rho_T0_S0 - rho_ref) + (E.dRho_dT * T + E.dRho_dS * S);
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
  const double al0 = (a0 + a1 * T) + a2 * S;
  const double p0 = (b0 + b4 * S) + T * (b1 + T * ((b2 + b3 * T)) + b5 * S);
  const double lambda = (c0 + c4 * S) + T * (c1 + T * ((c2 + c3 * T)) + c5 * S);
  double I_denom2 = 1.0 / (lambda + al0 * (pressure + p0));
  I_denom2 = I_denom2 * I_denom2;
  drho_dT = I_denom2 * (lambda * (b1 + T * (2.0 * b2 + 3.0 * b3 * T) + b5 * S) -
                        (pressure + p0) * ((pressure + p0) * a1 + (c1 + T * (c2 * 2.0 + c3 * 3.0 * T) + c5 * S)));
  drho_dS = I_denom2 * (lambda * (b4 + b5 * T) - (pressure + p0) * ((pressure + p0) * a2 + (c4 + c5 * T)));
}

}  // namespace eos
}  // namespace m6
