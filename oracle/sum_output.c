/* oracle/sum_output.c -- TEST INFRASTRUCTURE: a C restatement of the global integrals of write_energy
 * (src/diagnostics/MOM_sum_output.F90:490-760; Boussinesq, CALCULATE_APE = False): mass :503-510, kinetic energy :683-689,
 * salt and heat :693-712, the maximum CFL numbers :718-744, every total through the extended-fixed-point sums of oracle/coms.c.
 * The reference holds no known-answer vectors for this routine; the sums themselves are pinned against exact arithmetic
 * (tests/test_coms.py) and the integrands are checked against numpy in tests/test_sum_output.py. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mom6_oracle.h"

int orc_write_energy_sums(const mom6hip_grid_t *G, const double *u, const double *v, const double *h, const double *T, const double *S,
                          double dt, double C_p, double H_to_kg_m2, double *mass_lay, double *KE_lay, mom6hip_energy_sums_t *out)
{
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int Isq = is-1, Ieq = ie, Jsq = js-1, Jeq = je;
  const int nih = ORC_NIH(G), njh = ORC_NJH(G);
  const size_t nH = (size_t)nih*njh;
#define H2(i,j) ORC_H2(G,i,j)
#define U2(i,j) ORC_U2(G,i,j)
#define V2(i,j) ORC_V2(G,i,j)
#define H3(i,j,k) ORC_H3(G,i,j,k)
#define U3(i,j,k) ORC_U3(G,i,j,k)
#define V3(i,j,k) ORC_V3(G,i,j,k)
  memset(out, 0, sizeof(*out));
  const double HL2_to_kg = H_to_kg_m2*(1.0*1.0);                                   /* :490 */
  double *areaTm = calloc(nH, 8), *tmp1 = calloc(nH*nz, 8), *lay = calloc(nz, 8);
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) areaTm[H2(i,j)] = G->mask2dT[H2(i,j)]*G->areaT[H2(i,j)];
  const int i0 = is - G->isd, i1 = ie - G->isd, j0 = js - G->jsd, j1 = je - G->jsd;
  int rc = 0;
  for (int k = 1; k <= nz; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++)
    tmp1[H3(i,j,k)] = h[H3(i,j,k)] * (HL2_to_kg*areaTm[H2(i,j)]);                  /* :506 */
  rc |= orc_reproducing_sum_3d(tmp1, nih, njh, nz, i0, i1, j0, j1, &out->mass_tot, mass_lay ? mass_lay : lay, out->mass_EFP, NULL, NULL);
  const double KE_scale_factor = HL2_to_kg*(1.0*1.0);                              /* :682 */
  for (int k = 1; k <= nz; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
    const int I = i, J = j;
    tmp1[H3(i,j,k)] = (0.25 * KE_scale_factor * (areaTm[H2(i,j)] * h[H3(i,j,k)])) *
            ((u[U3(I-1,j,k)]*u[U3(I-1,j,k)] + u[U3(I,j,k)]*u[U3(I,j,k)]) + (v[V3(i,J-1,k)]*v[V3(i,J-1,k)] + v[V3(i,J,k)]*v[V3(i,J,k)]));
  }
  rc |= orc_reproducing_sum_3d(tmp1, nih, njh, nz, i0, i1, j0, j1, &out->KE_tot, KE_lay ? KE_lay : lay, NULL, NULL, NULL);
  out->PE_tot = 0.0;
  out->toten = out->KE_tot + out->PE_tot;
  if (T && S) {                                                                    /* :693-712 */
    double *Salt_int = calloc(nH, 8), *Temp_int = calloc(nH, 8);
    for (int k = 1; k <= nz; k++) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      Salt_int[H2(i,j)] = Salt_int[H2(i,j)] + 1.0*S[H3(i,j,k)] * (h[H3(i,j,k)]*(HL2_to_kg * areaTm[H2(i,j)]));
      Temp_int[H2(i,j)] = Temp_int[H2(i,j)] + (1.0*C_p * T[H3(i,j,k)]) * (h[H3(i,j,k)]*(HL2_to_kg * areaTm[H2(i,j)]));
    }
    /* reproducing_sum_EFP(..., only_on_PE) then EFP_sum_across_PEs then EFP_to_real: on one PE the regularised total */
    rc |= orc_reproducing_sum_3d(Salt_int, nih, njh, 1, i0, i1, j0, j1, &out->Salt, NULL, out->salt_EFP, NULL, NULL);
    rc |= orc_reproducing_sum_3d(Temp_int, nih, njh, 1, i0, i1, j0, j1, &out->Heat, NULL, out->heat_EFP, NULL, NULL);
    free(Salt_int); free(Temp_int);
  }
  double max_CFL[2] = {0.0, 0.0};                                                   /* :718-740 */
  for (int k = 1; k <= nz; k++) for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++) {
    const int i = I;
    double CFL_Iarea = G->IareaT[H2(i,j)];
    if (u[U3(I,j,k)] < 0.0) CFL_Iarea = G->IareaT[H2(i+1,j)];
    const double CFL_trans = fabs(u[U3(I,j,k)] * dt) * (G->dy_Cu[U2(I,j)] * CFL_Iarea);
    const double CFL_lin = fabs(u[U3(I,j,k)] * dt) * G->IdxCu[U2(I,j)];
    if (CFL_trans > max_CFL[0]) max_CFL[0] = CFL_trans;
    if (CFL_lin > max_CFL[1]) max_CFL[1] = CFL_lin;
  }
  for (int k = 1; k <= nz; k++) for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++) {
    const int j = J;
    double CFL_Iarea = G->IareaT[H2(i,j)];
    if (v[V3(i,J,k)] < 0.0) CFL_Iarea = G->IareaT[H2(i,j+1)];
    const double CFL_trans = fabs(v[V3(i,J,k)] * dt) * (G->dx_Cv[V2(i,J)] * CFL_Iarea);
    const double CFL_lin = fabs(v[V3(i,J,k)] * dt) * G->IdyCv[V2(i,J)];
    if (CFL_trans > max_CFL[0]) max_CFL[0] = CFL_trans;
    if (CFL_lin > max_CFL[1]) max_CFL[1] = CFL_lin;
  }
  out->max_CFL[0] = max_CFL[0]; out->max_CFL[1] = max_CFL[1];
  out->npoints = (int64_t)(ie-is+1)*(je-js+1)*nz;
  free(areaTm); free(tmp1); free(lay);
  return rc;
}
