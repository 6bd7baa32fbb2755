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

/* ---- the available potential energy of write_energy (CALCULATE_APE, :610-680) and the depth list it needs (:1109-1232) ---- */

/* create_depth_list :1109-1232 from the global lists (position (j_global-1)*niglobal + i_global, 1-based in the reference):
 * Dlist = bathyT + Z_ref, AreaList = mask2dT*areaT, mls entries each.  The heap sort is the reference's (the order of equal
 * depths decides the order in which their areas are added).  Returns listsize; depth / area / vol_below are malloc'ed. */
int orc_depth_list_create(int mls, const double *Dlist_in, const double *Area_in, double min_depth_inc, double **depth_out,
                          double **area_out, double **vol_below_out)
{
  double *Dlist = calloc((size_t)mls + 2, 8), *AreaList = calloc((size_t)mls + 2, 8);
  int *indx2 = calloc((size_t)mls + 2, sizeof(int));
  for (int q = 1; q <= mls; q++) { Dlist[q] = Dlist_in[q-1]; AreaList[q] = Area_in[q-1]; }
  for (int j = 1; j <= mls+1; j++) indx2[j] = j;
  int k = mls / 2 + 1, ir = mls;
  if (mls >= 2) for (;;) {                                       /* :1150-1169 */
    int indxt; double Dnow;
    if (k > 1) {
      k = k - 1;
      indxt = indx2[k];
      Dnow = Dlist[indxt];
    } else {
      indxt = indx2[ir];
      Dnow = Dlist[indxt];
      indx2[ir] = indx2[1];
      ir = ir - 1;
      if (ir == 1) { indx2[1] = indxt; break; }
    }
    int i = k, j = k*2;
    for (;;) {
      if (j > ir) break;
      if (j < ir && Dlist[indx2[j]] < Dlist[indx2[j+1]]) j = j + 1;
      if (Dnow < Dlist[indx2[j]]) { indx2[i] = indx2[j]; i = j; j = j + i; }
      else j = ir+1;
    }
    indx2[i] = indxt;
  }
  /* count the unique elements :1177-1186 */
  double D_list_prev = Dlist[indx2[mls]];
  int list_size = 2;
  for (k = mls-1; k >= 1; k--) {
    if (Dlist[indx2[k]] < D_list_prev-min_depth_inc) { list_size = list_size + 1; D_list_prev = Dlist[indx2[k]]; }
  }
  const int listsize = list_size+1;
  double *depth = calloc((size_t)listsize + 1, 8), *area_l = calloc((size_t)listsize + 1, 8), *vol_below = calloc((size_t)listsize + 1, 8);
  double vol = 0.0, area = 0.0;
  double Dprev = Dlist[indx2[mls]];
  D_list_prev = Dprev;
  int kl = 0;
  for (k = mls; k >= 1; k--) {                                    /* :1195-1214 */
    const int i = indx2[k];
    vol = vol + area * (Dprev - Dlist[i]);
    area = area + AreaList[i];
    int add_to_list = 0;
    if ((kl == 0) || (k==1)) add_to_list = 1;
    else if (Dlist[indx2[k-1]] < D_list_prev-min_depth_inc) { add_to_list = 1; D_list_prev = Dlist[indx2[k-1]]; }
    if (add_to_list) { kl = kl+1; depth[kl] = Dlist[i]; area_l[kl] = area; vol_below[kl] = vol; }
    Dprev = Dlist[i];
  }
  while (kl+1 < listsize) {                                       /* :1216-1222 */
    kl = kl+1;
    vol_below[kl] = vol_below[kl-1] * 1.000001;
    area_l[kl] = area_l[kl-1];
    depth[kl] = depth[kl-1];
  }
  vol_below[listsize] = vol_below[listsize-1] * 1000.0;
  area_l[listsize] = area_l[listsize-1];
  depth[listsize] = depth[listsize-1];
  /* hand back 0-based arrays of listsize entries */
  *depth_out = malloc((size_t)listsize*8); *area_out = malloc((size_t)listsize*8); *vol_below_out = malloc((size_t)listsize*8);
  memcpy(*depth_out, depth+1, (size_t)listsize*8); memcpy(*area_out, area_l+1, (size_t)listsize*8); memcpy(*vol_below_out, vol_below+1, (size_t)listsize*8);
  free(Dlist); free(AreaList); free(indx2); free(depth); free(area_l); free(vol_below);
  return listsize;
}

/* Z_0APE(K) :610-630 from the volumes of the layers: a search of the depth list from the bottom layer up.  lH [nz]: the
 * list positions remembered from the last call (CS%lH, 1-based; listsize-1 at first), updated.  DL arrays 0-based here. */
void orc_ape_reference_heights(int nz, int listsize, const double *DL_depth, const double *DL_area, const double *DL_vol_below,
                               const double *vol_lay, int *lH, double *Z_0APE /* [nz+1] */)
{
#define VB(l) DL_vol_below[(l)-1]
  int lbelow = 1, li = 1;
  double volbelow = 0.0;
  for (int k = nz; k >= 1; k--) {
    volbelow = volbelow + vol_lay[k-1];
    if ((volbelow >= VB(lH[k-1])) && (volbelow < VB(lH[k-1]+1))) {
      li = lH[k-1];
    } else {
      int labove = listsize;
      li = (labove + lbelow) / 2;
      while (li > lbelow) {
        if (volbelow < VB(li)) labove = li;
        else lbelow = li;
        li = (labove + lbelow) / 2;
      }
      lH[k-1] = li;
    }
    lbelow = li;
    Z_0APE[k-1] = DL_depth[li-1] - (volbelow - VB(li)) / DL_area[li-1];
  }
  Z_0APE[nz] = DL_depth[2-1];
#undef VB
}

/* The APE part of write_energy for one tile holding the whole domain (Boussinesq :633-645): PE [nz+1], PE_tot.  mass_lay from
 * orc_write_energy_sums; g_prime [nz+1] = GV%g_prime; lH as above (NULL: a first call). */
int orc_write_energy_ape(const mom6hip_grid_t *G, const double *h, const double *mass_lay, const double *g_prime, double Rho0,
                         double H_to_kg_m2, double Z_ref, double min_depth_inc, int *lH_io, double *PE, double *PE_tot, double *Z_0APE)
{
  const int is = G->isc, ie = G->iec, js = G->jsc, je = G->jec, nz = G->nk;
  const int ni = ie-is+1, nj = je-js+1, mls = ni*nj;
  const int nih = ORC_NIH(G), njh = ORC_NJH(G);
  double *Dl = calloc(mls, 8), *Al = calloc(mls, 8);
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
    const int pos = (j-js)*ni + (i-is);
    Dl[pos] = G->bathyT[H2(i,j)] + Z_ref;
    Al[pos] = G->mask2dT[H2(i,j)] * G->areaT[H2(i,j)];
  }
  double *Dd, *Da, *Dv;
  const int listsize = orc_depth_list_create(mls, Dl, Al, min_depth_inc, &Dd, &Da, &Dv);
  int *lH = malloc(sizeof(int)*nz);
  for (int k = 0; k < nz; k++) lH[k] = lH_io ? lH_io[k] : listsize-1;                      /* :1101-1103 */
  double *vol_lay = calloc(nz, 8);
  for (int k = 0; k < nz; k++) vol_lay[k] = ((1.0*1.0)*G->H_to_Z/H_to_kg_m2)*mass_lay[k];   /* :510 */
  orc_ape_reference_heights(nz, listsize, Dd, Da, Dv, vol_lay, lH, Z_0APE);
  if (lH_io) memcpy(lH_io, lH, sizeof(int)*nz);
  const double PE_scale_factor = 1.0;
  double *PE_pt = calloc((size_t)nih*njh*(nz+1), 8);
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
    const double areaTm = G->mask2dT[H2(i,j)]*G->areaT[H2(i,j)];
    double hbelow = 0.0;
    for (int K = nz; K >= 1; K--) {
      hbelow = hbelow + h[H3(i,j,K)] * G->H_to_Z;
      const double hint = Z_0APE[K-1] + (hbelow - (G->bathyT[H2(i,j)] + Z_ref));
      double hbot = Z_0APE[K-1] - (G->bathyT[H2(i,j)] + Z_ref);
      hbot = (hbot + fabs(hbot)) * 0.5;
      PE_pt[(size_t)nih*njh*(K-1) + H2(i,j)] = (0.5 * PE_scale_factor * areaTm) * (Rho0*g_prime[K-1]) * (hint * hint - hbot * hbot);
    }
  }
  const int rc = orc_reproducing_sum_3d(PE_pt, nih, njh, nz+1, is - G->isd, ie - G->isd, js - G->jsd, je - G->jsd, PE_tot, PE, NULL, NULL, NULL);
  free(Dl); free(Al); free(Dd); free(Da); free(Dv); free(lH); free(vol_lay); free(PE_pt);
  return rc;
}
