/*
 * domains.c -- single-tile restatement of the MOM_domains halo update (TEST INFRASTRUCTURE).
 *
 * Follows the semantics of pass_var_3d / pass_vector_3d / do_group_pass
 * (config_src/infra/FMS2/MOM_domain_infra.F90:171,660,1141), which wrap FMS mpp_update_domains
 * (FMS is not vendored in the reference; semantics as documented in SURVEY.md section 5):
 *   - data domain = compute domain + halo; an update fills E/W/N/S edges and corners;
 *   - with symmetric memory the u/q arrays carry one extra column at I = isc-1 (v/q: row at
 *     J = jsc-1) that belongs to the compute domain and is NOT a halo point;
 *   - REENTRANT_X / REENTRANT_Y wrap onto the tile itself when there is one tile in that
 *     direction; halos beyond a closed edge are not touched;
 *   - C-grid vector components change sign only across a tripolar fold (not modelled here).
 */
#include "mom6_oracle.h"

void orc_halo_update(const mom6hip_grid_t *G, double *f, int pos, int nk)
{
  const int ni = G->iec - G->isc + 1, nj = G->jec - G->jsc + 1;
  const int xs = (pos == MOM6HIP_POS_U || pos == MOM6HIP_POS_Q) ? 1 : 0; /* extra column at isd-1 */
  const int ys = (pos == MOM6HIP_POS_V || pos == MOM6HIP_POS_Q) ? 1 : 0; /* extra row at jsd-1 */
  const int ilo = G->isd - xs, ihi = G->ied;      /* allocated i range */
  const int jlo = G->jsd - ys, jhi = G->jed;
  const long nis = (long)(ihi - ilo + 1), njs = (long)(jhi - jlo + 1);
  /* compute-domain range of this staggering */
  const int ics = G->isc - xs, ice = G->iec, jcs = G->jsc - ys, jce = G->jec;
#define F(i,j,k) f[(long)((i)-ilo) + nis*((long)((j)-jlo) + njs*(long)(k))]
  for (int k = 0; k < nk; k++) {
    if (G->reentrant_x) {
      /* rows of the compute domain first; the y sweep below then carries the corners */
      for (int j = jcs; j <= jce; j++) {
        for (int i = ilo; i < ics; i++) F(i,j,k) = F(i+ni,j,k);
        for (int i = ice+1; i <= ihi; i++) F(i,j,k) = F(i-ni,j,k);
      }
    }
    if (G->reentrant_y) {
      for (int j = jlo; j < jcs; j++)
        for (int i = ilo; i <= ihi; i++) F(i,j,k) = F(i,j+nj,k);
      for (int j = jce+1; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++) F(i,j,k) = F(i,j-nj,k);
    }
  }
#undef F
}
