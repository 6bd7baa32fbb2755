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
 *   - TRIPOLAR_N (MOM_domains.F90:189; FMS FOLD_NORTH_EDGE): the northern edge is glued to itself turned by half a turn
 *     about the point (ni/2, nj): cell (i, nj+m) IS cell (ni+1-i, nj+1-m).  Hence, for m = 1..halo and with the x wrap done
 *     first, h(i, nj+m) = h(ni+1-i, nj+1-m); east faces u(I, nj+m) = s*u(ni-I, nj+1-m); north faces v(i, nj+m) =
 *     s*v(ni+1-i, nj-m); corners q(I, nj+m) = q(ni-I, nj-m); s = -1 for the components of a vector (the axes turn with
 *     the cell), +1 for a SCALAR_PAIR (MOM6HIP_PASS_SCALAR_PAIR in pos).  Rows on the fold line itself (v, q at J = nj) are
 *     computed points of both halves and are left as computed.  FMS is not vendored: this is the geometry of the fold,
 *     checked in tests/test_tripolar.py against the same operators on the unfolded (doubled) domain.
 */
#include "mom6_oracle.h"

void orc_halo_update(const mom6hip_grid_t *G, double *f, int pos_flags, int nk)
{
  const int pos = pos_flags & 3;
  const double fold_sign = ((pos == MOM6HIP_POS_U || pos == MOM6HIP_POS_V) && !(pos_flags & MOM6HIP_PASS_SCALAR_PAIR)) ? -1.0 : 1.0;
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
    if (G->tripolar_n) {
      for (int j = jce+1; j <= jhi; j++) {
        const int m = j - jce;
        const int sj = ys ? jce - m : jce + 1 - m;
        for (int i = ilo; i <= ihi; i++) {
          const int si = xs ? (G->isc + G->iec - 1 - i) : (G->isc + G->iec - i);
          if (si < ilo || si > ihi) continue;
          F(i,j,k) = (fold_sign < 0.0) ? -F(si,sj,k) : F(si,sj,k);
        }
      }
    }
  }
#undef F
}
