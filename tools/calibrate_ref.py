"""How fast the oracle's C restatement runs against the reference's own Fortran where the reference compiles with no stand-ins
(oracle/_ref: PLM_reconstruction of src/ALE/PLM_functions.F90, amdflang -O0 as built by oracle/build_ref.sh, and -O2 on request):
the calibration BASELINE.md section 3 asks for, as far as it can be had in this container.  Build container only.
usage: python tools/calibrate_ref.py [ncol] [n]"""
import ctypes as C, json, os, sys, time; sys.path.insert(0, '.')
import numpy as np
from oracle import orc

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 75
R = orc.ref_lib()
if R is None or not hasattr(R, "ref_plm_batch"):
    raise SystemExit("oracle/_ref not built (bash oracle/build_ref.sh)")
L = orc._remap_lib()
dp = C.POINTER(C.c_double)
for f in (R.ref_plm_batch, L.orc_plm_batch):
    f.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, C.c_double]; f.restype = None
rng = np.random.default_rng(1)
h = rng.random((ncol, n)) * 50.0; h[rng.random((ncol, n)) < 0.05] = 0.0
u = rng.standard_normal((ncol, n))
P = lambda a: a.ctypes.data_as(dp)
Er, cr = np.zeros((ncol, 2, n)), np.zeros((ncol, 2, n))
Eo, co = np.zeros((ncol, 2, n)), np.zeros((ncol, 3, n))
def T(f, *a):
    f(*a); t0 = time.perf_counter(); f(*a); f(*a); return (time.perf_counter() - t0) / 2
tr = T(R.ref_plm_batch, ncol, n, P(h), P(u), P(Er), P(cr), 1e-30)
R2 = C.CDLL(os.path.join(os.path.dirname(orc.__file__), "_ref", "libmom6ref_O2.so"))
R2.ref_plm_batch.argtypes = R.ref_plm_batch.argtypes; R2.ref_plm_batch.restype = None
E2, c2 = np.zeros((ncol, 2, n)), np.zeros((ncol, 2, n))
tr2 = T(R2.ref_plm_batch, ncol, n, P(h), P(u), P(E2), P(c2), 1e-30)
same2 = bool(np.array_equal(E2.view(np.uint64), Er.view(np.uint64)) and np.array_equal(c2.view(np.uint64), cr.view(np.uint64)))
to = T(L.orc_plm_batch, ncol, n, P(h), P(u), P(Eo), P(co), 1e-30)
same = bool(np.array_equal(Er.view(np.uint64), Eo.view(np.uint64)) and np.array_equal(cr.view(np.uint64), co[:, :2].view(np.uint64)))
print(json.dumps({"routine": "PLM_reconstruction", "columns": ncol, "layers": n, "reference_flang_O0_s": tr, "reference_flang_O2_s": tr2,
                  "restatement_gcc_O2_s": to, "restatement_over_reference_O2": to / tr2, "restatement_equals_reference_O0_bitwise": same,
                  "reference_O2_equals_O0_bitwise": same2,
                  "builds": "reference: amdflang -fdefault-real-8 -ffp-contract=off at -O0 and -O2 (oracle/build_ref.sh); restatement: gcc -O2 "
                            "-std=c99 -ffp-contract=off -fno-fast-math; one core"}))
