"""Calibration of the CPU port's whole step (bench.py's cpu_baseline, kind "port") against the reference's own dynamical core: the unmodified
MOM_dynamics_split_RK2.F90 with every module it steps through, compiled in place with amdflang -O2 against the stand-ins of tests/fortran/stubs
(tests/test_reference_kernels.py::build_ref_dyn_driver has the recipe and the bitwise check at -O0), timed over the same steps as the oracle's
DynState.step (gcc -O2), one thread each, on a closed basin with the settings of bench.py's step.  Seconds per step from two runs of
different length (the difference removes start-up, I/O and initialisation).  Build container only.
usage: python tools/calibrate_ref_core.py [NIxNJxNK] [steps]  ->  one JSON line (profiles/r05_calibrate_ref_core.json)"""
import json, os, resource, subprocess, sys, tempfile, time, pathlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import test_reference_kernels as trk
import test_testing_configs as tc

ni, nj, nk = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "120x80x20").split("x")]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
name = "bench_like"
tc.TC_INPUT[name] = dict(shape=(ni, nj, nk), pairs=trk.BENCH_LIKE["pairs"] + "\n        REENTRANT_X = False\n")
out = {"grid": [ni, nj, nk], "steps": nsteps, "parameters": "the settings of bench.py's step (tests/test_reference_kernels.py BENCH_LIKE), closed basin",
       "builds": "reference: amdflang -cpp -fdefault-real-8 -O2 -ffp-contract=off, 19 source files of the dynamical core in place, stand-ins "
                 "tests/fortran/stubs; port: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math (oracle/Makefile); one thread each"}
with tempfile.TemporaryDirectory() as td:
    td = pathlib.Path(td)
    exe = trk.build_ref_dyn_driver(td, opt="-O2")
    state = tc.case_state(name)
    g, d, taux, tauy, ustar, bbl, Rlay, g_prime = state
    unlimited = lambda: resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    wall = {}
    for n in (1, 1 + nsteps):
        tc.write_case(td, name, n, False, state, bbl_mode=1)
        t0 = time.perf_counter()
        r = subprocess.run([exe, str(td / "in.bin"), str(td / f"out{n}.bin"), str(td / "params.txt")], capture_output=True, text=True,
                           preexec_fn=unlimited, env=dict(os.environ, OMP_NUM_THREADS="1"))
        wall[n] = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-2000:]
    ref_s = (wall[1 + nsteps] - wall[1]) / nsteps
    # the -O2 build's fields against the oracle's (they were bitwise equal at -O0; at -O2 the compiler may reassociate nothing either)
    st, calc, _ = tc.oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
    st.bbl(); st.step(taux, tauy, calc_dtbt=calc(0))      # (the first step carries the initialisation's work)
    t0 = time.perf_counter()
    for n in range(1, 1 + nsteps):
        st.bbl(); st.step(taux, tauy, calc_dtbt=calc(n))
    port_s = (time.perf_counter() - t0) / nsteps
    got = tc.read_out(str(td / f"out{1 + nsteps}.bin"), g, meke=False)
    want = dict(u=st.u, v=st.v, h=st.h, uh=st.uh, vh=st.vh, uhtr=st.uhtr, vhtr=st.vhtr, eta_av=st.eta_av)
    # the -O2 build's fields against the oracle's after all the steps: bitwise equal at the test's sizes (and -O0); on this grid the one
    # libm power of btstep (bt_rem = av_rem ** (1/nstep), DESIGN.md section 3: 1 ulp from the correctly rounded one in ~0.1 % of arguments)
    # reaches a few faces a step and what follows them
    # ... and with the oracle taking the host's libm pow there (ORC_BT_LIBM_POW), all the steps again
    os.environ["ORC_BT_LIBM_POW"] = "1"
    st2, calc2, _ = tc.oracle_for(name, g, d, ustar, bbl, Rlay, g_prime)
    for n in range(1 + nsteps):
        st2.bbl(); st2.step(taux, tauy, calc_dtbt=calc2(n))
    del os.environ["ORC_BT_LIBM_POW"]
    want2 = dict(u=st2.u, v=st2.v, h=st2.h, uh=st2.uh, vh=st2.vh, uhtr=st2.uhtr, vhtr=st2.vhtr, eta_av=st2.eta_av)
    out["reference_O2_against_oracle_with_libm_pow"] = {"bitwise_equal": bool(all(
        trk.bits_equal(trk.interior(g, got[n], pos), trk.interior(g, want2[n], pos)) for n, pos, nd in tc.OUT if n in want2))}
    same, frac, rel = True, 0.0, 0.0
    for n, pos, nd in tc.OUT:
        if n in want:
            a, b = trk.interior(g, got[n], pos), trk.interior(g, want[n], pos)
            d = a != b
            same = same and not d.any()
            frac = max(frac, float(d.mean()))
            if d.any():
                rel = max(rel, float((np.abs(a - b)[d] / np.maximum(np.abs(b)[d], 1e-300)).max()))
    out["reference_O2_against_oracle"] = {"bitwise_equal": bool(same), "largest_fraction_of_points_differing": round(frac, 4),
                                          "largest_relative_difference": rel,
                                          "why": "libm pow in bt_rem (the declared deviation); bitwise at the sizes of tests/test_reference_kernels.py"}
out["seconds_per_step"] = {"reference_1thr": round(ref_s, 4), "port_1thr": round(port_s, 4)}
out["port_over_reference_1thr"] = round(port_s / ref_s, 3)
out["ns_per_gridpoint_step"] = {"reference_1thr": round(ref_s * 1e9 / (ni * nj * nk), 1), "port_1thr": round(port_s * 1e9 / (ni * nj * nk), 1)}
out["note"] = ("step_MOM_dyn_split_RK2 + set_viscous_BBL per step, no tracer advection / ALE; the reference files are compiled unmodified against "
               "hand-written stand-ins for the infrastructure (FMS is not vendored): a calibration of the port, not a reference build")
print(json.dumps(out))
