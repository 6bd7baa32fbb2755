"""Per-kernel calibration of the CPU port (bench.py's cpu_baseline, kind "port") against the reference's own kernels: the unmodified
MOM_continuity_PPM.F90, MOM_CoriolisAdv.F90 and MOM_tracer_advect.F90 compiled in place with amdflang -O2 against the stand-ins of
tests/fortran/stubs (tests/test_reference_kernels.py has the recipe and the bitwise check at -O0), timed on the same inputs as the oracle's
C restatement (gcc -O2), at 1 thread and at NT threads (-fopenmp: the loops the reference marks !$OMP; the port's OpenMP build marks the
same loops).  SURVEY.md section 8d(2) / BASELINE.md section 3.3.  Build container only.
usage: python tools/calibrate_ref_kernels.py [NIxNJxNK] [threads] [repetitions]  ->  one JSON line (profiles/r05_calibrate_ref.json)"""
import json, os, resource, subprocess, sys, tempfile, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import orc
import test_reference_kernels as trk

ni, nj, nk = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "240x160x20").split("x")]
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
out = {"grid": [ni, nj, nk], "threads": nt, "repetitions": nrep, "kernels": {},
       "builds": "reference: amdflang -cpp -fdefault-real-8 -O2 -ffp-contract=off [-fopenmp], sources in place, stand-ins tests/fortran/stubs; "
                 "port: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math [-fopenmp] (oracle/Makefile)"}
with tempfile.TemporaryDirectory() as td:
    g, names, want, inp = trk.write_case(os.path.join(td, "in.bin"), ni, nj, nk, scheme="PPM:H3", seed=91, ntr=4, land_frac=0.25)
    ref = {}
    for label, omp, threads in (("1", False, 1), (str(nt), True, nt)):
        bd = os.path.join(td, "b" + label); os.makedirs(bd)
        exe = trk.build_ref_kernels(bd, opt="-O2", openmp=omp)
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_STACKSIZE="2G")
        unlimited = lambda: resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))      # (the reference's automatic arrays)
        r = subprocess.run([exe, os.path.join(td, "in.bin"), os.path.join(td, "out" + label + ".bin"), str(nrep)], capture_output=True, text=True, env=env,
                           preexec_fn=unlimited)
        assert r.returncode == 0, r.stderr[-2000:]
        ref[label] = {ln.split()[1]: float(ln.split()[2]) for ln in r.stdout.splitlines() if ln.startswith("time ")}
        raw = np.fromfile(os.path.join(td, "out" + label + ".bin"), dtype="<f8")
        sizes = [w.size for w in want]
        same = all(trk.bits_equal(trk.interior(g, a.reshape(w.shape), trk.position_of(n)), trk.interior(g, w, trk.position_of(n)))
                   for n, a, w in zip(names, np.split(raw, np.cumsum(sizes)[:-1]), want))
        out[f"reference_O2_{label}_threads_equals_oracle_bitwise"] = bool(same)
    d, adv = inp["d"], inp["adv"]
    ccs = orc.continuity_cs(nk, g.Angstrom_H)

    def port_times(threads):
        orc.set_threads(threads)
        arrs, bt = orc.make_bt_cont(g, with_h=True)
        hp2 = d["h"].copy(); uh2 = np.zeros_like(d["u"]); vh2 = np.zeros_like(d["v"]); uc = np.zeros_like(d["u"]); vc = np.zeros_like(d["v"])
        def cont():
            orc.continuity(g, ccs, d["u"], d["v"], d["h"], hp2, uh2, vh2, inp["dt"], uhbt=inp["uhbt"], vhbt=inp["vhbt"], visc_rem_u=inp["vru"],
                           visc_rem_v=inp["vrv"], u_cor=uc, v_cor=vc, bt_cont=bt)
        def corad():
            orc.coradcalc(g, d["u"], d["v"], d["h"], uh2, vh2, bound_coriolis=True)
        def advect():
            tr = [t.copy() for t in adv["tr"]]
            orc.advect_tracer(g, adv["h_end"], adv["uhtr"], adv["vhtr"], inp["dt_adv"], 900.0, "PPM:H3", tr)
        res = {}
        for name, f in (("continuity_PPM", cont), ("CorAdCalc", corad), ("advect_tracer", advect)):
            f(); t0 = time.perf_counter()
            for _ in range(nrep):
                f()
            res[name] = (time.perf_counter() - t0) / nrep
        orc.set_threads(1)
        return res
    port = {"1": port_times(1), str(nt): port_times(nt)}
    for k in ("continuity_PPM", "CorAdCalc", "advect_tracer"):
        out["kernels"][k] = {f"reference_s_{t}thr": ref[t][k] for t in ref}
        out["kernels"][k].update({f"port_s_{t}thr": port[t][k] for t in port})
        out["kernels"][k].update({f"port_over_reference_{t}thr": port[t][k] / ref[t][k] for t in ref})
    out["note"] = ("advect_tracer's port time includes the copy of its tracers (as the reference's timing loop restores them); the port's "
                   "ctypes call overhead is inside its times")
print(json.dumps(out))
