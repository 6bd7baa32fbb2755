"""The module boundary, checked with the reference's own callers (build container only: needs /root/reference and amdflang).

1. The reference's UNMODIFIED src/core/MOM_dynamics_split_RK2.F90, MOM_continuity.F90 and MOM_PressureForce.F90 are compiled where
   they lie against the .mod files of the sub-module shims (mom6_amd/fortran/*_hip.F90): every keyword, optional argument, generic
   and public type those callers use of MOM_continuity_PPM, MOM_CoriolisAdv, MOM_barotropic, MOM_PressureForce_FV, MOM_vert_friction,
   MOM_set_visc, MOM_hor_visc, MOM_thickness_diffuse and MOM_ALE has to exist in the shims with the reference's shape.  The
   framework modules (MOM_grid, MOM_domains, MOM_diag_mediator, MOM_restart ... which end in FMS) are the declaration-only stand-ins
   of tests/fortran/stubs; the memory macros are the reference's own headers (-I, in place).
2. Every `use <replaced module>, only : ...` of the reference tree (src/ and config_src/drivers/, e.g. MOM.F90:53-59, :78-81) is
   collected and compiled against the shims, MOM_dynamics_split_RK2 and MOM_ALE included: no name the tree imports may be missing.
3. The other direction, by text: every name a shim imports from a module it does NOT replace must be public in the reference's
   module of that name (a shim that asked MOM_io for `directories`, which lives in MOM_get_input, compiled against the stand-ins
   for three rounds).
Nothing of the reference is copied: its files are read (and compiled) in place and the outputs go to a temporary directory."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FC = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
FDIR = os.path.join(ROOT, "mom6_amd", "fortran")
STUBS = os.path.join(ROOT, "tests", "fortran", "stubs")

pytestmark = [pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference is not mounted (GPU box)"),
              pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not present")]

# shim file -> (module it replaces, the reference file it takes the place of)
REPLACED = {
    "MOM_ALE_hip.F90": ("MOM_ALE", "src/ALE/MOM_ALE.F90"),
    "MOM_continuity_PPM_hip.F90": ("MOM_continuity_PPM", "src/core/MOM_continuity_PPM.F90"),
    "MOM_CoriolisAdv_hip.F90": ("MOM_CoriolisAdv", "src/core/MOM_CoriolisAdv.F90"),
    "MOM_barotropic_hip.F90": ("MOM_barotropic", "src/core/MOM_barotropic.F90"),
    "MOM_PressureForce_FV_hip.F90": ("MOM_PressureForce_FV", "src/core/MOM_PressureForce_FV.F90"),
    "MOM_tracer_advect_hip.F90": ("MOM_tracer_advect", "src/tracer/MOM_tracer_advect.F90"),
    "MOM_tracer_hor_diff_hip.F90": ("MOM_tracer_hor_diff", "src/tracer/MOM_tracer_hor_diff.F90"),
    "MOM_set_viscosity_hip.F90": ("MOM_set_visc", "src/parameterizations/vertical/MOM_set_viscosity.F90"),
    "MOM_vert_friction_hip.F90": ("MOM_vert_friction", "src/parameterizations/vertical/MOM_vert_friction.F90"),
    "MOM_thickness_diffuse_hip.F90": ("MOM_thickness_diffuse", "src/parameterizations/lateral/MOM_thickness_diffuse.F90"),
    "MOM_mixed_layer_restrat_hip.F90": ("MOM_mixed_layer_restrat", "src/parameterizations/lateral/MOM_mixed_layer_restrat.F90"),
    "MOM_hor_visc_hip.F90": ("MOM_hor_visc", "src/parameterizations/lateral/MOM_hor_visc.F90"),
    "MOM_dynamics_split_RK2_hip.F90": ("MOM_dynamics_split_RK2", "src/core/MOM_dynamics_split_RK2.F90"),
}
SUBMODULE_SHIMS = ["mom6hip_c_api.F90", "mom6hip_MOM_glue.F90", "MOM_ALE_hip.F90", "MOM_continuity_PPM_hip.F90", "MOM_CoriolisAdv_hip.F90",
                   "MOM_barotropic_hip.F90", "MOM_PressureForce_FV_hip.F90", "MOM_tracer_advect_hip.F90", "MOM_tracer_hor_diff_hip.F90",
                   "MOM_set_viscosity_hip.F90", "MOM_vert_friction_hip.F90", "MOM_thickness_diffuse_hip.F90",
                   "MOM_mixed_layer_restrat_hip.F90", "MOM_hor_visc_hip.F90"]


def _fc(args, cwd):
    r = subprocess.run([FC, *args], cwd=cwd, capture_output=True, text=True)
    errs = [ln for ln in r.stderr.splitlines() if "error" in ln.lower()]
    return r.returncode, errs, r.stderr


@pytest.fixture(scope="module")
def shim_mods(tmp_path_factory):
    """the stand-ins and the sub-module shims compiled once: a directory of .mod files"""
    d = tmp_path_factory.mktemp("shim_mods")
    flags = ["-cpp", "-DMOM6HIP_WITH_ALE_SHIM", "-fdefault-real-8", "-O0", "-ffp-contract=off", f"-I{STUBS}", f"-I{d}", "-J", str(d)]
    for src in [os.path.join(STUBS, "mom6_stubs.F90")] + [os.path.join(FDIR, s) for s in SUBMODULE_SHIMS]:
        rc, errs, full = _fc([*flags, "-c", src, "-o", str(d / (os.path.basename(src)[:-4] + ".o"))], str(d))
        assert rc == 0, f"{src}:\n" + full[-3000:]
    return d, flags


def test_reference_callers_compile_unmodified_against_the_shims(shim_mods, tmp_path):
    d, _ = shim_mods
    flags = ["-cpp", "-fdefault-real-8", "-O0", f"-I{REF}/config_src/memory/dynamic_symmetric", f"-I{REF}/src/framework", f"-I{d}",
             "-J", str(tmp_path)]
    for rel in ("src/core/MOM_continuity.F90", "src/core/MOM_PressureForce.F90", "src/core/MOM_dynamics_split_RK2.F90"):
        src = os.path.join(REF, rel)
        rc, errs, full = _fc([*flags, "-c", src, "-o", str(tmp_path / (os.path.basename(rel)[:-4] + ".o"))], str(tmp_path))
        assert rc == 0, f"{rel} does not compile against the shims ({len(errs)} errors):\n" + "\n".join(errs[:40])
    for m in ("mom_continuity.mod", "mom_pressureforce.mod", "mom_dynamics_split_rk2.mod"):
        assert (tmp_path / m).exists()


def _imports_of_replaced_modules():
    """{module: {name: [files]}} over the reference tree, the replaced module's own file left out"""
    mods = {m.lower(): (m, own) for m, own in REPLACED.values()}
    uses = {m: {} for m, _ in REPLACED.values()}
    files = glob.glob(REF + "/src/**/*.F90", recursive=True) + glob.glob(REF + "/config_src/drivers/**/*.F90", recursive=True)
    for fn in files:
        rel = os.path.relpath(fn, REF)
        txt = re.sub(r"&\s*\n\s*&?", "", open(fn, errors="replace").read())
        for line in txt.split("\n"):
            m = re.match(r"\s*use\s+(\w+)\s*,\s*only\s*:\s*(.*)", line.split("!")[0], re.I)
            if not m or m.group(1).lower() not in mods:
                continue
            mod, own = mods[m.group(1).lower()]
            if rel == own:
                continue
            for nm in m.group(2).split(","):
                nm = nm.strip()
                if "=>" in nm:
                    nm = nm.split("=>")[1].strip()
                if nm:
                    uses[mod].setdefault(nm.lower(), []).append(rel)
    return uses


def test_every_import_of_a_replaced_module_in_the_reference_tree_resolves(shim_mods, tmp_path):
    d, flags = shim_mods
    # the module MOM_dynamics_split_RK2 itself (it takes the place of the reference file test 1 compiles, hence its own directory)
    flags2 = [f for f in flags if f != str(d)]
    flags2 = flags2[:flags2.index("-J")] + [f"-I{d}", "-J", str(tmp_path)]
    rc, errs, full = _fc([*flags2, "-c", os.path.join(FDIR, "MOM_dynamics_split_RK2_hip.F90"), "-o", str(tmp_path / "rk2.o")], str(tmp_path))
    assert rc == 0, full[-3000:]
    uses = _imports_of_replaced_modules()
    assert sum(len(v) for v in uses.values()) > 100      # (the scan found the tree)
    assert "init_dyn_split_rk2_diabatic" in uses["MOM_dynamics_split_RK2"] and "adjustgridforintegrity" in uses["MOM_ALE"]
    src = []
    for mod, names in uses.items():
        src.append(f"module imports_of_{mod}\n" + "\n".join(f"use {mod}, only : {n}" for n in sorted(names)) + "\nimplicit none\nend module\n")
    (tmp_path / "imports.F90").write_text("\n".join(src))
    rc, errs, full = _fc(["-cpp", "-fdefault-real-8", f"-I{tmp_path}", f"-I{d}", "-J", str(tmp_path), "-fsyntax-only", "imports.F90"], str(tmp_path))
    missing = sorted(set(re.findall(r"'(\w+)' not found in module '(\w+)'", full)))
    assert rc == 0 and not missing, "names the reference tree imports that a shim does not export: " + \
        ", ".join(f"{m}:{n} ({uses[[k for k in uses if k.lower() == m][0]][n][0]})" for n, m in missing) + "\n" + full[-1500:]


def _public_names(path):
    """the public names of the (first) module in a reference file; None if the module is public by default"""
    txt = re.sub(r"&\s*\n\s*&?", "", open(path, errors="replace").read())
    lines = [ln.split("!")[0] for ln in txt.split("\n")]
    head = []
    for ln in lines:
        if re.match(r"\s*contains\s*$", ln, re.I):
            break
        head.append(ln)
    default_private = any(re.match(r"\s*(implicit\s+none\s*;\s*)?private\s*$", ln, re.I) for ln in head)
    if not default_private:
        return None
    names = set()
    for ln in head:
        m = re.match(r"\s*public\s*(?:::)?\s*(.*)", ln, re.I)
        if m and not re.match(r"\s*public\s*$", ln, re.I):
            for nm in m.group(1).split(","):
                nm = nm.strip()
                mo = re.match(r"(?:operator|assignment)\s*\(.*\)", nm, re.I)
                if nm:
                    names.add((mo.group(0) if mo else nm).lower().replace(" ", ""))
        m = re.match(r"\s*type\s*,[^:]*\bpublic\b[^:]*::\s*(\w+)", ln, re.I)
        if m:
            names.add(m.group(1).lower())
        m = re.match(r"\s*(?:integer|real|logical|character)[^:]*\bpublic\b[^:]*::\s*(.*)", ln, re.I)
        if m:
            for nm in m.group(1).split(","):
                names.add(nm.split("=")[0].strip().lower())
    return names


def test_shims_import_only_what_the_reference_modules_export():
    where = {}
    for fn in glob.glob(REF + "/src/**/*.F90", recursive=True) + glob.glob(REF + "/config_src/infra/FMS2/*.F90") + \
            glob.glob(REF + "/config_src/memory/**/*.F90", recursive=True):
        m = re.search(r"^\s*module\s+(\w+)\s*$", open(fn, errors="replace").read(), re.I | re.M)
        if m:
            where.setdefault(m.group(1).lower(), fn)
    replaced = {m.lower() for m, _ in REPLACED.values()}
    # re-exports: a module that `use`s another and lists its names public re-exports them; followed one level below
    bad, checked = [], 0
    for shim in sorted(glob.glob(os.path.join(FDIR, "*.F90"))):
        txt = re.sub(r"&\s*\n\s*&?", "", open(shim).read())
        for line in txt.split("\n"):
            m = re.match(r"\s*use\s+(\w+)\s*,\s*only\s*:\s*(.*)", line.split("!")[0], re.I)
            if not m:
                continue
            mod = m.group(1).lower()
            if mod.startswith("mom6hip") or mod == "iso_c_binding" or mod in replaced:
                continue
            assert mod in where, f"{os.path.basename(shim)} uses module {m.group(1)}, which the reference does not have"
            pub = _public_names(where[mod])
            for nm in m.group(2).split(","):
                nm = nm.strip()
                if "=>" in nm:
                    nm = nm.split("=>")[1].strip()
                if not nm or pub is None:
                    continue
                checked += 1
                if nm.lower().replace(" ", "") not in pub:
                    bad.append(f"{os.path.basename(shim)}: {m.group(1)} does not export {nm}")
    assert checked > 150
    assert not bad, "\n".join(bad)
