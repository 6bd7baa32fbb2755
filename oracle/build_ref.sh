#!/bin/bash
# Compiles, UNMODIFIED and where they lie under /root/reference, the reference source files of the hot
# path that need nothing but themselves (no MOM_error_handler / MOM_grid / FMS behind them), plus our
# bind(C) caller oracle/ref_wrap.F90, into oracle/_ref/libmom6ref.so.  Everything else on the path
# `use`s modules that end in FMS (not vendored) and is treated as unbuildable here.
# -O0 -ffp-contract=off: the unfused IEEE evaluation of the source's parenthesisation.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference
FC=${FC:-/opt/rocm/bin/amdflang}
[ -d "$REF" ] || { echo "build_ref.sh: $REF not present, skipping"; exit 0; }
[ -x "$FC" ] || { echo "build_ref.sh: $FC not present, skipping"; exit 0; }
OUT="$HERE/_ref"
mkdir -p "$OUT/mod"
FLAGS="-fdefault-real-8 -O0 -ffp-contract=off -fPIC -J $OUT/mod -I $OUT/mod"
$FC $FLAGS -c "$REF/src/ALE/PCM_functions.F90" -o "$OUT/PCM_functions.o"
$FC $FLAGS -c "$REF/src/ALE/PLM_functions.F90" -o "$OUT/PLM_functions.o"
$FC $FLAGS -c "$REF/src/framework/MOM_array_transform.F90" -o "$OUT/MOM_array_transform.o"
$FC $FLAGS -c "$REF/src/ALE/MOM_hybgen_remap.F90" -o "$OUT/MOM_hybgen_remap.o"
$FC $FLAGS -c "$REF/src/equation_of_state/MOM_EOS_base_type.F90" -o "$OUT/MOM_EOS_base_type.o"
$FC $FLAGS -c "$REF/src/equation_of_state/MOM_EOS_UNESCO.F90" -o "$OUT/MOM_EOS_UNESCO.o"
$FC $FLAGS -c "$HERE/ref_wrap.F90" -o "$OUT/ref_wrap.o"
$FC -shared -o "$OUT/libmom6ref.so" "$OUT/PCM_functions.o" "$OUT/PLM_functions.o" "$OUT/MOM_array_transform.o" "$OUT/MOM_hybgen_remap.o" "$OUT/MOM_EOS_base_type.o" "$OUT/MOM_EOS_UNESCO.o" "$OUT/ref_wrap.o"
echo "built $OUT/libmom6ref.so"
# the same sources at -O2, for the timing calibration of tools/calibrate_ref.py only (the parity checks use the -O0 build)
mkdir -p "$OUT/mod_O2"
FLAGS2="-fdefault-real-8 -O2 -ffp-contract=off -fPIC -J $OUT/mod_O2 -I $OUT/mod_O2"
for f in src/ALE/PCM_functions src/ALE/PLM_functions src/ALE/MOM_hybgen_remap src/framework/MOM_array_transform \
         src/equation_of_state/MOM_EOS_base_type src/equation_of_state/MOM_EOS_UNESCO; do
  $FC $FLAGS2 -c "$REF/$f.F90" -o "$OUT/$(basename $f).O2.o"
done
$FC $FLAGS2 -c "$HERE/ref_wrap.F90" -o "$OUT/ref_wrap.O2.o"
$FC -shared -o "$OUT/libmom6ref_O2.so" "$OUT/PCM_functions.O2.o" "$OUT/PLM_functions.O2.o" "$OUT/MOM_array_transform.O2.o" "$OUT/MOM_hybgen_remap.O2.o" "$OUT/MOM_EOS_base_type.O2.o" "$OUT/MOM_EOS_UNESCO.O2.o" "$OUT/ref_wrap.O2.o"
echo "built $OUT/libmom6ref_O2.so"
