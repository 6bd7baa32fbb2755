#!/bin/bash
# Collects the profiles/ set of a round on the GPU box:  tools/profile_round.sh r01_f
# (kernel trace of bench.py -> kernel stats; two PMC passes FETCH_SIZE / WRITE_SIZE -> per-kernel bytes).
set -e
TAG=${1:-r01_x}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
echo "profile_round: kernel trace" > $OUT/progress.txt
rocprofv3 --kernel-trace -d /tmp/kt_$TAG -o kt -- python3 bench.py --steps 8 --warmup 4 --spinup 0 --no-cpu-baseline --no-roofline > $OUT/bench_under_trace.json 2> /tmp/kt_$TAG.err
python3 tools/rocpd_stats.py $(find /tmp/kt_$TAG -name "*.db" | head -n 1) $OUT/${TAG}_bench_om4_kernel_stats.csv > $OUT/${TAG}_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  echo "profile_round: pmc $c" >> $OUT/progress.txt
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${TAG}_$c -o p -- python3 bench.py --steps 4 --warmup 0 --spinup 0 --no-cpu-baseline --no-roofline > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err
  python3 tools/rocpd_pmc.py $(find /tmp/pmc_${TAG}_$c -name "*.db" | head -n 1) > $OUT/${TAG}_pmc_$c.txt
done
echo "profile_round: pmc sq" >> $OUT/progress.txt
# fp64-VALU evidence for the compute-bound kernels (PressureForce, continuity): instruction and cycle counters, one more pass
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc_${TAG}_sq -o p -- python3 bench.py --steps 4 --warmup 0 --spinup 0 --no-cpu-baseline --no-roofline > /tmp/pmc_sq.json 2> /tmp/pmc_sq.err
python3 tools/rocpd_pmc.py $(find /tmp/pmc_${TAG}_sq -name "*.db" | head -n 1) > $OUT/${TAG}_pmc_sq.txt
echo done
