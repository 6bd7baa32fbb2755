#!/bin/bash
# Counter passes for one tool script on the GPU box (one rocprofv3 run per counter group; --kernel-trace only):
#   tools/pmc_one.sh TAG "FETCH_SIZE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- tools/perf_hor_visc.py 1440x1080x75 2
# writes gpurun_out/pmc_TAG_<n>.txt (tools/rocpd_pmc.py listings)
set -e
export TMPDIR=/tmp
TAG=$1; shift
GRPS=()
while [ "$1" != "--" ]; do GRPS+=("$1"); shift; done
shift
n=0
for grp in "${GRPS[@]}"; do
  echo "pmc_one: $TAG pass $n: $grp" | tee -a gpurun_out/pmc_${TAG}_progress.txt
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_${TAG}_$n -o p -- python3 "$@" > /tmp/pmc_${TAG}_$n.out 2> /tmp/pmc_${TAG}_$n.err
  python3 tools/rocpd_pmc.py $(find /tmp/pmc_${TAG}_$n -name "*.db" | head -n 1) > gpurun_out/pmc_${TAG}_$n.txt
  n=$((n+1))
done
