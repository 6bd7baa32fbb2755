#!/bin/bash
# Kernel statistics of tools/perf_lateral.py on the GPU box:  tools/prof_lateral.sh TAG  -> gpurun_out/TAG_lateral_kernel_stats.txt
set -e
TAG=${1:-x}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d /tmp/kt_lat_$TAG -o kt -- python3 $GRAFT_REPO_ROOT/tools/perf_lateral.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_lateral.json 2> /tmp/kt_lat_$TAG.err
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt_lat_$TAG -name "*.db" | head -n 1) > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_lateral_kernel_stats.txt
