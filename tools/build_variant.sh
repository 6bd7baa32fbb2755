#!/bin/bash
# tools/build_variant.sh TAG FILE.hip "-DNAME=VALUE ..."  ->  variants/libmom6hip_TAG.so : the library with one source rebuilt
# with extra defines (run with MOM6HIP_LIB_PATH=variants/libmom6hip_TAG.so).  For kernel experiments on the GPU box.
set -e
cd "$(dirname "$0")/../mom6_amd/csrc"
TAG=$1; SRC=$2; DEFS=$3
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $DEFS -c $SRC -o /tmp/variant_$TAG.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../variants/libmom6hip_$TAG.so $OBJS /tmp/variant_$TAG.o
echo variants/libmom6hip_$TAG.so
