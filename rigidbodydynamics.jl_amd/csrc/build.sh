#!/bin/bash
# Builds librbd_hip.so (gfx950 only) in-tree.
#   ./build.sh                      -> librbd_hip.so
#   OUT=librbd_hip_x.so ./build.sh -DFOO ...   -> a variant for A/B experiments (select with env RBD_LIB=<path>)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${OUT:-librbd_hip.so}
TAG=${OUT%.so}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -ffp-contract=fast -Wall -Wno-unused-function"
$HIPCC $FLAGS -c rbd_kernels.hip -o ${TAG}_kernels.o "$@"
$HIPCC $FLAGS -c rbd_capi.hip -o ${TAG}_capi.o "$@"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT ${TAG}_kernels.o ${TAG}_capi.o
echo "built $(pwd)/$OUT"
