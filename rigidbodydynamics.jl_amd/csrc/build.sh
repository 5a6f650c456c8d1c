#!/bin/bash
# Builds librbd_hip.so (gfx950 only) in-tree. Usage: ./build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -ffp-contract=fast -Wall -Wno-unused-function"
$HIPCC $FLAGS -c rbd_kernels.hip -o rbd_kernels.o "$@"
$HIPCC $FLAGS -c rbd_capi.hip -o rbd_capi.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o librbd_hip.so rbd_kernels.o rbd_capi.o
echo "built $(pwd)/librbd_hip.so"
