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
pids=()
# the headers the run-time compiled kernels include, as string literals for hiprtc (rbd_jit.hip)
python3 ../../scripts/embed_jit_headers.py
TUS="rbd_kernels rbd_bank_kernels rbd_big_kernels rbd_walk_kernels rbd_state_kernels rbd_contact_kernels rbd_capi rbd_comm rbd_jit"
for tu in $TUS; do
  $HIPCC $FLAGS -c $tu.hip -o ${TAG}_${tu#rbd_}.o "$@" &
  pids+=($!)
done
$HIPCC $FLAGS --cuda-device-only -S rbd_walk_kernels.hip -o ${TAG}_walk_kernels.s "$@" &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
# the walk kernel addresses accumulation registers by number: the register allocator must stay clear of them (scripts/check_walk_agprs.py)
python3 ../../scripts/check_walk_agprs.py ${TAG}_walk_kernels.s 11
OBJS=""; for tu in $TUS; do OBJS="$OBJS ${TAG}_${tu#rbd_}.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS -ldl
echo "built $(pwd)/$OUT"
