#!/bin/bash
# A/B variants of the banked kernels: recompiles only rbd_bank_kernels.hip with the given flags and links it with the objects of the
# default build (run csrc/build.sh first).  usage: scripts/build_bank_variant.sh NAME -DRBD_BANK_WAVES=1 ...   -> csrc/librbd_hip_NAME.so
set -e
cd "$(dirname "$0")/../rigidbodydynamics.jl_amd/csrc"
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -ffp-contract=fast -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -c rbd_bank_kernels.hip -o librbd_hip_${NAME}_bank_kernels.o "$@"
T=librbd_hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librbd_hip_${NAME}.so ${T}_kernels.o librbd_hip_${NAME}_bank_kernels.o ${T}_big_kernels.o ${T}_walk_kernels.o ${T}_state_kernels.o ${T}_contact_kernels.o ${T}_capi.o ${T}_comm.o ${T}_jit.o -ldl
echo "built librbd_hip_${NAME}.so"
