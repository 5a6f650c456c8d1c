#!/bin/bash
# device assembly of the banked kernels + static loop summary of one instantiation (default: fp64 SIMPLE, the bench kernel)
# usage: scripts/bank_asm.sh [mangled-prefix] [extra hipcc flags]
cd "$(dirname "$0")/../rigidbodydynamics.jl_amd/csrc"
K=${1:-_ZN3rbd15aba_bank_kernelIdLb0ELb1E}; shift
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -ffp-contract=fast -Wall -Wno-unused-function --cuda-device-only -S rbd_bank_kernels.hip -o /tmp/asm/bank2.s "$@" 2>&1 | grep -v "hip-link"
cd /tmp/asm
a=$(grep -n "^$K" bank2.s | head -1 | cut -d: -f1)
b=$(awk -v a=$a 'NR>a && /^\.Lfunc_end/ {print NR; exit}' bank2.s)
sed -n "${a},${b}p" bank2.s > bank_sel.s
sed -n "${b},\$p" bank2.s | grep -m4 "NumVgprs\|ScratchSize\|codeLenInByte\|NumAgprs"
python3 /root/repo/scripts/asm_loops.py bank_sel.s
