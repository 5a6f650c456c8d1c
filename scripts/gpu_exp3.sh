#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "mass_matrix or config3 or chol or solve or crba or state" 2>&1 | tail -4
TAG=full timeout 200 python scripts/exp_config3.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp3.txt
TAG=crba_only RBD_EXP_NO_CHOL=1 timeout 200 python scripts/exp_config3.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/exp3.txt
