#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/ablation.log
for ph in 1 2 3 4 5 0; do
  RBD_ABA_STOP_AFTER=$ph python scripts/ablate_once.py 2>&1 | tail -1 | tee -a gpurun_out/ablation.log
done
