# fp64 dynamics! of randmech(): the two programs (RBD_TUNE spec_f64_stash = 0: every row in LDS, 1: spare rows in the HBM stash), then the library's own choice
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_state_kernels.py -x -q -m gpu -k "compiled_aba" 2>&1 | tail -3)
for S in 0 1; do
  export RBD_TUNE="spec_f64_stash=$S"
  echo "=== spec_f64_stash=$S"
  for B in 4096 8192 16384 32768 49152 65536; do python scripts/exp_aba_f64.py randmech1 $B 2>/dev/null | tail -1; done
  python scripts/exp_aba_f64.py randmech2 65536 2>/dev/null | tail -1
  python scripts/exp_aba_f64.py randmech3 65536 2>/dev/null | tail -1
done
unset RBD_TUNE
echo "=== the library's choice"
for B in 16384 32768 49152 65536 131072; do python scripts/exp_aba_f64.py randmech1 $B 2>/dev/null | tail -1; done
python scripts/exp_aba_f64.py randmech2 65536 2>/dev/null | tail -1
python scripts/exp_aba_f64.py randmech3 65536 2>/dev/null | tail -1
