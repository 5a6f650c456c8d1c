export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_state_kernels.py -x -q -m gpu -k "compiled_aba" 2>&1 | tail -5)
for B in 4096 8192 16384 32768 65536; do python scripts/exp_aba_f64.py randmech1 $B 2>/dev/null | tail -1; done
python scripts/exp_aba_f64.py randmech2 65536 2>/dev/null | tail -1
python scripts/exp_aba_f64.py randmech3 65536 2>/dev/null | tail -1
