export TMPDIR=/tmp
python scripts/sweep_mappings.py --dtypes f32 --batches 4096,8192,12288,16384,20480,24576,32768,49152,65536,98304,131072 --algos aba_banks,aba_walk,aba_compiled,aba --reps 100 2>&1 | tail -14
python scripts/bench_ops.py --batch 65536 --dtype f32 --only inverse 2>&1 | tail -4 | cut -c1-300
python scripts/bench_ops.py --batch 16384 --dtype f32 --only inverse 2>&1 | tail -4 | cut -c1-300
