export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -c 300 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
for k, v in d.items():
    if isinstance(v, dict) and ("ms_per_step" in v or "us_per_step" in v):
        print(f"  {k:40s}", v.get("value"), v.get("ms_per_step", v.get("us_per_step")), (v.get("roofline") or {}).get("kernel"), v.get("parity_check"))
    elif isinstance(v, str) and v.startswith("failed"):
        print("  ", k, v[:300])
PY
