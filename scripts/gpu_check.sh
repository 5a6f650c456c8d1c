#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench (+variants), rocprofv3 kernel trace + PMC passes. Outputs under gpurun_out/.
mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/pmc*
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.json
echo "== bench, the N > 1 code path on this one GPU (two ranks on cuda:0, gloo for the collectives: sharding / max over ranks / per-rank rates run; the numbers mean nothing)"
RBD_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-400 | tee gpurun_out/bench_two_ranks.json
echo "== bench variants"; rm -f gpurun_out/bench_variants.jsonl
for extra in "--graph" "--dtype f32" "--dtype f32 --batch 65536 --steps 200" "--batch 65536 --steps 200" "--layout soa" "--wrenches" "--model atlas_fixed" "--batch 524288 --steps 30 --dtype f32"; do
  timeout 600 python bench.py --no-cpu-baseline $extra 2>&1 | tail -1 >> gpurun_out/bench_variants.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/bench_variants.jsonl"):
    try:
        d = json.loads(l); c = d["config"]
        print(f"{c['workload'][:60]:60s} graph={c['hip_graph']} layout={c['layout']} wrenches={c['external_wrenches']}: {d['value']/1e6:8.1f} Mevals/s kernel {d['roofline']['kernel_ms']*1e3:8.2f} us err {d['parity_rel_err_vs_oracle']:.1e}")
    except Exception as e:
        print("bad line", l[:200])
PY
echo "== rocprof kernel trace"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --no-cpu-baseline --steps 500 > $R/gpurun_out/rocprof_run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc1 -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc2 -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3 -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc4 -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2>&1
cd $R
python scripts/summarize_prof.py | tee gpurun_out/prof_summary.txt
