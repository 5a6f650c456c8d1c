export TMPDIR=/tmp
for v in "" fixed fixedilp; do
  if [ -n "$v" ]; then export RBD_LIB=$PWD/rigidbodydynamics.jl_amd/csrc/librbd_hip_$v.so; fi
  for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-extra-legs 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] steps 2000:', round(d['ms_per_step']*1e3,2), 'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2), 'err', d['parity_rel_err_vs_oracle'])"
  done
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-extra-legs --steps 20 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] steps 20  :', round(d['ms_per_step']*1e3,2), 'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2))"
done
cd /tmp; rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/pf -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs --no-extra-legs --steps 30 --warmup 5 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
per=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pf/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'aba_bank' in r['Kernel_Name']: per[r['Counter_Name']][r['Dispatch_Id']].append(float(r['Counter_Value']))
tot={c: sum(sum(v) for v in d.values())/len(d) for c,d in per.items()}
w=tot.get('SQ_WAVES',1)
print('fixedilp per wave:', {c: round(v/w,1) for c,v in tot.items()})
PY
