#!/usr/bin/env python3
"""Static ISA summary of one kernel: instruction mix per loop (backward branch) — used to spot serialised
shuffles / waitcnt storms before spending GPU minutes.  usage: isa_loops.py <file.s> <kernel-name-substring>"""
import re
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end + 1]
ins = [l.split()[0] for l in body if re.match(r"^\s+[a-z]", l)]
c = Counter(ins)
print(f"{key}: {len(ins)} instructions; f64 {sum(v for k, v in c.items() if 'f64' in k)}, f32 {sum(v for k, v in c.items() if 'f32' in k)}, "
      f"bpermute {c['ds_bpermute_b32']}, waitcnt {c['s_waitcnt']}, v_mov {c['v_mov_b32_e32'] + c['v_mov_b64_e32']}, saveexec {c['s_and_saveexec_b64']}")
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
seen = {}
for i, l in enumerate(body):
    m = re.match(r"\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        seen[m.group(2)] = (labels[m.group(2)], i)  # keep the outermost back-edge per header
for lab, (a, b) in sorted(seen.items(), key=lambda kv: kv[1]):
    cc = Counter(x.split()[0] for x in body[a:b] if re.match(r"^\s+[a-z]", x))
    tot = sum(cc.values())
    print(f"  loop {lab:12s} {tot:5d} instrs: f64 {sum(v for k, v in cc.items() if 'f64' in k):4d} f32 {sum(v for k, v in cc.items() if 'f32' in k):4d} "
          f"bperm {cc['ds_bpermute_b32']:4d} waitcnt {cc['s_waitcnt']:4d} mov {cc['v_mov_b32_e32'] + cc['v_mov_b64_e32']:4d} "
          f"cndmask {cc['v_cndmask_b32_e32'] + cc['v_cndmask_b32_e64']:3d} saveexec {cc['s_and_saveexec_b64']:3d} sload {sum(v for k, v in cc.items() if k.startswith('s_load')):2d} vmem {sum(v for k, v in cc.items() if k.startswith('global_')):3d}")
