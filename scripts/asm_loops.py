#!/usr/bin/env python3
"""Static summary of a gfx950 kernel's assembly: for every loop (a backward branch) the instruction mix of its body.
usage: asm_loops.py file.s [mangled-kernel-substring]   (file.s from hipcc --cuda-device-only -S)"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2] if len(sys.argv) > 2 else None
# cut out the kernel
if key:
    start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ": " in l))
    end = next(i for i in range(start, len(src)) if src[i].strip().startswith(".Lfunc_end"))
    src = src[start:end]
labels = {}
ins = []
for l in src:
    s = l.strip()
    m = re.match(r"^(\.LBB[0-9_]+):", s)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    ins.append(s)
def kind(s):
    op = s.split()[0]
    if op.startswith("v_"):
        if "dpp" in s or "row_sh" in s or "wave_sh" in s or "quad_perm" in s or "row_bcast" in s: return "dpp"
        if op.startswith("v_cndmask"): return "cnd"
        if op.startswith("v_accvgpr"): return "acc"
        if op.startswith("v_mov") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"): return "mov"
        if "f64" in op: return "f64"
        return "valu"
    if op.startswith("ds_bpermute"): return "bperm"
    if op.startswith("ds_"): return "ds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"): return "vmem"
    return "other"
tot = {}
for s in ins:
    tot[kind(s)] = tot.get(kind(s), 0) + 1
print("static total", len(ins), dict(sorted(tot.items())))
loops = []
for i, s in enumerate(ins):
    if s.startswith("s_cbranch") or s.startswith("s_branch"):
        t = s.split()[-1]
        if t in labels and labels[t] <= i:
            loops.append((labels[t], i))
for a, b in sorted(loops):
    mix = {}
    for s in ins[a:b + 1]:
        mix[kind(s)] = mix.get(kind(s), 0) + 1
    print(f"loop ins[{a}:{b}] len {b - a + 1}: {dict(sorted(mix.items()))}")
