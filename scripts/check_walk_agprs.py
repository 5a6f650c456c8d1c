#!/usr/bin/env python3
"""Build-time check for aba_walk_kernel (csrc/rbd_walk.hpp): WalkStash addresses accumulation registers by number, from a255 downwards,
through inline asm the register allocator cannot see.  The allocator itself hands out accumulation registers from a0 upwards when VGPRs
run short.  This script reads the device assembly of rbd_walk_kernels.hip and fails if, in any instantiation, an allocator-chosen
accumulation register (printed `aN`; the stash's are printed `a[N]`) reaches the WALK_MAX_STEPS steps' worth the stash may use, or if
the stash's own lowest register is not where the header says.   usage: check_walk_agprs.py <file.s> <max_steps>"""
import re, sys
src, max_steps = sys.argv[1], int(sys.argv[2])
name, worst = None, {}
for line in open(src):
    m = re.match(r"^(_Z\S*(?:aba|rnea)_(?:walk|pipe)_kernel\S*):", line)
    if m:
        name = m.group(1); worst[name] = [-1, 256]
        continue
    if name is None or "v_accvgpr" not in line and " a" not in line:
        continue
    for tok in re.findall(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]", line):  # allocator-chosen: aN or a[N:M]
        hi = int(tok[0]) if tok[0] else int(tok[2])
        worst[name][0] = max(worst[name][0], hi)
    for tok in re.findall(r"\ba\[(0x[0-9a-f]+|\d+)\]", line):          # the stash: a[N]
        worst[name][1] = min(worst[name][1], int(tok, 0))
bad = 0
for k, (alloc_hi, stash_lo) in worst.items():
    words = 2 if ("kernelId" in k or "kernelIDv2_f" in k) else 1  # fp64 and the packed pair take two registers per value
    floor = 256 - max_steps * 9 * words
    ok = alloc_hi < floor and stash_lo >= floor
    print(f"{'ok ' if ok else 'BAD'} {k[:44]}: allocator uses a0..a{alloc_hi}, stash lowest a{stash_lo}, reserved from a{floor}")
    bad += not ok
sys.exit(1 if bad or not worst else 0)
