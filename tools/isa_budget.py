#!/usr/bin/env python3
"""usage: isa_budget.py <kernel .s extracted from `hipcc -S --cuda-device-only`>
Static instruction budget of k_sweep_lean4 by phase (vector / scalar / LDS / vector-memory instructions; waits and nops apart):
head (descriptor, per-lane loads), window fill (loop nest, static body), lane state, each action-loop form (per group of four
actions) and its remainder tails, exact pass, store + statistics.  DESIGN.md 4.2d quotes it."""
import re, sys
L = open(sys.argv[1]).read().split("\n")
def cnt(a, b):
    v = s = l = m = w = 0
    for x in L[a:b]:
        t = x.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."): continue
        op = t.split()[0]
        if op.startswith("v_"): v += 1
        elif op.startswith("ds_"): l += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): m += 1
        elif op in ("s_waitcnt", "s_nop"): w += 1
        elif op.startswith("s_"): s += 1
    return "VALU %4d  SALU %4d  LDS %3d  VMEM %3d  wait/nop %3d" % (v, s, l, m, w)
bars = [i for i, x in enumerate(L) if x.strip() == "s_barrier"]
fill = next(i for i, x in enumerate(L) if "Loop Header: Depth=1" in x)
loops = []
for i, x in enumerate(L):
    if "Inner Loop Header: Depth=1" in x and bars[0] < i < bars[1]:
        lab = x.split(":")[0]
        e = next((j for j in range(i, len(L)) if re.search(r"s_cbranch_\w+ " + re.escape(lab) + r"\b", L[j])), None)
        if e: loops.append((i, e))
print("head (descriptor, per-lane loads)      ", cnt(0, fill))
print("window fill (static body of the nest)  ", cnt(fill, bars[0]))
print("lane state, loop choice                ", cnt(bars[0], loops[0][0]))
names = ["loop A: one group of 4 actions", "loop B: one group of 4 actions", "exact pass: one action"]
for k, (a, b) in enumerate(loops[:3]):
    print("%-39s" % names[k], cnt(a, b + 1))
    nxt = loops[k + 1][0] if k + 1 < len(loops) else bars[1]
    print("  ... behind it (tails / set-up of next)", cnt(b + 1, nxt))
print("store + statistics                     ", cnt(bars[1] - 40, len(L)))
print("whole kernel                           ", cnt(0, len(L)))
for key in ("private_segment_fixed_size", "next_free_vgpr", "next_free_sgpr"):
    print([x.strip() for x in L if key in x][:1])
