#!/usr/bin/env python3
"""Static instruction census of the row loop of a cost kernel, from the device assembly
(hipcc --cuda-device-only -S).  Prints per basic block of the outermost loop of the chosen kernel:
VALU / DPP / s_nop / SALU / SMEM / VMEM / LDS / branch counts.  The row loop is unrolled x5, so
"per row" = loop total / 5 (lazy D->D inner loops are listed separately: they run a data-dependent
number of turns).
usage: isa_census.py kernels.s <mangled-name-substring>"""
import re
import sys
from collections import Counter

src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]


def kind(ins):
    op = ins.split()[0]
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_waitcnt", "s_barrier")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        if "_dpp" in op or " row_" in ins or "wave_sh" in ins or "quad_perm" in ins:
            return "dpp"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "lane"
        return "valu"
    return "other"


blocks, cur = [], None
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
    if m:
        cur = [m.group(1), m.group(2) or "", Counter()]
        blocks.append(cur)
        continue
    s = l.strip()
    if not s or s.startswith((";", ".", "//")) or cur is None:
        continue
    if re.match(r"^[a-z_0-9]+(\s|$)", s):
        cur[2][kind(s)] += 1
tot = Counter()
print(f"{'block':12s} {'valu':>5s} {'dpp':>4s} {'lane':>4s} {'nop':>4s} {'salu':>5s} {'smem':>5s} {'vmem':>5s} {'lds':>4s} {'br':>3s}  note")
for name, note, c in blocks:
    inloop = "in Loop" in note or "Loop Header" in note
    if not inloop:
        continue
    inner = "Depth=2" in note or "Inner Loop" in note
    print(f"{name:12s} {c['valu']:5d} {c['dpp']:4d} {c['lane']:4d} {c['nop']:4d} {c['salu']:5d} {c['smem']:5d} {c['vmem']:5d} {c['lds']:4d} {c['branch']:3d}  {'INNER ' if inner else ''}{note.strip('; ')[:50]}")
    if not inner:
        tot.update(c)
print("outer-loop blocks (5 rows), inner lazy loops excluded:", dict(tot))
v = tot["valu"] + tot["dpp"] + tot["lane"]
print(f"per row: VALU-class {v / 5:.1f} (+ s_nop {tot['nop'] / 5:.1f}), SALU {tot['salu'] / 5:.1f}, SMEM {tot['smem'] / 5:.1f}, VMEM {tot['vmem'] / 5:.1f}, LDS {tot['lds'] / 5:.1f}")
