#!/usr/bin/env python3
"""Condenses a scripts/profile_bench.sh output directory: per-kernel time from the
rocprofv3 kernel trace and per-kernel sums of every PMC counter collected."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    return name.split("(")[0][:60]


rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = defaultdict(lambda: [0, 0.0, None])
for r in rows:
    d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    a = agg[short(r["Kernel_Name"])]
    a[0] += 1
    a[1] += d
    a[2] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
print("== kernel trace (ns) ==")
for k, (n, t, meta) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} calls={n:5d} total={t:14.0f} avg={t / n:12.0f}  vgpr/sgpr/lds/grid/wg={meta}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel_stats.csv ==")
    print(open(f).read())

print("== PMC (sum over dispatches, per kernel) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    for k in acc:
        n = max(1, len(cnt[k]))
        print(f"[{os.path.basename(d)}] {k} dispatches={n}")
        for c, v in sorted(acc[k].items()):
            print(f"    {c:36s} total={v:18.1f} per_dispatch={v / n:16.1f}")
