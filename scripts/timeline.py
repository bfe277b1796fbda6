#!/usr/bin/env python3
"""Start / end of every cost kernel of the LAST step of a rocprofv3 --kernel-trace of bench.py --profile,
relative to the step's first start: shows which kernels run side by side and what the tail is made of.
usage: timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "dcp_cost" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are separated by gaps with no cost kernel running: split on a start later than every earlier end
steps, cur, end = [], [], 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur and s > end:
        steps.append(cur)
        cur = []
    cur.append(r)
    end = max(end, e)
steps.append(cur)
last = steps[-1]
t0 = min(int(r["Start_Timestamp"]) for r in last)
t1 = max(int(r["End_Timestamp"]) for r in last)
print(f"{len(steps)} steps; last step {1e-6 * (t1 - t0):.2f} ms")
for r in last:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{name:45s} start {1e-6 * s:8.2f}  end {1e-6 * e:8.2f}  grid {r.get('Grid_Size', '?'):>9s}  wg {r.get('Workgroup_Size', '?')}")
