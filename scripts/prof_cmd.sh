#!/bin/bash
# Runs on the GPU box: kernel trace + two PMC passes of an arbitrary python command line, summarised.
# Usage: scripts/prof_cmd.sh <tag> <python script> [args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/$1" "${@:2}" > "$OUT/trace.log" 2>&1 || exit 1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/$1" "${@:2}" > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i failed" >> "$OUT/errors.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
# per dispatch: duration and counters, in dispatch order (the script launches kernels one after the other)
trace = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    trace += list(csv.DictReader(open(f)))
trace = [r for r in trace if "dcp_cost" in r["Kernel_Name"]]
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
pmc = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "dcp_cost" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        rank = {i: k for k, i in enumerate(ids)}
        for r in rows:
            pmc[rank[int(r["Dispatch_Id"])]][r["Counter_Name"]] = pmc[rank[int(r["Dispatch_Id"])]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(f"{'kernel':34s} {'ms':>8s} {'VALU/wave':>10s} {'G VALU/s':>9s} {'waves':>8s} {'wait_inst%':>10s} {'vmem/wave':>9s}")
for k, r in enumerate(trace):
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    c = pmc.get(k, {})
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    w = c.get("SQ_WAVES", 0) or 1
    print(f"{name:34s} {ms:8.3f} {c.get('SQ_INSTS_VALU', 0) / w:10.0f} {c.get('SQ_INSTS_VALU', 0) / ms / 1e6:9.1f} {w:8.0f} "
          f"{100 * c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):10.1f} {c.get('SQ_INSTS_VMEM_RD', 0) / w:9.0f}")
PY
