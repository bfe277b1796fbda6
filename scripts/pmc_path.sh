#!/bin/bash
# Runs on the GPU box: PMC passes over one whole scan (scripts/scan_headline.py), summed per kernel of the PATH pass
# (dcp_cost_ckpt_kernel, dcp_cost_store_kernel, dcp_traceback_kernel).  --pmc serialises the kernels: the durations of
# this run are not the pass's; the counters are.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_path
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/scripts/scan_headline.py" --repeat 1 --no-callback > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(k in n for k in ("ckpt", "store", "traceback")):
                continue
            tot[n][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[n].add((d, r["Dispatch_Id"]))
print(f"{'kernel':34s} {'launches':>8s} {'waves':>8s} {'VALU/wave':>10s} {'active%':>8s} {'wait_any%':>9s} {'rd/wave':>8s} {'wr/wave':>8s} {'fetch MB':>9s} {'write MB':>9s}")
for n in sorted(tot):
    c = tot[n]
    w = c.get("SQ_WAVES", 0) or 1
    cyc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    nl = len({x for x in disp[n] if x[0].endswith("pmc1")})
    print(f"{n:34s} {nl:8d} {w:8.0f} {c.get('SQ_INSTS_VALU', 0) / w:10.0f} {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / cyc:8.1f} "
          f"{100 * c.get('SQ_WAIT_ANY', 0) / cyc:9.1f} {c.get('SQ_INSTS_VMEM_RD', 0) / w:8.0f} {c.get('SQ_INSTS_VMEM_WR', 0) / w:8.0f} "
          f"{2 * c.get('FETCH_SIZE', 0) / 1024:9.1f} {c.get('WRITE_SIZE', 0) / 1024:9.1f}")
PY
rm -rf "$OUT"/pmc*/
