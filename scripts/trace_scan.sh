ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/scan_trace -- python3 $ROOT/scripts/scan_headline.py --repeat 2 --no-callback > $ROOT/gpurun_out/scan_trace.log 2>&1 || exit 1
python3 - $ROOT/gpurun_out/scan_trace <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "dcp_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_enc = max(i for i, r in enumerate(rows) if "dcp_encode" in r["Kernel_Name"])
rows = rows[last_enc:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if e - s < 300000: continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print(f"{name:48s} {1e-6 * s:8.2f} {1e-6 * e:8.2f}  q={r.get('Queue_Id','?')}")
PY
rm -rf $ROOT/gpurun_out/scan_trace
