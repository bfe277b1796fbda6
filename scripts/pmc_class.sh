#!/bin/bash
# PMC pass of scripts/class_throughput.py for the given core sizes: scripts/pmc_class.sh <tag> "<counters>" K...
set -u
TAG=$1; PMC=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc1" -- python3 "$ROOT/scripts/class_throughput.py" "$@" > "$OUT/pmc1.log" 2>&1 || exit 1
python3 "$ROOT/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -A12 "dcp_cost" "$OUT/summary.txt" | head -60
tail -3 "$OUT/pmc1.log"
