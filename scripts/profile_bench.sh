#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Usage: scripts/profile_bench.sh <tag> [bench args...]; results land in gpurun_out/<tag>/
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# --profile: nothing but warmup + timed launches, so every dispatch of a cost kernel belongs to one step
ARGS="--steps 5 --warmup 1 --profile $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || exit 1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i failed" >> "$OUT/errors.log"
done
python3 "$ROOT/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
