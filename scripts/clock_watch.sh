#!/bin/bash
# Samples the GPU's shader clock and power (rocm-smi, read-only) while bench.py runs its timed steps.
# Usage (GPU box): scripts/clock_watch.sh > gpurun_out/clock_watch.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
python3 "$ROOT/bench.py" --steps 40 --warmup 2 --no-cpu-baseline > /tmp/cw_bench.json 2>/tmp/cw_bench.err &
BPID=$!
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6
sleep 8   # import, workload build, staging
for i in $(seq 1 30); do
  kill -0 $BPID 2>/dev/null || break
  echo "sample $i:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
  sleep 1
done
wait $BPID
cut -c1-160 /tmp/cw_bench.json
