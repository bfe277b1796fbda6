#!/bin/bash
# Pure VALU streams for seconds, with the shader clock and package power sampled beside them (rocm-smi, read-only).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/vs "$ROOT/scripts/valu_sustained.hip" || exit 1
for cfg in "2 0" "4 0" "8 0" "8 1"; do
  set -- $cfg
  /tmp/vs $1 6 $2 > /tmp/vs.out &
  P=$!
  sleep 2.5
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '; echo; sleep 1; done
  wait $P
  cat /tmp/vs.out
done
