#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== first fit"; python scripts/class_throughput.py --real 3 6 12 14 16 20 28 30 45 50 60 62 2>&1 | grep GCUPS
for sh in 1 2 3 4 5 6 7 8 9; do echo "== prefer shape $sh"; DECIPHON_HIP_PACK_PREFER=$sh python scripts/class_throughput.py --real 3 6 12 14 16 20 28 30 45 50 60 62 2>&1 | grep GCUPS; done
