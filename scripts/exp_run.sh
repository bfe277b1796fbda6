#!/bin/bash
# kernel experiments: the same throughput probe under several builds of the library (gpurun_exp/<name>)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in main "$@"; do
  if [ $lib = main ]; then unset DECIPHON_HIP_LIBDIR; else export DECIPHON_HIP_LIBDIR=$PWD/gpurun_exp/$lib; fi
  echo "== $lib (packed)"; python scripts/class_throughput.py --real 3 12 28 60 93 124 173 256 400 2>&1 | grep GCUPS
done
