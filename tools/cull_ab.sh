#!/bin/bash
# Position correction with / without the reach cull (variants/nocull.so = -DLFA_CORR_NO_REACH_CULL=1): correction alone, then the step.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for W in C4 C3late; do
  for R in 1 2; do
    for V in default libfluid_amd/variants/nocull.so; do
      unset LFA_LIB_PATH
      if [ $V != default ]; then export LFA_LIB_PATH=$V; fi
      python3 tools/correct_ab.py $W 2>/dev/null | tail -1
    done
  done
done
unset LFA_LIB_PATH
bash tools/lib_ab.sh "C4" default libfluid_amd/variants/nocull.so
