#!/bin/bash
# usage: tools/pmc_kernel.sh TAG KERNEL_PATTERN "CTR CTR ..." ["CTR ..." more passes] -- program args...
# One rocprofv3 --pmc pass per counter group (never combined with other trace domains); prints the per-kernel medians
# (tools/pmc_summary.py) of the kernels matching the pattern and keeps the csv under gpurun_out/.
TAG=$1; PAT=$2; shift 2
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
: > gpurun_out/pmc_$TAG.csv
i=0
for G in "${GROUPS_[@]}"; do
  rm -rf /tmp/pmc_${TAG}_$i
  rocprofv3 --pmc $G --kernel-trace -d /tmp/pmc_${TAG}_$i -- "$@" > /tmp/pmc_${TAG}_$i.log 2>&1
  DB=$(find /tmp/pmc_${TAG}_$i -name "*results.db" | head -1)
  python3 tools/pmc_summary.py "$DB" | grep -i "$PAT\|kernel,counter" >> gpurun_out/pmc_$TAG.csv 2>> /tmp/pmc_${TAG}_$i.log || tail -5 /tmp/pmc_${TAG}_$i.log
  i=$((i+1))
done
cat gpurun_out/pmc_$TAG.csv
