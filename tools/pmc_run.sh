#!/bin/bash
# usage: tools/pmc_run.sh TAG "COUNTER COUNTER ..." -- program args...   (one rocprofv3 --pmc pass, summary csv under gpurun_out/)
TAG=$1; CTRS=$2; shift 3
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
rocprofv3 --pmc $CTRS --kernel-trace -d /tmp/pmc_$TAG -- "$@" > /tmp/pmc_$TAG.log 2>&1
DB=$(find /tmp/pmc_$TAG -name "*results.db" | head -1)
mkdir -p gpurun_out
python tools/pmc_summary.py "$DB" > gpurun_out/pmc_$TAG.csv 2>> /tmp/pmc_$TAG.log || tail -20 /tmp/pmc_$TAG.log
grep -i "k_pcg\|kernel,counter" gpurun_out/pmc_$TAG.csv | head -40
