#!/bin/bash
# tools/hip_api_counts.sh -- HIP API calls of the full step at C4 (rocprofv3 --hip-trace --stats, no counters): how many host
# synchronisations and small copies a step makes. 30 steps after 20; counts are totals over the run.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/hipapi
rocprofv3 --hip-trace --stats --output-format csv -d /tmp/hipapi -o api -- python3 tools/fullstep_stages.py C4 30 2 > /tmp/hipapi.log 2>&1
f=$(find /tmp/hipapi -name "*hip_api_stats.csv" | head -1)
[ -z "$f" ] && { grep -v "^W2026" /tmp/hipapi.log | tail -8; tail -3 /tmp/hipapi.log; find /tmp/hipapi | head; exit 1; }
head -25 "$f" | cut -c1-140 | tee gpurun_out/r03_hip_api_stats.txt
