#!/bin/bash
# tools/hip_api_counts.sh [CONFIG] [TAG] -- HIP API calls of the full step (rocprofv3 --hip-trace --stats, no counters): how many host
# synchronisations, launches and small copies a step makes. 30 steps after 20 + 4 x 2 steps of the stage loop = 38 steps; counts are
# totals over the run, the last line divides the three that matter by the step count.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
C=${1:-C4}; TAG=${2:-r04}
rm -rf /tmp/hipapi
rocprofv3 --hip-trace --stats --output-format csv -d /tmp/hipapi -o api -- python3 tools/fullstep_stages.py $C 30 2 > /tmp/hipapi.log 2>&1
f=$(find /tmp/hipapi -name "*hip_api_stats.csv" | head -1)
[ -z "$f" ] && { grep -v "^W2026" /tmp/hipapi.log | tail -8; tail -3 /tmp/hipapi.log; find /tmp/hipapi | head; exit 1; }
OUT=gpurun_out/${TAG}_hip_api_stats_${C}.txt
{ echo "# $C, 38 full time steps (30 lead-in + 8 measured by tools/fullstep_stages.py), rocprofv3 --hip-trace --stats"; head -25 "$f" | cut -c1-140; python3 - "$f" <<'PY'
import csv, sys
rows = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(sys.argv[1]))}
n = 38.0
print("# per step: hipStreamSynchronize %.1f + hipEventSynchronize %.1f, hipLaunchKernel %.0f, hipMemcpyAsync %.1f, hipOccupancyMaxActiveBlocksPerMultiprocessor %.2f"
      % (rows.get("hipStreamSynchronize", 0) / n, rows.get("hipEventSynchronize", 0) / n, rows.get("hipLaunchKernel", 0) / n,
         rows.get("hipMemcpyAsync", 0) / n, rows.get("hipOccupancyMaxActiveBlocksPerMultiprocessor", 0) / n))
PY
} | tee $OUT
