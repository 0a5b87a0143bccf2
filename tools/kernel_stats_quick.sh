#!/bin/bash
# Per-kernel durations of the bench command with the stages back to back: tools/kernel_stats_quick.sh OUT [bench args]
OUT=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
LIGHT="--no-cpu-baseline --no-hot-path --no-kernel-timing --no-mic0-record"
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -- python3 bench.py $LIGHT --no-overlap "$@" > /tmp/kt.json 2> /tmp/kt.log
python3 tools/kernel_trace_summary.py "$(find /tmp/kt -name '*results.db' | head -1)" 20 > gpurun_out/$OUT
head -18 gpurun_out/$OUT
