#!/bin/bash
# kernel trace of the C2 step (stages back to back): what a PCG iteration is made of at the size hosts actually use
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python3 bench.py --config ${1:-C2} --steps 30 --warmup 20 --no-cpu-baseline --no-hot-path --no-kernel-timing --no-mic0-record --no-overlap > /tmp/kt2.json 2> /tmp/kt2.log
python3 tools/kernel_trace_summary.py "$(find /tmp/kt2 -name '*results.db' | head -1)" 20 > gpurun_out/r03_${1:-C2}_kernel_stats.csv
head -24 gpurun_out/r03_${1:-C2}_kernel_stats.csv | cut -c1-130
