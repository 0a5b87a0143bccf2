#!/bin/bash
# Runs on the GPU box (gpurun): the bench line, the kernel-trace summary (mean / median / p95 per kernel) of the same command and
# the two HBM PMC passes; everything lands under gpurun_out/ (copy what is to be judged into profiles/).
# usage: tools/make_profiles.sh TAG [bench args...]
TAG=${1:-r04_c4}; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 bench.py "$@" > gpurun_out/${TAG}_bench.json 2> /tmp/bench.err || tail -5 /tmp/bench.err
# Per-kernel tables are taken with the stages of a step back to back (--no-overlap): in the default step the position correction
# runs beside the pressure solve on a second stream and the kernels of both stretch each other. The bench line's stage_ms_* /
# roofline figures are taken the same way (its second, serial loop); the kernel trace of the default, overlapped run is kept too.
LIGHT="--no-cpu-baseline --no-hot-path --no-kernel-timing --no-mic0-record"
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -- python3 bench.py $LIGHT --no-overlap "$@" > gpurun_out/${TAG}_bench_under_rocprof.json 2> /tmp/kt.log
# the first (lead-in) dispatches of every kernel are dropped from mean / median / p95: the table describes the timed region
python3 tools/kernel_trace_summary.py "$(find /tmp/kt -name '*results.db' | head -1)" 20 > gpurun_out/${TAG}_kernel_stats.csv
rm -rf /tmp/kto && rocprofv3 --kernel-trace --stats -d /tmp/kto -- python3 bench.py $LIGHT --no-serial-stages "$@" > gpurun_out/${TAG}_bench_under_rocprof_overlapped.json 2> /tmp/kto.log
python3 tools/kernel_trace_summary.py "$(find /tmp/kto -name '*results.db' | head -1)" 20 > gpurun_out/${TAG}_kernel_stats_overlapped.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C && rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -- python3 bench.py $LIGHT --no-overlap --steps 10 --warmup 20 "$@" > /dev/null 2> /tmp/pmc_$C.log
  python3 tools/pmc_summary.py "$(find /tmp/pmc_$C -name '*results.db' | head -1)" > gpurun_out/${TAG}_pmc_$(echo $C | tr A-Z a-z | sed 's/_size//').csv
done
python3 tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch.csv gpurun_out/${TAG}_pmc_write.csv gpurun_out/${TAG}_pmc_traffic.json "$TAG: python3 bench.py $*"
head -14 gpurun_out/${TAG}_kernel_stats.csv
