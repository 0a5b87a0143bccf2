#!/bin/bash
# C4 late-window NaN seen once in round 6's baseline: is it reproducible, and at which step?
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
L="--no-cpu-baseline --no-hot-path --no-mic0-record"
for i in 1 2; do
timeout 900 python3 bench.py --config C4 --steps 20 --warmup 20 --late 300 --late-steps 20 $L > gpurun_out/r06_nan_$i.json 2> gpurun_out/r06_nan_$i.err; echo "bench run $i rc $?"; tail -2 gpurun_out/r06_nan_$i.err
done
timeout 900 python3 tools/soak.py C4 340 > gpurun_out/r06_nan_soak.txt 2>&1; echo "soak rc $?"; tail -3 gpurun_out/r06_nan_soak.txt
