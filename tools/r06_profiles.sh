#!/bin/bash
# Round 6's judged measurements (GPU box): bench lines, kernel traces, PMC traffic for C4 and C3 (with their late windows), the other
# configurations' bench lines, the 700-step C3 soak.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/make_profiles.sh r06_c4 --late 300 --late-steps 20 > gpurun_out/r06_make_profiles.log 2>&1
bash tools/make_profiles.sh r06_c3 --config C3 --steps 30 --warmup 20 --late 550 --late-steps 20 > gpurun_out/r06_make_profiles_c3.log 2>&1
bash tools/bench_configs.sh r06 > gpurun_out/r06_bench_configs.log 2>&1
timeout 600 python3 tools/long_run_check.py C3 700 2>&1 | grep -v amdgpu > gpurun_out/r06_soak.txt
timeout 600 python3 tools/long_run_check.py C4 300 2>&1 | grep -v amdgpu > gpurun_out/r06_soak_c4.txt
tail -12 gpurun_out/r06_bench_configs.log
head -16 gpurun_out/r06_c4_kernel_stats.csv
tail -3 gpurun_out/r06_soak.txt
