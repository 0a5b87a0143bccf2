cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/make_profiles.sh r04_c4 --late 300 --late-steps 20 > gpurun_out/r04_make_profiles.log 2>&1
bash tools/bench_configs.sh r04 > gpurun_out/r04_bench_configs.log 2>&1
tail -12 gpurun_out/r04_bench_configs.log
head -16 gpurun_out/r04_c4_kernel_stats.csv
