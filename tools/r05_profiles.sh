cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/make_profiles.sh r05_c4 --late 300 --late-steps 20 > gpurun_out/r05_make_profiles.log 2>&1
bash tools/make_profiles.sh r05_c3 --config C3 --steps 30 --warmup 20 --late 550 --late-steps 20 > gpurun_out/r05_make_profiles_c3.log 2>&1
bash tools/bench_configs.sh r05 > gpurun_out/r05_bench_configs.log 2>&1
tail -12 gpurun_out/r05_bench_configs.log
head -16 gpurun_out/r05_c4_kernel_stats.csv
