set -x
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/lds_bench && /tmp/lds_bench > gpurun_out/r04_lds_atomic_bench.txt 2>&1
timeout 600 python tools/cell_count_stats.py C4 1 30 60 > gpurun_out/r04_cell_count_stats.txt 2>&1
timeout 600 python bench.py > gpurun_out/r04_baseline_bench.json 2> gpurun_out/r04_baseline_bench.err
tail -3 gpurun_out/r04_cell_count_stats.txt
