cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 20 > gpurun_out/r02a_c4_bench.json 2> /tmp/bench.err || tail -20 /tmp/bench.err
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-hot-path --no-kernel-timing > gpurun_out/r02a_c4_bench_under_rocprof.json 2> /tmp/kt.log || tail -5 /tmp/kt.log
python3 tools/kernel_trace_summary.py "$(find /tmp/kt -name '*results.db' | head -1)" 3 > gpurun_out/r02a_c4_kernel_stats.csv 2>&1
head -40 gpurun_out/r02a_c4_kernel_stats.csv
python3 -c "
import json
b=json.load(open('gpurun_out/r02a_c4_bench.json'))
print(b['value'], b['ms_per_step'], b['pcg'])
print(b['stage_ms_median'])
print(b.get('roofline'))
print(b.get('roofline_groups'))
print(b.get('hbm_ceiling_measured'))
print(b.get('hot_path'))
print(b.get('cpu_baseline'))
"
