cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
for V in default b6w3 b4w3; do
  if [ $V = default ]; then unset LFA_LIB_PATH; else export LFA_LIB_PATH=$PWD/libfluid_amd/variants/$V.so; fi
  python3 bench.py --steps 20 --warmup 20 $L 2>/dev/null | grep "^{" > /tmp/abl.json
  python3 - <<P
import json
o=json.load(open("/tmp/abl.json")); sm=o["stage_ms_median"]
print("$V ms/step %.3f" % o["ms_per_step"], {k: round(v, 3) for k, v in sm.items() if k in ("advect_collide", "bin", "p2g", "p2g_scatter_kernel", "correct_cell_index", "g2p")})
P
done
unset LFA_LIB_PATH
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -- python3 bench.py --steps 10 --warmup 20 --no-overlap $L > /tmp/kt.json 2> /tmp/kt.log || tail -5 /tmp/kt.log
python3 tools/kernel_trace_summary.py "$(find /tmp/kt -name '*results.db' | head -1)" 3 > gpurun_out/r04_v3_kernel_stats.csv 2>&1
head -30 gpurun_out/r04_v3_kernel_stats.csv
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_next_rows.py -m gpu -x -q 2>&1 | tail -4
