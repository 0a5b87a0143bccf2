cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_handles.py tests/test_gpu_parity.py -m gpu -x -q -k "multigrid or handles or wait or xcd or fault or launch" 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -4
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -- python3 bench.py $L --no-overlap --steps 20 --warmup 20 > /dev/null 2> /tmp/kt.log
python3 tools/kernel_trace_summary.py "$(find /tmp/kt -name '*results.db' | head -1)" 20 | grep "k_mg_coarse\|k_pcg_a\|k_mg_axpy"
for C in C4 C2; do
  python3 bench.py --config $C --steps 30 --warmup 20 $L 2>/dev/null | grep "^{" > /tmp/b.json
  python3 - <<P
import json
o=json.load(open("/tmp/b.json")); sm=o["stage_ms_median"]
print("$C ms/step %.3f" % o["ms_per_step"], "pcg_loop %.3f iter %.4f" % (sm["pcg_loop"], sm["pcg_iteration_mean"]))
P
done
