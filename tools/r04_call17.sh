cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multigrid" > gpurun_out/r04_call17_pytest.log 2>&1; grep -v "^W2026\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|^RCCL" gpurun_out/r04_call17_pytest.log | tail -25
L="--no-cpu-baseline --no-mic0-record --no-hot-path"
for V in merge nomerge merge nomerge; do
  if [ $V = nomerge ]; then export LFA_MG_NO_MERGE=1; else unset LFA_MG_NO_MERGE; fi
  for C in C4 C3; do
  timeout 600 python3 bench.py --config $C --steps 30 --warmup 20 $L 2>/dev/null | grep "^{" > /tmp/b.json
  python3 - <<P
import json
o=json.load(open("/tmp/b.json")); sm=o["stage_ms_median"]
print("$V $C ms/step %.3f" % o["ms_per_step"], "pcg_loop %.3f iter %.4f its %.2f serial %.3f" % (sm["pcg_loop"], sm["pcg_iteration_mean"], o["pcg"]["iterations_per_step"], sm["time_step"]), o["pcg"]["solver_stats_last_solve"]["launches_per_iteration"], o["pcg"]["solver_stats_last_solve"]["device_waits_given_up"])
P
  done
done
