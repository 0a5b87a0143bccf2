cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_handles.py tests/test_gpu_parity.py -m gpu -x -q -k "multigrid or handles or wait or xcd or fault" 2>&1 | tail -8
L="--no-cpu-baseline --no-hot-path --no-mic0-record"
for V in xcd noxcd xcd noxcd; do
  if [ $V = noxcd ]; then export LFA_MG_NO_XCD=1; else unset LFA_MG_NO_XCD; fi
  for C in C2 C3; do
  python3 bench.py --config $C --steps 30 --warmup 20 $L 2>/dev/null | grep "^{" > /tmp/b.json
  python3 - <<P
import json
o=json.load(open("/tmp/b.json")); sm=o["stage_ms_median"]
print("$V $C ms/step %.3f" % o["ms_per_step"], "pcg_loop %.3f iter %.4f its %.2f" % (sm["pcg_loop"], sm["pcg_iteration_mean"], o["pcg"]["iterations_per_step"]), o["pcg"]["solver_stats_last_solve"]["device_waits_given_up"])
P
  done
done
unset LFA_MG_NO_XCD
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
