cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
L="--no-cpu-baseline --no-mic0-record --no-hot-path"
for CAP in 512 768 640 896 512 768; do
  export LFA_PCG_GRID_CAP=$CAP
  for C in C4 C3 C2; do
  python3 bench.py --config $C --steps 30 --warmup 20 $L 2>/dev/null | grep "^{" > /tmp/b.json
  python3 - <<P
import json
o=json.load(open("/tmp/b.json")); sm=o["stage_ms_median"]
print("cap $CAP $C ms/step %.3f" % o["ms_per_step"], "pcg_loop %.3f iter %.4f serial %.3f" % (sm["pcg_loop"], sm["pcg_iteration_mean"], sm["time_step"]))
P
  done
done
