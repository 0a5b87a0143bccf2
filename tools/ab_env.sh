#!/bin/bash
# tools/ab_env.sh VAR [CONFIGS...] -- A/B of one environment switch on the GPU box: bench.py with VAR unset, then VAR=1.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
VAR=$1; shift
L="--no-cpu-baseline --no-hot-path --no-mic0-record"
for C in ${@:-C4}; do
  for V in default set default set; do
    if [ $V = set ]; then export $VAR=1; else unset $VAR; fi
    python3 bench.py --config $C --steps 20 --warmup 20 $L > gpurun_out/ab_${C}_${V}.json 2> /tmp/ab.err || tail -5 /tmp/ab.err
    python3 - <<P
import json
b = json.load(open("gpurun_out/ab_${C}_${V}.json"))
sm = b.get("stage_ms_median", {})
print("${C} ${VAR} ${V}", "ms/step %.3f" % b["ms_per_step"], "serial %.3f" % b.get("ms_per_step_serial_stages", 0), {k: round(v, 3) for k, v in sm.items() if k in ("p2g", "p2g_scatter_kernel", "g2p", "bin", "pcg_iteration_mean", "correct_collide", "advect_collide", "build_system")})
P
  done
done
