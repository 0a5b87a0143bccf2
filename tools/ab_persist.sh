#!/bin/bash
# A/B on the GPU box: the coarse levels of the V-cycle in one launch (default) against a launch per phase (LFA_MG_NO_PERSIST=1).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
for C in C2 C4; do
  for V in persist nopersist; do
    if [ $V = nopersist ]; then export LFA_MG_NO_PERSIST=1; else unset LFA_MG_NO_PERSIST; fi
    python3 bench.py --config $C --steps 20 --warmup 20 $L > gpurun_out/ab_${C}_${V}.json 2> /tmp/ab.err || tail -5 /tmp/ab.err
    python3 - <<P
import json
b = json.load(open("gpurun_out/ab_${C}_${V}.json"))
sm = b.get("stage_ms_median", {})
print("${C} ${V}", "ms/step %.3f" % b["ms_per_step"], "serial %.3f" % b.get("ms_per_step_serial_stages", 0), "it", b["pcg"]["iterations_per_step"], "pcg_iter_ms", sm.get("pcg_iteration_mean"), "pcg_loop", sm.get("pcg_loop"), "groups", (b.get("roofline_groups") or {}).get("p2g_plus_pcg"))
P
  done
done
