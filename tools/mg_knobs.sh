#!/bin/bash
# Iteration count / time per iteration of the multigrid PCG under the V-cycle's experiment knobs (GPU box).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing --no-serial-stages --no-overlap"
run() {  # name, config, env...
  local name=$1 cfg=$2; shift 2
  env "$@" python3 bench.py --config $cfg --steps 20 --warmup 20 $L > /tmp/knob.json 2> /tmp/knob.err || { tail -3 /tmp/knob.err; return; }
  python3 - "$name" "$cfg" <<'P'
import json, sys
b = json.load(open("/tmp/knob.json"))
sm = b.get("stage_ms_median_overlapped") or b.get("stage_ms_median") or {}
print("%-28s %s ms/step %.3f it %.2f pcg_iter_ms %.4f pcg_loop %.3f" % (sys.argv[1], sys.argv[2], b["ms_per_step"], b["pcg"]["iterations_per_step"], sm.get("pcg_iteration_mean", 0), sm.get("pcg_loop", 0)))
P
}
for C in ${CONFIGS:-C2 C3 C4}; do
  run default $C X=1
  run nsw2 $C LFA_MG_NSW=2
  run nsw3 $C LFA_MG_NSW=3
  run stop_single $C LFA_MG_STOP_AT_SINGLE=1
  run stop_single_nsw6 $C LFA_MG_STOP_AT_SINGLE=1 LFA_MG_NSW=6
  run stop_single_nsw2 $C LFA_MG_STOP_AT_SINGLE=1 LFA_MG_NSW=2
  run nopersist $C LFA_MG_NO_PERSIST=1
done
