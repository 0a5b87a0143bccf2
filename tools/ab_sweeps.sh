#!/bin/bash
# tools/ab_sweeps.sh ALT.so -- the library as built against a variant of it (e.g. mg.hip compiled with -DMG_INNER_SWEEPS=3): iterations
# and time of the pressure solve early (C3 / C4 step 20-40) and late (C3 step 550).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
ALT=$1
L="--no-cpu-baseline --no-hot-path --no-mic0-record --no-kernel-timing"
cp libfluid_amd/libfluid_amd.so /tmp/lfa_keep.so
for V in base alt; do
  [ $V = alt ] && cp $ALT libfluid_amd/libfluid_amd.so
  for C in C3 C4; do
    python3 bench.py --config $C --steps 20 --warmup 20 $L $([ $C = C3 ] && echo "--late 550 --late-steps 20") 2>/dev/null | grep "^{" > /tmp/absw.json
    python3 - <<P
import json
o=json.load(open("/tmp/absw.json")); sm=o["stage_ms_median"]; l=o.get("late_phase")
print("$V $C ms/step %.3f it %.2f pcg_loop %.3f iter %.4f" % (o["ms_per_step"], o["pcg"]["iterations_per_step_timed_region"], sm["pcg_loop"], sm["pcg_iteration_mean"]), "| late:", l and ("%.3f ms it %.2f pcg_loop %.3f iter %.4f" % (l["ms_per_step"], l["pcg_iterations_per_step"], l["stage_ms_median"]["pcg_loop"], l["stage_ms_median"]["pcg_iteration_mean"])))
P
  done
done
cp /tmp/lfa_keep.so libfluid_amd/libfluid_amd.so
